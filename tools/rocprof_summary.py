"""Turn rocprofv3's rocpd sqlite output (gpurun_out/prof/...) into the small text
summaries committed under profiles/.

  python tools/rocprof_summary.py stats <results.db> <out.csv>
  python tools/rocprof_summary.py pmc <results.db> [<results.db> ...] <out.json>
"""
import json
import sqlite3
import sys


def stats(db, out):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage "
                          "from top_kernels order by total_duration desc"))
    with open(out, 'w') as f:
        f.write("# rocprofv3 --kernel-trace --stats summary (durations in us; view top_kernels)\n")
        f.write("kernel,calls,total_us,avg_us,percent\n")
        for name, calls, tot, avg, pct in rows:
            f.write('"%s",%d,%.3f,%.3f,%.3f\n' % (name, calls, tot, avg, pct))
    # per-dispatch resource usage of the dominant kernel
    try:
        r = c.execute("select name, vgpr_count, accum_vgpr_count, sgpr_count, lds_block_size, "
                      "grid_size, workgroup_size from kernels limit 0")
    except Exception:
        pass


def pmc(dbs, out):
    res = {}
    for db in dbs:
        c = sqlite3.connect(db)
        q = ("select kernel_name, counter_name, avg(value), min(value), max(value), count(*), "
             "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_block_size), "
             "max(grid_size), max(workgroup_size) "
             "from counters_collection group by kernel_name, counter_name")
        for k, cn, avg, mn, mx, n, vg, ag, sg, lds, grid, wg in c.execute(q):
            d = res.setdefault(k, {})
            d[cn] = {"avg": avg, "min": mn, "max": mx, "dispatches": n}
            d["_launch"] = {"vgpr": vg, "agpr": ag, "sgpr": sg, "lds_bytes": lds,
                            "grid_threads": grid, "workgroup": wg}
    with open(out, 'w') as f:
        json.dump(res, f, indent=1, sort_keys=True)


def timeline(db, out, last=40):
    """Per-dispatch timeline of the last `last` kernel dispatches: start (us, relative), duration, gap to the end of
    the previous dispatch on the device, queue/stream -- shows what is exposed between the kernels of one evaluation."""
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    name = 'name' if 'name' in cols else 'kernel_name'
    extra = [k for k in ('stream_id', 'queue_id', 'stream') if k in cols]
    q = "select %s, start, end%s from kernels order by start" % (name, ''.join(', ' + k for k in extra))
    rows = list(c.execute(q))
    rows = rows[-last:]
    t0 = rows[0][1]
    with open(out, 'w') as f:
        f.write("# last %d kernel dispatches: start_us,dur_us,gap_after_prev_end_us,stream,kernel\n" % len(rows))
        prev_end = None
        for r in rows:
            nm, st, en = r[0], r[1], r[2]
            gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
            f.write("%.1f,%.1f,%.1f,%s,%s\n" % ((st - t0) / 1e3, (en - st) / 1e3, gap,
                                               '/'.join(str(x) for x in r[3:]), nm[:60]))
            prev_end = en if prev_end is None else max(prev_end, en)


if __name__ == '__main__':
    if sys.argv[1] == 'timeline':
        timeline(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 40)
    elif sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2:-1], sys.argv[-1])
