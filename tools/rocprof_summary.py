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


if __name__ == '__main__':
    if sys.argv[1] == 'stats':
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2:-1], sys.argv[-1])
