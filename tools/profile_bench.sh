#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace + separate PMC passes of bench.py,
# summaries into profiles/<tag>_*.  Usage: tools/profile_bench.sh <tag> [bench args...]
set -u
TAG=${1:-rXX}; shift || true
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python3 -c "import __graft_entry__ as g; g.build_hip(); g.build_oracle()"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT" profiles
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-map --no-mcmc --no-ab "$@" > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-map --no-mcmc --no-ab "$@" > $OUT/bench_pmc1.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-map --no-mcmc --no-ab "$@" > $OUT/bench_pmc2.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d $OUT/pmc_sq -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-map --no-mcmc --no-ab "$@" > $OUT/bench_pmc3.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES -d $OUT/pmc_mfma -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-map --no-mcmc --no-ab "$@" > $OUT/bench_pmc4.log 2>&1
T=$(find $OUT/trace -name '*results.db' | head -1)
python3 tools/rocprof_summary.py stats "$T" profiles/${TAG}_kernel_stats.csv
python3 tools/rocprof_summary.py pmc $(find $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_sq $OUT/pmc_mfma -name '*results.db') profiles/${TAG}_pmc.json
grep -h '"metric"' $OUT/bench_trace.log > profiles/${TAG}_bench_line.json
cp profiles/${TAG}_* gpurun_out/ 2>/dev/null
rm -rf "$OUT"   # keep gpurun_out small (merge limit 64 MiB); the summaries are what is kept
cat profiles/${TAG}_kernel_stats.csv | head -6
python3 - <<PY
import json
d=json.load(open('profiles/${TAG}_pmc.json'))
for k,v in d.items():
    if 'fused' in k: print(k, {kk:(vv.get('avg') if isinstance(vv,dict) and 'avg' in vv else vv) for kk,vv in v.items()})
PY
