#!/bin/bash
# rocprofv3 kernel stats of tools/quick_bench.py (dev tool): tools/prof_quick.sh [quick_bench args]
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_quick
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python3 tools/quick_bench.py "$@" > $OUT/log.txt 2>&1
T=$(find $OUT/trace -name '*results.db' | head -1)
python3 tools/rocprof_summary.py stats "$T" $OUT/stats.csv
head -8 $OUT/stats.csv
rm -rf "$OUT/trace"
