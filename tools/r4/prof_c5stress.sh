#!/bin/bash
# rocprofv3 kernel stats of the C5 stress variant (tools/c5_stress.py): tools/r4/prof_c5stress.sh [tag]
TAG=${1:-c5stress}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
python3 tools/c5_stress.py > $OUT/plain.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python3 tools/c5_stress.py > $OUT/log.txt 2>&1
T=$(find $OUT/trace -name '*results.db' | head -1)
python3 tools/rocprof_summary.py stats "$T" $OUT/stats.csv
head -30 $OUT/stats.csv
cat $OUT/plain.txt
rm -rf "$OUT/trace"
