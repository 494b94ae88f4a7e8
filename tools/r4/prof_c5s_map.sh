#!/bin/bash
# kernel trace of the C5-stress MAP sweep at reduced T (dev): tools/r4/prof_c5s_map.sh
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_c5s_map; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/t -- python3 tools/r4/c5_stress_map.py 64 40 1024 > $OUT/log.txt 2>&1
T=$(find $OUT/t -name '*results.db' | head -1)
python3 tools/rocprof_summary.py stats "$T" $OUT/stats.csv
head -16 $OUT/stats.csv | cut -c1-90,100-400 | cut -c1-170
tail -5 $OUT/log.txt
rm -rf $OUT/t
