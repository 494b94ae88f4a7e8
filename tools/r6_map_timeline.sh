#!/bin/bash
# MAP sweeps under rocprofv3: kernel stats + the dispatch timeline of the last launches (gaps between dependent kernels)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
R=gpurun_out/r6; mkdir -p $R
TAG=${1:-a}
rocprofv3 --kernel-trace --stats -d $R/tr_c2 -- python3 tools/map_bench.py 32 300 default > $R/tr_c2.log 2>&1
T=$(find $R/tr_c2 -name '*results.db' | head -1)
python3 tools/rocprof_summary.py stats "$T" $R/map_C2_kernel_stats_$TAG.csv
python3 tools/rocprof_summary.py timeline "$T" $R/map_C2_timeline_$TAG.csv 70; rm -rf $R/tr_c2
rocprofv3 --kernel-trace --stats -d $R/tr_c3 -- python3 tools/map_bench.py 128 600 default > $R/tr_c3.log 2>&1
T=$(find $R/tr_c3 -name '*results.db' | head -1)
python3 tools/rocprof_summary.py stats "$T" $R/map_C3_kernel_stats_$TAG.csv
python3 tools/rocprof_summary.py timeline "$T" $R/map_C3_timeline_$TAG.csv 70; rm -rf $R/tr_c3
rocprofv3 --kernel-trace --stats -d $R/tr_c5 -- python3 tools/stress_map.py --maxiter 225 --reps 2 > $R/tr_c5.log 2>&1
T=$(find $R/tr_c5 -name '*results.db' | head -1)
python3 tools/rocprof_summary.py stats "$T" $R/map_C5stress_kernel_stats_$TAG.csv
python3 tools/rocprof_summary.py timeline "$T" $R/map_C5stress_timeline_$TAG.csv 70; rm -rf $R/tr_c5
head -14 $R/map_C2_kernel_stats_$TAG.csv | cut -c1-160
