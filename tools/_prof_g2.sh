#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "gibbs" 2>&1 | tail -5
OUT=gpurun_out/prof_g2
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python3 tools/gibbs_kernel_only.py > $OUT/log1.txt 2>&1
python3 tools/rocprof_summary.py stats $(find $OUT/trace -name '*results.db' | head -1) $OUT/stats.csv
head -8 $OUT/stats.csv | cut -c1-200
rm -rf $OUT/trace
python tools/gibbs_sweep_profile.py 2>&1 | tail -4
