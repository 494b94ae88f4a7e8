#!/bin/bash
# per-kernel durations (rocprofv3) of two library builds: tools/r4/ab_prof.sh <variant> <cfg>
V=$1; C=${2:-C3}
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
L=$PWD/theano_pyglm_amd
for lib in main $V; do
  OUT=gpurun_out/abp_$lib; rm -rf $OUT; mkdir -p $OUT
  if [ $lib = main ]; then unset PYGLM_HIP_LIB; else export PYGLM_HIP_LIB=$L/libpyglm_hip_$V.so; fi
  TIMING=0 rocprofv3 --kernel-trace --stats -d $OUT/t -- python3 tools/step_bench.py $C > $OUT/log.txt 2>&1
  T=$(find $OUT/t -name '*results.db' | head -1)
  python3 tools/rocprof_summary.py stats "$T" $OUT/stats.csv
  echo "== $lib"; grep -E "k_fused|k_finalize|k_prep" $OUT/stats.csv | cut -c1-60,150-260 | head -5
  rm -rf $OUT/t
done
