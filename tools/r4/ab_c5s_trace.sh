#!/bin/bash
# kernel trace of the C5 stress evaluation for the shipped library and a variant: tools/r4/ab_c5s_trace.sh <variant>
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for lib in main $1; do
  R=gpurun_out/abtrace_$lib; rm -rf $R; mkdir -p $R
  if [ $lib = main ]; then unset PYGLM_HIP_LIB; else export PYGLM_HIP_LIB=$PWD/theano_pyglm_amd/libpyglm_hip_$lib.so; fi
  rocprofv3 --kernel-trace --stats -d $R/t -- python3 tools/cfg_loop.py C5S 12 > $R/log 2>&1
  python3 tools/rocprof_summary.py stats "$(find $R/t -name '*results.db' | head -1)" $R/stats.csv
  echo "== $lib"; grep "k_sepf\|k_fused7" $R/stats.csv | cut -c1-45,80-200 | cut -c1-120; tail -1 $R/log
  rm -rf $R/t
done
