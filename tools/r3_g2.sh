#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for v in ${VARIANTS}; do
  if [ "$v" = main ]; then unset PYGLM_HIP_LIB; else export PYGLM_HIP_LIB=$PWD/theano_pyglm_amd/libpyglm_hip_$v.so; fi
  echo "== $v"
  for c in ${CFGS:-C5}; do TIMING=1 timeout 300 python3 tools/step_bench.py $c 2>&1 | grep "per evaluation" | cut -c1-150; done
done
