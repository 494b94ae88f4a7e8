#!/bin/bash
# A/B of the Gibbs rate kernel between the shipped library and variants: tools/r4/ab_gibbs.sh <variant> [<variant> ...]
cd "$GRAFT_REPO_ROOT"
L=$PWD/theano_pyglm_amd
for rep in 1 2; do
  echo "== main"; python3 tools/gibbs_kernel_only.py 2>&1 | tail -5
  for V in "$@"; do
    echo "== $V"; PYGLM_HIP_LIB=$L/libpyglm_hip_$V.so python3 tools/gibbs_kernel_only.py 2>&1 | tail -5
  done
done
