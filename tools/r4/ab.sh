#!/bin/bash
# A/B of two library builds inside one gpurun call: tools/r4/ab.sh <variant-name> <cfg ...>   (interleaved runs)
V=$1; shift
cd "$GRAFT_REPO_ROOT"
L=$PWD/theano_pyglm_amd
for rep in 1 2 3; do
  for c in "$@"; do
    echo -n "main  "; python3 tools/step_bench.py $c 2>&1 | tail -1 | cut -c1-110
    echo -n "$V  "; PYGLM_HIP_LIB=$L/libpyglm_hip_$V.so python3 tools/step_bench.py $c 2>&1 | tail -1 | cut -c1-110
  done
done
