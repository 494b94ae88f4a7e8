#!/bin/bash
# A/B several builds of the library on the same box: tools/ab_libs.sh lib1.so lib2.so ... (paths relative to repo root)
# quick_bench runs 3 evaluations per "dbg" entry: 0,0,0 = 9 evaluations, the last one is reported (steady state)
for rep in $(seq 1 ${AB_REPS:-2}); do
for lib in "$@"; do
  r=$(PYGLM_HIP_LIB=$PWD/$lib python tools/quick_bench.py 128 600 0 0 0,0,0 2>&1 | grep "iter 2" | tail -1 | sed 's/.*fused \([0-9.]*\) ms total \([0-9.]*\) ms.*/\1 total \2/')
  echo "$lib fused_ms $r"
done; done
