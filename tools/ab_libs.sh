#!/bin/bash
# A/B several builds of the library on the same box: tools/ab_libs.sh lib1.so lib2.so ... (paths relative to repo root)
for rep in 1 2; do
for lib in "$@"; do
  r=$(PYGLM_HIP_LIB=$PWD/$lib python tools/quick_bench.py 128 600 0 0 0 2>&1 | grep "iter 2" | sed 's/.*fused \([0-9.]*\) ms.*/\1/')
  echo "$lib fused_ms $r"
done; done
