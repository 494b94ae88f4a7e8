#!/bin/bash
# Build the microbenchmarks under tools/ubench (hipcc cross-compiles without a GPU): <name>_ubench.hip -> <name>_ubench
# (executables are git-ignored; they travel to the GPU box with gpurun).  Usage: tools/build_ubench.sh [name ...]
set -e
cd "$(dirname "$0")/ubench"
names=("$@"); [ ${#names[@]} -eq 0 ] && names=($(ls *_ubench.hip | sed 's/_ubench.hip//'))
for n in "${names[@]}"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o ${n}_ubench ${n}_ubench.hip 2>&1 | grep -E "error" -A3 || true
  ls -la ${n}_ubench | awk '{print $NF, $5}'
done
