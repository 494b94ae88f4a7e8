#!/bin/bash
# every fuzzer on the current build (via gpurun):  tools/fuzz_all.sh [first kernel seed] [seeds]   -> one summary line per seed
S0=${1:-0}; NS=${2:-4}
for ((s=S0; s<S0+NS; s++)); do echo "== fuzz_kernels seed $s"; timeout 900 python tools/fuzz_kernels.py $s 2>&1 | grep -v amdgpu | tail -4 | cut -c1-300; done
for ((s=S0+1; s<S0+NS; s++)); do echo "== fuzz_sepf seed $s"; timeout 600 python tools/fuzz_sepf.py $s 2>&1 | grep -v amdgpu | tail -2 | cut -c1-300; done
for ((s=S0; s<S0+2; s++)); do echo "== fuzz_gibbs seed $s"; timeout 600 python tools/fuzz_gibbs.py $s 2>&1 | grep -v amdgpu | tail -2 | cut -c1-300; done
echo "== fuzz_map seeds $((S0*16)) .. $((S0*16+NS*8-1))"; timeout 1200 python tools/fuzz_map.py $(seq $((S0*16)) $((S0*16+NS*8-1))) 2>&1 | grep -v amdgpu | grep "DISCREPANCY\|seeds," | cut -c1-400
