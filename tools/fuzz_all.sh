for s in 0 1 2 3; do echo "== fuzz_kernels seed $s"; timeout 900 python tools/fuzz_kernels.py $s 2>&1 | grep -v amdgpu | tail -4 | cut -c1-300; done
for s in 1 2 3; do echo "== fuzz_sepf seed $s"; timeout 600 python tools/fuzz_sepf.py $s 2>&1 | grep -v amdgpu | tail -2 | cut -c1-300; done
for s in 0 1; do echo "== fuzz_gibbs seed $s"; timeout 600 python tools/fuzz_gibbs.py $s 2>&1 | grep -v amdgpu | tail -2 | cut -c1-300; done
