#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s5; mkdir -p $OUT
L=$PWD/theano_pyglm_amd
timeout 2400 python -m pytest tests -m gpu -q -s --deselect tests/test_gpu_configs.py::test_map_lockstep_matches_sequential_c5_subset > $OUT/pytest.log 2>&1; tail -8 $OUT/pytest.log; grep -E "C2 structured|correlation of|lock-step BFGS" $OUT/pytest.log
echo "== gibbs kernels"
timeout 900 python tools/gibbs_kernel_only.py 2>&1 | tail -6 | tee $OUT/gibbs_ab.txt
echo "== phase C2 / C5"
for c in C2 C5; do PYGLM_HIP_LIB=$L/libpyglm_hip_prof.so timeout 600 python tools/phase_profile_small.py $c 2>&1 | tail -6 | tee $OUT/phase_$c.txt; done
echo "== MAP bench"
timeout 600 python tools/map_bench.py 128 600 default 2>&1 | tail -3 | tee $OUT/map_default.txt
echo "== C5 MAP check (template sigma)"
timeout 1500 python tools/c5_map_check.py 64 300 poisson 2>&1 | tail -12 | tee $OUT/c5_map_poisson.txt
echo "== C5 MAP check (sigma 1.0)"
timeout 1500 python tools/c5_map_check.py 64 300 poisson 1.0 2>&1 | tail -12 | tee $OUT/c5_map_sigma1.txt
echo "== C5 MAP check (sigma 0.05)"
timeout 1500 python tools/c5_map_check.py 64 300 poisson 0.05 2>&1 | tail -12 | tee $OUT/c5_map_sigma005.txt
