#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s4; mkdir -p $OUT
L=$PWD/theano_pyglm_amd
timeout 2400 python -m pytest tests -m gpu -q -s --deselect tests/test_gpu_configs.py::test_map_lockstep_matches_sequential_c5_subset > $OUT/pytest.log 2>&1; tail -8 $OUT/pytest.log; grep -E "C2 structured|correlation of|lock-step BFGS" $OUT/pytest.log
echo "== config table new"
CFG_ONLY="C1 ,C2 standard_glm,C3 standard_glm,C5 spatio" timeout 900 python tools/config_table.py $OUT/cfg_new.json 2>&1 | grep "^|" | tee $OUT/cfg_new.md
echo "== phase C2"
PYGLM_HIP_LIB=$L/libpyglm_hip_prof.so timeout 600 python tools/phase_profile_small.py C2 2>&1 | tail -14 | tee $OUT/phase_C2.txt
echo "== MAP bench"
timeout 600 python tools/map_bench.py 128 600 default 2>&1 | tail -4 | tee $OUT/map_default.txt
timeout 600 python tools/map_bench.py 32 300 default 2>&1 | tail -4 | tee -a $OUT/map_default.txt
echo "== C5 MAP check (poisson)"
timeout 1500 python tools/c5_map_check.py 64 300 poisson 2>&1 | tail -14 | tee $OUT/c5_map_poisson.txt
echo "== C5 MAP check (model)"
timeout 1500 python tools/c5_map_check.py 64 300 model 2>&1 | tail -14 | tee $OUT/c5_map_model.txt
