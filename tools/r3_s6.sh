#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s6; mkdir -p $OUT
L=$PWD/theano_pyglm_amd
echo "== gibbs"
python tools/gibbs_kernel_only.py 2>&1 | tail -6
python tools/fuzz_gibbs.py 2>&1 | tail -6
timeout 2400 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; tail -6 $OUT/pytest.log; grep -E "lock-step BFGS|^E  " $OUT/pytest.log | head -20
for lib in w52 new; do
  f=$L/libpyglm_hip_$lib.so; [ $lib = new ] && f=$L/libpyglm_hip.so
  echo "== config table $lib"
  PYGLM_HIP_LIB=$f CFG_ONLY="C2 standard_glm,C5 spatio" timeout 900 python tools/config_table.py 2>&1 | grep "^| C"
done
