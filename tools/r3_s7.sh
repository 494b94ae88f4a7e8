#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s7; mkdir -p $OUT
L=$PWD/theano_pyglm_amd
timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
for lib in prev new prev new; do
  f=$L/libpyglm_hip_$lib.so; [ $lib = new ] && f=$L/libpyglm_hip.so
  echo "== config table $lib"
  PYGLM_HIP_LIB=$f CFG_ONLY="C1 ,C2 standard_glm,C3 standard_glm,C5 spatio" timeout 900 python tools/config_table.py 2>&1 | grep "^| C" | grep -v "in-kernel\|resident K"
done
echo "== shard step"
for lib in prev new; do
  f=$L/libpyglm_hip_$lib.so; [ $lib = new ] && f=$L/libpyglm_hip.so
  PYGLM_HIP_LIB=$f timeout 600 python tools/shard_step_bench.py 1 8 2>&1 | grep "^G="
done
echo "== phase C2"
PYGLM_HIP_LIB=$L/libpyglm_hip_prof.so timeout 600 python tools/phase_profile_small.py C2 2>&1 | tail -14
