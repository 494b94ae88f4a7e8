#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s8; mkdir -p $OUT
L=$PWD/theano_pyglm_amd
timeout 2400 python -m pytest tests -m gpu -q -x > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
for lib in prev new prev new; do
  f=$L/libpyglm_hip_$lib.so; [ $lib = new ] && f=$L/libpyglm_hip.so
  echo "== config table $lib"
  PYGLM_HIP_LIB=$f CFG_ONLY="C1 ,C2 standard_glm,C3 standard_glm,C5 spatio" timeout 900 python tools/config_table.py 2>&1 | grep "^| C" | grep -v "in-kernel\|resident K"
done
bash tools/prof_small_pmc.sh C2 2>&1 | grep -A22 "k_fused6" | grep -E "k_fused|CONFLICT|IDX_ACTIVE|MFMA_BUSY|GRBM|INSTS_VALU|INSTS_MFMA"
bash tools/prof_small_pmc.sh C3 2>&1 | grep -E "k_fused5|CONFLICT|IDX_ACTIVE|MFMA_BUSY|GRBM|INSTS_VALU |INSTS_MFMA"
