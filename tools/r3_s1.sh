#!/bin/bash
# round-3 GPU session 1: parity of the new build, A/B against the round-2 library, phase profiles of the small kernels
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s1; mkdir -p $OUT
L=$PWD/theano_pyglm_amd
timeout 1200 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_configs.py -m gpu -x -q > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for lib in r2 lds0 new; do
  f=$L/libpyglm_hip_$lib.so; [ $lib = new ] && f=$L/libpyglm_hip.so
  echo "== config table $lib"
  PYGLM_HIP_LIB=$f CFG_ONLY="C1 ,C2 standard_glm,C3 standard_glm,C5 spatio,C3 neuron shard" timeout 900 python tools/config_table.py $OUT/cfg_$lib.json 2>&1 | grep "^|" | tee $OUT/cfg_$lib.md
done
for lib in r2 new; do
  f=$L/libpyglm_hip_$lib.so; [ $lib = new ] && f=$L/libpyglm_hip.so
  echo "== shard step $lib"
  PYGLM_HIP_LIB=$f timeout 600 python tools/shard_step_bench.py 1 8 2>&1 | grep "^G=" | tee $OUT/shard_$lib.txt
done
for c in C2 C5 C1; do
  echo "== phase $c"
  PYGLM_HIP_LIB=$L/libpyglm_hip_prof.so timeout 600 python tools/phase_profile_small.py $c 2>&1 | tail -16 | tee $OUT/phase_$c.txt
done
echo "== rocprof shard 8"
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python3 tools/shard_step_bench.py 8 > $OUT/shard_trace.log 2>&1
T=$(find $OUT/trace -name '*results.db' | head -1)
python3 tools/rocprof_summary.py stats "$T" $OUT/shard8_kernel_stats.csv; head -12 $OUT/shard8_kernel_stats.csv
rm -rf $OUT/trace
echo "== MAP bench (default path)"
timeout 600 python tools/map_bench.py 128 600 default 2>&1 | tail -5 | tee $OUT/map_default.txt
timeout 600 python tools/map_bench.py 32 300 default 2>&1 | tail -5 | tee -a $OUT/map_default.txt
