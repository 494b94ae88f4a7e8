#!/bin/bash
# round-3 GPU session 3: full GPU suite, config table, workgroup timelines, dispatch timelines (shard step, MAP)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s3; mkdir -p $OUT
L=$PWD/theano_pyglm_amd
timeout 2400 python -m pytest tests -m gpu -q -s > $OUT/pytest.log 2>&1; tail -15 $OUT/pytest.log; grep -E "C2 structured|correlation of|lock-step BFGS" $OUT/pytest.log
echo "== config table new"
CFG_ONLY="C1 ,C2 standard_glm,C3 standard_glm,C5 spatio,C3 neuron shard" timeout 900 python tools/config_table.py $OUT/cfg_new.json 2>&1 | grep "^|" | tee $OUT/cfg_new.md
echo "== shard step"
timeout 600 python tools/shard_step_bench.py 1 8 2>&1 | grep "^G=" | tee $OUT/shard.txt
for c in C2 C5 C1; do
  echo "== phase $c"
  PYGLM_HIP_LIB=$L/libpyglm_hip_prof.so timeout 600 python tools/phase_profile_small.py $c 2>&1 | tail -18 | tee $OUT/phase_$c.txt
done
echo "== phase C3 (k_fused5), 1/8 of the recording"
PYGLM_HIP_LIB=$L/libpyglm_hip_prof.so timeout 600 python tools/phase_profile.py 128 75 2>&1 | tail -22 | tee $OUT/phase_C3_75s.txt
echo "== rocprof shard 8 timeline"
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python3 tools/shard_step_bench.py 8 > $OUT/shard_trace.log 2>&1
T=$(find $OUT/trace -name '*results.db' | head -1)
python3 tools/rocprof_summary.py stats "$T" $OUT/shard8_kernel_stats.csv; head -6 $OUT/shard8_kernel_stats.csv
python3 tools/rocprof_summary.py timeline "$T" $OUT/shard8_timeline.csv 16; cat $OUT/shard8_timeline.csv
rm -rf $OUT/trace
echo "== rocprof MAP"
rocprofv3 --kernel-trace --stats -d $OUT/trace2 -- python3 tools/map_bench.py 128 600 default > $OUT/map_trace.log 2>&1
T=$(find $OUT/trace2 -name '*results.db' | head -1)
python3 tools/rocprof_summary.py stats "$T" $OUT/map_kernel_stats.csv; head -30 $OUT/map_kernel_stats.csv
tail -4 $OUT/map_trace.log
rm -rf $OUT/trace2
