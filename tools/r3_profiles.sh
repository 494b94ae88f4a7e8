#!/bin/bash
# Round-3 evidence run (via gpurun): every summary that DESIGN.md / README.md cite, into gpurun_out/r03/ (copied to profiles/ afterwards).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
R=gpurun_out/r03; rm -rf $R; mkdir -p $R profiles
L=$PWD/theano_pyglm_amd
python3 -c "import __graft_entry__ as g; g.build_hip(); g.build_oracle()"
echo "== bench under rocprofv3 (kernel trace + PMC passes)"
bash tools/profile_bench.sh r03 > $R/profile_bench.log 2>&1; cp profiles/r03_* $R/ 2>/dev/null; tail -3 $R/profile_bench.log
echo "== config table"
bash tools/profile_configs.sh r03 > $R/profile_configs.log 2>&1; cp profiles/r03_config* $R/ 2>/dev/null; cat profiles/r03_config_table.md
echo "== steady-state cost per evaluation"
for t in 1 0; do for c in C1 C2 C5 C3; do TIMING=$t python3 tools/step_bench.py $c 2>&1 | tail -1; done; done | tee $R/r03_step_bench.txt
echo "== time shards"
(echo "# HIP events around every evaluation"; python3 tools/shard_step_bench.py 1 2 4 8 2>&1 | grep "^G="; echo "# no events (the product path)"; TIMING=0 python3 tools/shard_step_bench.py 1 2 4 8 2>&1 | grep "^G=") | tee $R/r03_shard_steps.txt
rocprofv3 --kernel-trace --stats -d $R/trace -- python3 tools/shard_step_bench.py 8 > $R/shard_trace.log 2>&1
T=$(find $R/trace -name '*results.db' | head -1)
python3 tools/rocprof_summary.py stats "$T" $R/r03_shard8_kernel_stats.csv
python3 tools/rocprof_summary.py timeline "$T" $R/r03_shard8_timeline.csv 12; cat $R/r03_shard8_timeline.csv
rm -rf $R/trace
echo "== PMC of the small configurations"
for c in C2 C5; do bash tools/prof_small_pmc.sh $c > $R/pmc_$c.log 2>&1; cp gpurun_out/pmc_$c/pmc.json $R/r03_pmc_$c.json; grep -A12 "k_fused" $R/pmc_$c.log | head -14; done
echo "== phase profiles"
for c in C2 C5 C1; do PYGLM_HIP_LIB=$L/libpyglm_hip_prof.so python3 tools/phase_profile_small.py $c 2>&1 | tail -18 > $R/r03_phase_$c.txt; tail -16 $R/r03_phase_$c.txt; done
PYGLM_HIP_LIB=$L/libpyglm_hip_prof.so python3 tools/phase_profile.py 128 600 2>&1 | tail -24 > $R/r03_phase_C3.txt; cat $R/r03_phase_C3.txt
PYGLM_HIP_LIB=$L/libpyglm_hip_prof.so python3 tools/phase_profile.py 128 75 2>&1 | tail -24 > $R/r03_phase_C3_eighth.txt
echo "== Gibbs"
bash tools/prof_gibbs.sh > $R/prof_gibbs.log 2>&1; cp gpurun_out/prof_gibbs/stats.csv $R/r03_gibbs_kernel_stats.csv; cp gpurun_out/prof_gibbs/pmc.json $R/r03_gibbs_pmc.json; head -5 $R/r03_gibbs_kernel_stats.csv
python3 tools/gibbs_kernel_only.py 2>&1 | tail -5 | tee $R/r03_gibbs_launch.txt
python3 tools/gibbs_x_hist.py 2>&1 | tail -11 > $R/r03_gibbs_x_hist.txt
python3 tools/gibbs_ablate.py 2>&1 | grep "^dbg" | tee $R/r03_gibbs_ablation.txt
python3 tools/gibbs_sweep_profile.py 2>&1 | tail -7 | tee $R/r03_gibbs_sweep.txt
./tools/ubench/occ_gibbs_ubench 2>&1 | grep "workgroups\|shared" | tee $R/r03_gibbs_occupancy.txt
echo "== VALU issue costs (tools/ubench/valu_rates_ubench.hip)"
./tools/ubench/valu_rates_ubench 2>&1 | grep "waves/WG 16" | tee $R/r03_valu_issue_costs.txt
echo "== MAP"
python3 tools/map_bench.py 128 600 default 2>&1 | tail -4 | tee $R/r03_map.txt
python3 tools/map_bench.py 32 300 default 2>&1 | tail -4 | tee -a $R/r03_map.txt
python3 tools/map_bench.py 128 600 seq 2>&1 | tail -2 | head -1 | tee -a $R/r03_map.txt
rocprofv3 --kernel-trace --stats -d $R/trace2 -- python3 tools/map_bench.py 128 600 default > $R/map_trace.log 2>&1
T=$(find $R/trace2 -name '*results.db' | head -1)
python3 tools/rocprof_summary.py stats "$T" $R/r03_map_kernel_stats.csv; rm -rf $R/trace2
python3 - <<'PY'
import csv
rows=list(csv.reader(l for l in open('gpurun_out/r03/r03_map_kernel_stats.csv') if not l.startswith('#')))
print("MAP kernel time by kernel (3 sweeps), top 12:")
for r in rows[1:13]: print("  %-60s calls %s total_us %s" % (r[0][:60], r[1], r[2]))
PY
echo "== two ranks on one GPU (gloo): both shardings of bench.py"
for sh in time neurons; do
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --shard $sh --debug-single-device --no-cpu-baseline --no-map --no-mcmc 2>/dev/null | grep '"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$sh', d['value'], d['ms_per_step'], json.dumps(d.get('per_rank')))"
done | tee $R/r03_two_rank_single_gpu.txt
echo "== bench.py multi-rank path on RCCL, one rank"
for sh in time neurons; do python3 bench.py --rccl-selftest --shard $sh --steps 20 --warmup 3 2>/dev/null | tail -1; done | tee $R/r03_rccl_selftest.jsonl | cut -c1-300
echo "== GPU test suite"
python3 -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $R/r03_gpu_tests.txt
rm -rf gpurun_out/pmc_C2 gpurun_out/pmc_C5 gpurun_out/prof_gibbs
ls $R
