#!/bin/bash
# Round-4 evidence run (via gpurun): every summary that DESIGN.md / README.md cite, into gpurun_out/r04/ (copied to profiles/ afterwards).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
R=gpurun_out/r04; rm -rf $R; mkdir -p $R profiles
L=$PWD/theano_pyglm_amd
python3 -c "import __graft_entry__ as g; g.build_hip(); g.build_oracle()"
echo "== bench under rocprofv3 (kernel trace + PMC passes)"
bash tools/profile_bench.sh r04 --no-stim > $R/profile_bench.log 2>&1; cp profiles/r04_* $R/ 2>/dev/null; tail -3 $R/profile_bench.log
echo "== config table"
bash tools/profile_configs.sh r04 > $R/profile_configs.log 2>&1; cp profiles/r04_config* $R/ 2>/dev/null; cat profiles/r04_config_table.md
echo "== steady-state cost per evaluation"
for t in 1 0; do for c in C1 C2 C5 C3; do TIMING=$t python3 tools/step_bench.py $c 2>&1 | tail -1; done; done | tee $R/r04_step_bench.txt
echo "== time shards"
(echo "# HIP events around every evaluation"; python3 tools/shard_step_bench.py 1 2 4 8 2>&1 | grep "^G="; echo "# no events (the product path)"; TIMING=0 python3 tools/shard_step_bench.py 1 2 4 8 2>&1 | grep "^G=") | tee $R/r04_shard_steps.txt
rocprofv3 --kernel-trace --stats -d $R/trace -- python3 tools/shard_step_bench.py 8 > $R/shard_trace.log 2>&1
T=$(find $R/trace -name '*results.db' | head -1)
python3 tools/rocprof_summary.py stats "$T" $R/r04_shard8_kernel_stats.csv
python3 tools/rocprof_summary.py timeline "$T" $R/r04_shard8_timeline.csv 12; cat $R/r04_shard8_timeline.csv
rm -rf $R/trace
echo "== PMC of the small configurations"
for c in C2 C5; do bash tools/prof_small_pmc.sh $c > $R/pmc_$c.log 2>&1; cp gpurun_out/pmc_$c/pmc.json $R/r04_pmc_$c.json; grep -A12 "k_fused" $R/pmc_$c.log | head -14; done
echo "== C5 stress variant (separable stimulus at the frame rate): kernel trace + PMC"
rocprofv3 --kernel-trace --stats -d $R/trace_c5s -- python3 tools/cfg_loop.py C5S 12 > $R/c5s_trace.log 2>&1
T=$(find $R/trace_c5s -name '*results.db' | head -1)
python3 tools/rocprof_summary.py stats "$T" $R/r04_C5stress_kernel_stats.csv
python3 tools/rocprof_summary.py timeline "$T" $R/r04_C5stress_timeline.csv 18; rm -rf $R/trace_c5s
head -14 $R/r04_C5stress_kernel_stats.csv
bash tools/prof_small_pmc.sh C5S > $R/pmc_C5S.log 2>&1; cp gpurun_out/pmc_C5S/pmc.json $R/r04_pmc_C5stress.json; rm -rf gpurun_out/pmc_C5S
python3 - <<'PY'
import json
d = json.load(open('gpurun_out/r04/r04_pmc_C5stress.json'))
for k, v in d.items():
    if any(t in k for t in ('k_fused7', 'k_sepf', 'k_gemm_mfma')):
        g = lambda n: v[n]['avg'] if n in v else float('nan')
        print("%-40s fetch %.1f MB (x2 for 16-byte streams) write %.1f MB  MFMA busy %.3f" % (k[:40], g('FETCH_SIZE') / 1024, g('WRITE_SIZE') / 1024,
              g('SQ_VALU_MFMA_BUSY_CYCLES') / (g('GRBM_GUI_ACTIVE') / 8 * 1024)))
PY
echo "== C5 stress: MAP sweep through the host mirror (bench.py --stim-map)"
python3 bench.py --steps 5 --warmup 2 --no-map --no-mcmc --no-cpu-baseline --no-ab --stim-map 2>/dev/null | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read())['secondary_stim']; print(json.dumps(d, indent=1))" | tee $R/r04_C5stress_bench_block.json | grep -E '"value"|map_sweep|iterations|evaluations'
echo "== phase profiles"
for c in C2 C5 C1; do PYGLM_HIP_LIB=$L/libpyglm_hip_prof.so python3 tools/phase_profile_small.py $c 2>&1 | tail -18 > $R/r04_phase_$c.txt; tail -16 $R/r04_phase_$c.txt; done
PYGLM_HIP_LIB=$L/libpyglm_hip_prof.so python3 tools/phase_profile.py 128 600 2>&1 | tail -24 > $R/r04_phase_C3.txt; cat $R/r04_phase_C3.txt
PYGLM_HIP_LIB=$L/libpyglm_hip_prof.so python3 tools/phase_profile.py 128 75 2>&1 | tail -24 > $R/r04_phase_C3_eighth.txt
echo "== Gibbs"
bash tools/prof_gibbs.sh > $R/prof_gibbs.log 2>&1; cp gpurun_out/prof_gibbs/stats.csv $R/r04_gibbs_kernel_stats.csv; cp gpurun_out/prof_gibbs/pmc.json $R/r04_gibbs_pmc.json; head -5 $R/r04_gibbs_kernel_stats.csv
python3 tools/gibbs_kernel_only.py 2>&1 | tail -5 | tee $R/r04_gibbs_launch.txt
python3 tools/gibbs_x_hist.py 2>&1 | tail -11 > $R/r04_gibbs_x_hist.txt
PYGLM_HIP_LIB=$L/libpyglm_hip_ablate.so python3 tools/gibbs_ablate.py 2>&1 | grep "^dbg" | tee $R/r04_gibbs_ablation.txt
python3 tools/gibbs_sweep_profile.py 2>&1 | tail -7 | tee $R/r04_gibbs_sweep.txt
python3 tools/r4/gibbs_nloop_scan.py 2>&1 | grep nloop > $R/r04_gibbs_nloop_scan.txt
./tools/ubench/occ_gibbs_ubench 2>&1 | grep "workgroups\|shared" | tee $R/r04_gibbs_occupancy.txt
echo "== VALU issue costs (tools/ubench/valu_rates_ubench.hip)"
./tools/ubench/valu_rates_ubench 2>&1 | grep "waves/WG 16" | tee $R/r04_valu_issue_costs.txt
echo "== MAP"
python3 tools/map_bench.py 128 600 default 2>&1 | tail -4 | tee $R/r04_map.txt
python3 tools/map_bench.py 32 300 default 2>&1 | tail -4 | tee -a $R/r04_map.txt
python3 tools/map_bench.py 128 600 seq 2>&1 | tail -2 | head -1 | tee -a $R/r04_map.txt
rocprofv3 --kernel-trace --stats -d $R/trace2 -- python3 tools/map_bench.py 128 600 default > $R/map_trace.log 2>&1
T=$(find $R/trace2 -name '*results.db' | head -1)
python3 tools/rocprof_summary.py stats "$T" $R/r04_map_kernel_stats.csv; rm -rf $R/trace2
python3 - <<'PY'
import csv
rows=list(csv.reader(l for l in open('gpurun_out/r04/r04_map_kernel_stats.csv') if not l.startswith('#')))
print("MAP kernel time by kernel (3 sweeps), top 12:")
for r in rows[1:13]: print("  %-60s calls %s total_us %s" % (r[0][:60], r[1], r[2]))
PY
echo "== two ranks on one GPU (gloo): both shardings of bench.py"
for sh in time neurons; do
  python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 10 --warmup 3 --shard $sh --debug-single-device --no-cpu-baseline --no-map --no-mcmc 2>/dev/null | grep '"metric"' | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$sh', d['value'], d['ms_per_step'], json.dumps(d.get('per_rank')))"
done | tee $R/r04_two_rank_single_gpu.txt
echo "== bench.py --gpus 2 starting its own ranks (one GPU, gloo): time-sharded headline + neuron-sharded step"
python3 bench.py --gpus 2 --debug-single-device --steps 10 --warmup 3 2>/dev/null | tail -1 | tee $R/r04_self_launch_two_ranks.json | cut -c1-400
echo "== hipGraph replay against stream launches"
python3 tools/r4/graph_step.py 2>&1 | grep "^N=" | tee $R/r04_graph_replay.txt
echo "== a recording 40 times the length of C3 (126 GB of resident tiles)"
python3 tools/r4/long_recording.py 24000000 2>&1 | grep -v amdgpu.ids | tail -9 | tee $R/r04_long_recording.txt
echo "== bench.py multi-rank path on RCCL, one rank"
for sh in time neurons; do python3 bench.py --rccl-selftest --shard $sh --steps 20 --warmup 3 2>/dev/null | tail -1; done | tee $R/r04_rccl_selftest.jsonl | cut -c1-300
echo "== GPU test suite"
python3 -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $R/r04_gpu_tests.txt
rm -rf gpurun_out/pmc_C2 gpurun_out/pmc_C5 gpurun_out/prof_gibbs
ls $R
