#!/bin/bash
# Round-6 evidence run (via gpurun): every summary that DESIGN.md / README.md cite, into gpurun_out/r06/ (copied to profiles/
# afterwards).  Sections can be selected: tools/r6_profiles.sh [bench configs pmc stress gibbs map sweep shards ranks tests]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
R=gpurun_out/r06; mkdir -p $R profiles
SEL=" ${*:-bench configs pmc stress gibbs map sweep shards ranks cold tests} "
python3 -c "import __graft_entry__ as g; g.build_hip(); g.build_oracle()"
has() { [[ "$SEL" == *" $1 "* ]]; }
if has bench; then
echo "== bench under rocprofv3 (kernel trace + PMC passes), then the default line un-profiled"
bash tools/profile_bench.sh r06 --no-stim > $R/profile_bench.log 2>&1; cp profiles/r06_* $R/ 2>/dev/null; tail -3 $R/profile_bench.log
python3 bench.py 2>/dev/null | tail -1 > $R/r06_bench_default_line.json; cut -c1-600 $R/r06_bench_default_line.json
fi
if has configs; then
echo "== config table"
bash tools/profile_configs.sh r06 > $R/profile_configs.log 2>&1; cp profiles/r06_config* $R/ 2>/dev/null; cat profiles/r06_config_table.md
fi
if has pmc; then
echo "== PMC of the small configurations: MFMA-busy + f64-VALU-busy per SIMD"
for c in C2 C5; do bash tools/prof_small_pmc.sh $c > $R/pmc_$c.log 2>&1; cp gpurun_out/pmc_$c/pmc.json $R/r06_pmc_$c.json; rm -rf gpurun_out/pmc_$c; done
python3 tools/pipe_busy.py $R/r06_pmc_C2.json $R/r06_pmc_C5.json | tee $R/r06_pipe_busy.txt
fi
if has stress; then
echo "== C5 stress variant (separable stimulus at the frame rate): kernel trace, PMC, A/B of the backward forms, MAP sweep"
rocprofv3 --kernel-trace --stats -d $R/trace_c5s -- python3 tools/cfg_loop.py C5S 12 > $R/c5s_trace.log 2>&1
T=$(find $R/trace_c5s -name '*results.db' | head -1)
python3 tools/rocprof_summary.py stats "$T" $R/r06_C5stress_kernel_stats.csv
python3 tools/rocprof_summary.py timeline "$T" $R/r06_C5stress_timeline.csv 14; rm -rf $R/trace_c5s
head -12 $R/r06_C5stress_kernel_stats.csv
bash tools/prof_small_pmc.sh C5S > $R/pmc_C5S.log 2>&1; cp gpurun_out/pmc_C5S/pmc.json $R/r06_pmc_C5stress.json; rm -rf gpurun_out/pmc_C5S
python3 tools/pipe_busy.py $R/r06_pmc_C5stress.json | tee -a $R/r06_pipe_busy.txt
(for rep in 1 2; do echo "fused forward + backward (default)"; python3 tools/cfg_loop.py C5S 12 2>&1 | tail -1; echo "residual slab + k_sepf_bwd (option 94 = 4)"; python3 tools/cfg_loop.py C5S 12 4 2>&1 | tail -1; done) | tee $R/r06_C5stress_backward_ab.txt
python3 tools/stress_map.py --maxiter 225 --reps 2 --scipy 0,21,42,63 2>&1 | grep -v amdgpu.ids | tee $R/r06_C5stress_map.txt | cut -c1-400
rocprofv3 --kernel-trace --stats -d $R/trace_c5m -- python3 tools/stress_map.py --maxiter 225 --reps 2 > $R/c5m_trace.log 2>&1
python3 tools/rocprof_summary.py stats "$(find $R/trace_c5m -name '*results.db' | head -1)" $R/r06_C5stress_map_kernel_stats.csv; rm -rf $R/trace_c5m
head -14 $R/r06_C5stress_map_kernel_stats.csv
fi
if has gibbs; then
echo "== Gibbs"
bash tools/prof_gibbs.sh > $R/prof_gibbs.log 2>&1; cp gpurun_out/prof_gibbs/stats.csv $R/r06_gibbs_kernel_stats.csv; cp gpurun_out/prof_gibbs/pmc.json $R/r06_gibbs_pmc.json; head -5 $R/r06_gibbs_kernel_stats.csv
python3 tools/pipe_busy.py --gibbs $R/r06_gibbs_pmc.json | tee $R/r06_gibbs_counters.txt
python3 tools/gibbs_kernel_only.py 2>&1 | tail -5 | tee $R/r06_gibbs_launch.txt
python3 tools/gibbs_sweep_profile.py 2>&1 | tail -7 | tee $R/r06_gibbs_sweep.txt
rm -rf gpurun_out/prof_gibbs
fi
if has map; then
echo "== MAP"
python3 tools/map_bench.py 128 600 default 2>&1 | tail -4 | tee $R/r06_map.txt | cut -c1-500
python3 tools/map_bench.py 32 300 default 2>&1 | tail -4 | tee -a $R/r06_map.txt | cut -c1-500
python3 tools/map_bench.py 128 600 seq 2>&1 | tail -2 | head -1 | tee -a $R/r06_map.txt | cut -c1-300
rocprofv3 --kernel-trace --stats -d $R/trace2 -- python3 tools/map_bench.py 128 600 default > $R/map_trace.log 2>&1
python3 tools/rocprof_summary.py stats "$(find $R/trace2 -name '*results.db' | head -1)" $R/r06_map_kernel_stats.csv; rm -rf $R/trace2
head -14 $R/r06_map_kernel_stats.csv
# per-launch view: kernel statistics and the dispatch timeline (gaps between dependent kernels) of the C2 and C3 sweeps
bash tools/r6_map_timeline.sh r06 > $R/map_timeline.log 2>&1
for c in C2 C3 C5stress; do cp gpurun_out/r6/map_${c}_kernel_stats_r06.csv $R/r06_map_${c}_kernel_stats.csv; cp gpurun_out/r6/map_${c}_timeline_r06.csv $R/r06_map_${c}_timeline.csv; done
tail -24 $R/r06_map_C2_timeline.csv
fi
if has sweep; then
echo "== shape sweep"
python3 tools/shape_sweep.py 300 16 32 48 64 80 96 128 144 160 192 200 256 320 384 512 2>&1 | grep "^|" | tee $R/r06_shape_sweep.md
(echo "# wide populations with a light last post block on the chunk-major grid (dev option 91 = 1)"; python3 tools/shape_sweep.py 300 --chunk-major 144 160 192 320 2>&1 | grep "^|") | tee $R/r06_shape_sweep_chunk_major.md
(echo "# the same launches without helper waves (dev option 92 = 1): blocks of five / six post tiles"; python3 tools/shape_sweep.py 300 --no-helpers 80 96 2>&1 | grep "^|") | tee $R/r06_shape_sweep_no_helpers.md
fi
if has shards; then
echo "== time shards and neuron shards on one GPU"
(echo "# HIP events around every evaluation"; python3 tools/shard_step_bench.py 1 2 4 8 2>&1 | grep "^G="; echo "# no events (the product path)"; TIMING=0 python3 tools/shard_step_bench.py 1 2 4 8 2>&1 | grep "^G="; echo "# neuron shards (north star's split): 64 / 32 / 16 neurons of C3 against the whole feature row"; python3 tools/narrow_shard.py 64 32 16 2>&1 | grep "^T=600" | cut -c1-200; echo "# the 16-neuron shard with f32 resident blocks (PGL_OPT_FEATURE_F32 = 2, opt-in; deviation from the whole-population evaluation at T = 60 s)"; python3 tools/narrow_shard.py --f32 16 2>&1 | grep "^T=\|vs the" | cut -c1-200) | tee $R/r06_shard_steps.txt
fi
if has ranks; then
echo "== bench.py --gpus N starting its own ranks on the one GPU (gloo)"
python3 bench.py --gpus 2 --debug-single-device --steps 10 --warmup 3 2>/dev/null | tail -1 | tee $R/r06_self_launch_two_ranks.json | cut -c1-300
python3 bench.py --gpus 8 --debug-single-device --steps 4 --warmup 1 --neurons 100 --seconds 30 2>/dev/null | tail -1 | tee $R/r06_self_launch_eight_ranks_uneven.json | cut -c1-300
for sh in time neurons; do python3 bench.py --rccl-selftest --shard $sh --steps 20 --warmup 3 2>/dev/null | tail -1; done | tee $R/r06_rccl_selftest.jsonl | cut -c1-300
fi
if has cold; then
echo "== first sweep of a process (torch imported and initialised, as in bench.py)"
(python3 tools/cold_map.py C3 --no-profile; python3 tools/cold_map.py C2 --no-profile; python3 tools/cold_map.py stress --no-profile) 2>&1 | grep -v amdgpu.ids | tee $R/r06_cold_map.txt
python3 tools/cold_map.py stress 2>&1 | grep -v amdgpu.ids | grep -A 26 "cProfile of sweep 0" | cut -c1-160 > $R/r06_cold_map_stress_cprofile.txt
fi
if has tests; then
echo "== GPU test suite"
python3 -m pytest tests -m gpu -q 2>&1 | tail -3 | tee $R/r06_gpu_tests.txt
fi
ls $R
