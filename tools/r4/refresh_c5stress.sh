#!/bin/bash
# The C5-stress part of tools/r4_profiles.sh alone (after a change to the separable-stimulus kernels only).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
R=gpurun_out/r04; mkdir -p $R
rocprofv3 --kernel-trace --stats -d $R/trace_c5s -- python3 tools/cfg_loop.py C5S 12 > $R/c5s_trace.log 2>&1
T=$(find $R/trace_c5s -name '*results.db' | head -1)
python3 tools/rocprof_summary.py stats "$T" $R/r04_C5stress_kernel_stats.csv
python3 tools/rocprof_summary.py timeline "$T" $R/r04_C5stress_timeline.csv 18; rm -rf $R/trace_c5s
head -14 $R/r04_C5stress_kernel_stats.csv | cut -c1-70,120-300 | cut -c1-160
bash tools/prof_small_pmc.sh C5S > $R/pmc_C5S.log 2>&1; cp gpurun_out/pmc_C5S/pmc.json $R/r04_pmc_C5stress.json; rm -rf gpurun_out/pmc_C5S
python3 bench.py --steps 5 --warmup 2 --no-map --no-mcmc --no-cpu-baseline --no-ab --stim-map 2>/dev/null | tail -1 | python3 -c "import sys, json; d = json.loads(sys.stdin.read())['secondary_stim']; print(json.dumps(d, indent=1))" | tee $R/r04_C5stress_bench_block.json | grep -E '"value"|map_sweep|iterations|evaluations'
CFG_ONLY="C5" python3 tools/config_table.py 2>&1 | grep "^|" | tee $R/r04_config_table_C5.md
