#!/bin/bash
# Gibbs part of tools/r4_profiles.sh alone (after a change to the Gibbs kernels) + a fresh default bench line.
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
R=gpurun_out/r04g; rm -rf $R; mkdir -p $R
L=$PWD/theano_pyglm_amd
( time python3 -c "import torch" ) 2>&1 | grep real
bash tools/prof_gibbs.sh > $R/prof_gibbs.log 2>&1; cp gpurun_out/prof_gibbs/stats.csv $R/r04_gibbs_kernel_stats.csv; cp gpurun_out/prof_gibbs/pmc.json $R/r04_gibbs_pmc.json; head -5 $R/r04_gibbs_kernel_stats.csv
python3 tools/gibbs_kernel_only.py 2>&1 | tail -5 | tee $R/r04_gibbs_launch.txt
PYGLM_HIP_LIB=$L/libpyglm_hip_ablate.so python3 tools/gibbs_ablate.py 2>&1 | grep "^dbg" | tee $R/r04_gibbs_ablation.txt
python3 tools/gibbs_sweep_profile.py 2>&1 | tail -7 | tee $R/r04_gibbs_sweep.txt
python3 tools/r4/gibbs_nloop_scan.py 2>&1 | grep nloop | tee $R/r04_gibbs_nloop_scan.txt
./tools/ubench/occ_gibbs_ubench 2>&1 | grep "workgroups\|shared" | tee $R/r04_gibbs_occupancy.txt
python3 bench.py 2>/dev/null | tail -1 > $R/r04_bench_default_line.json; cut -c1-200 $R/r04_bench_default_line.json
rm -rf gpurun_out/prof_gibbs
