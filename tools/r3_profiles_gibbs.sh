#!/bin/bash
# The Gibbs part of tools/r3_profiles.sh alone (after a change to the Gibbs kernels only).
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
R=gpurun_out/r03g; rm -rf $R; mkdir -p $R
bash tools/prof_gibbs.sh > $R/prof_gibbs.log 2>&1; cp gpurun_out/prof_gibbs/stats.csv $R/r03_gibbs_kernel_stats.csv; cp gpurun_out/prof_gibbs/pmc.json $R/r03_gibbs_pmc.json; head -6 $R/r03_gibbs_kernel_stats.csv
python3 tools/gibbs_kernel_only.py 2>&1 | tail -5 | tee $R/r03_gibbs_launch.txt
python3 tools/gibbs_ablate.py 2>&1 | grep "^dbg" | tee $R/r03_gibbs_ablation.txt
python3 tools/gibbs_sweep_profile.py 2>&1 | tail -7 | tee $R/r03_gibbs_sweep.txt
./tools/ubench/occ_gibbs_ubench 2>&1 | grep "workgroups\|shared" > $R/r03_gibbs_occupancy.txt
rm -rf gpurun_out/prof_gibbs
