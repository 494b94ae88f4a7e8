cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/shape_sweep.py 300 8 12 16 48 52 64 > gpurun_out/r5_sweep_new.md 2>&1
python tools/shape_sweep.py 300 --alts --ptw=2 128 160 256 > gpurun_out/r5_sweep_ptw2.md 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/r5_gibbs_prof -o g -- python3 tools/gibbs_kernel_only.py > gpurun_out/r5_gibbs_k.log 2>&1
python -m pytest tests/test_gpu_population.py tests/test_gpu_mcmc.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r5_t5.log
cat gpurun_out/r5_sweep_new.md gpurun_out/r5_sweep_ptw2.md gpurun_out/r5_t5.log
