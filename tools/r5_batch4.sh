cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_population.py tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r5_t7.log
python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "stress or c5" 2>&1 | tail -6 >> gpurun_out/r5_t7.log
for rep in 1 2; do
echo "== merged head/tail (default)"; python tools/cfg_loop.py C5S 12 2>&1 | tail -1
echo "== separate launches (94=6)"; python tools/cfg_loop.py C5S 12 6 2>&1 | tail -1
echo "== slab form (94=4)"; python tools/cfg_loop.py C5S 12 4 2>&1 | tail -1
done
cat gpurun_out/r5_t7.log
