cd $GRAFT_REPO_ROOT
python tools/wide_check.py 2>&1 | grep -v amdgpu.ids
python tools/shape_sweep.py 300 128 160 256 2>&1 | grep -v amdgpu.ids
python -m pytest tests/test_gpu_kernels.py -x -q -m gpu 2>&1 | tail -4
