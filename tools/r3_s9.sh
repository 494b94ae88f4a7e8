#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/s9; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -x -k "map or MAP or lockstep or population or parallel or mcmc or c1 or structured" > $OUT/pytest.log 2>&1; tail -5 $OUT/pytest.log; grep -E "^E  " $OUT/pytest.log | head
echo "== MAP bench"
timeout 600 python tools/map_bench.py 128 600 default 2>&1 | tail -4
timeout 600 python tools/map_bench.py 32 300 default 2>&1 | tail -4
