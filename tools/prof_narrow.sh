#!/bin/bash
# rocprofv3 kernel trace + HBM fetch counter of tools/narrow_shard_bench.py -> gpurun_out/r03/r03_narrow_shard_*
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
R=gpurun_out/r03; mkdir -p $R
python3 tools/narrow_shard_bench.py 2>&1 | grep -v amdgpu | tee $R/r03_narrow_shard.txt
rocprofv3 --kernel-trace --stats -d $R/tn -- python3 tools/narrow_shard_bench.py > /dev/null 2>&1
python3 tools/rocprof_summary.py stats $(find $R/tn -name '*results.db' | head -1) $R/r03_narrow_shard_kernel_stats.csv; head -8 $R/r03_narrow_shard_kernel_stats.csv | cut -c1-120
rocprofv3 --pmc FETCH_SIZE -d $R/tp -- python3 tools/narrow_shard_bench.py > /dev/null 2>&1
python3 tools/rocprof_summary.py pmc $(find $R/tp -name '*results.db') $R/r03_narrow_shard_pmc.json
python3 - <<PY
import json
d=json.load(open('$R/r03_narrow_shard_pmc.json'))
for k,v in d.items():
    if 'fused' in k: print(k[:60], {kk:(vv.get('avg') if isinstance(vv,dict) and 'avg' in vv else vv) for kk,vv in v.items() if kk!='_launch'})
PY
rm -rf $R/tn $R/tp
