#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python3 -c "import __graft_entry__ as g; g.build_hip()"
OUT=gpurun_out/prof_gibbs
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -d $OUT/trace -- python3 tools/gibbs_kernel_only.py > $OUT/log1.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_SALU -d $OUT/pmc1 -- python3 tools/gibbs_kernel_only.py > $OUT/log2.txt 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM -d $OUT/pmc2 -- python3 tools/gibbs_kernel_only.py > $OUT/log3.txt 2>&1
python3 tools/rocprof_summary.py stats $(find $OUT/trace -name '*results.db' | head -1) $OUT/stats.csv
python3 tools/rocprof_summary.py pmc $(find $OUT/pmc1 $OUT/pmc2 -name '*results.db') $OUT/pmc.json
head -5 $OUT/stats.csv
python3 - <<PY
import json
d=json.load(open('$OUT/pmc.json'))
for k,v in d.items():
    if 'gibbs_' in k: print(k, {kk:(vv.get('avg') if isinstance(vv,dict) and 'avg' in vv else vv) for kk,vv in v.items()})
PY
tail -3 $OUT/log3.txt
rm -rf $OUT/trace $OUT/pmc1 $OUT/pmc2
