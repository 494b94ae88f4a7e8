#!/bin/bash
# PMC counters of the fused kernel of one small configuration (dev tool; run via gpurun): tools/prof_small_pmc.sh C2
CFG=${1:-C2}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$CFG
rm -rf "$OUT"; mkdir -p "$OUT"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d $OUT/p1 -- python3 tools/cfg_loop.py $CFG > $OUT/l1.txt 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY -d $OUT/p2 -- python3 tools/cfg_loop.py $CFG > $OUT/l2.txt 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/p3 -- python3 tools/cfg_loop.py $CFG > $OUT/l3.txt 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/p4 -- python3 tools/cfg_loop.py $CFG > $OUT/l4.txt 2>&1
python3 tools/rocprof_summary.py pmc $(find $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 -name '*results.db') $OUT/pmc.json
python3 - <<PY
import json
d=json.load(open('$OUT/pmc.json'))
for k,v in d.items():
    if 'k_fused' in k or 'k_finalize' in k:
        print(k[:70])
        for kk,vv in sorted(v.items()):
            if isinstance(vv,dict) and 'avg' in vv: print("   %-28s %.5g" % (kk, vv['avg']))
            elif kk == '_launch': print("   launch", vv)
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
tail -1 $OUT/l1.txt
