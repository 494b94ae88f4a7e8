#!/bin/bash
# Wait / stall counters of the bench kernels (dev tool; run via gpurun)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
python3 -c "import __graft_entry__ as g; g.build_hip()"
OUT=gpurun_out/prof_wait
rm -rf "$OUT"; mkdir -p "$OUT"
A="--steps 5 --warmup 1 --no-cpu-baseline --no-map --no-mcmc"
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/p1 -- python3 bench.py $A > $OUT/l1.txt 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_INST_CYCLES_SMEM SQ_IFETCH -d $OUT/p2 -- python3 bench.py $A > $OUT/l2.txt 2>&1
rocprofv3 --pmc SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_COEXEC_CYCLES -d $OUT/p3 -- python3 bench.py $A > $OUT/l3.txt 2>&1
rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM -d $OUT/p4 -- python3 bench.py $A > $OUT/l4.txt 2>&1
python3 tools/rocprof_summary.py pmc $(find $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4 -name '*results.db') $OUT/pmc.json
python3 - <<PY
import json
d=json.load(open('$OUT/pmc.json'))
for k,v in d.items():
    if 'k_fused5' in k:
        print(k)
        for kk,vv in sorted(v.items()):
            if isinstance(vv,dict) and 'avg' in vv: print("   %-32s %.4g" % (kk, vv['avg']))
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
