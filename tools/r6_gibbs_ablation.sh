#!/bin/bash
# k_gibbs_rate_cols: is the 29 % LDS bank-conflict share of its table gather on the critical path?  Same launches with the
# product library and with a timing-only variant whose table reads are lane-linear (conflict-free; wrong results):
# build first: tools/build_variant.sh sptlin -DPGL_SPT_LINEAR
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
R=gpurun_out/r6; mkdir -p $R
for v in "" _sptlin; do
  L=theano_pyglm_amd/libpyglm_hip$v.so
  export PYGLM_HIP_LIB=$GRAFT_REPO_ROOT/$L
  rocprofv3 --kernel-trace --stats -d $R/tr_g$v -- python3 tools/gibbs_kernel_only.py > $R/gibbs_ab$v.log 2>&1
  python3 tools/rocprof_summary.py stats "$(find $R/tr_g$v -name '*results.db' | head -1)" $R/gibbs_ab_stats$v.csv; rm -rf $R/tr_g$v
  echo "== $L"; grep "k_gibbs_rate_cols\|k_gibbs_spike" $R/gibbs_ab_stats$v.csv | cut -c1-120
  rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU -d $R/pm_g$v -- python3 tools/gibbs_kernel_only.py > /dev/null 2>&1
  python3 tools/rocprof_summary.py pmc $(find $R/pm_g$v -name '*results.db') $R/gibbs_ab_pmc$v.json; rm -rf $R/pm_g$v
  python3 - <<PY
import json
d=json.load(open('$R/gibbs_ab_pmc$v.json'))
for k,v in d.items():
    if 'gibbs_rate' in k: print({kk:vv.get('avg') for kk,vv in v.items() if isinstance(vv,dict) and 'avg' in vv})
PY
done
