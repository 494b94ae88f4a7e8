cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "" pw8 sb0 pw8sb0; do
  if [ -z "$v" ]; then unset PYGLM_HIP_LIB; else export PYGLM_HIP_LIB=$GRAFT_REPO_ROOT/theano_pyglm_amd/libpyglm_hip_$v.so; fi
  echo "== variant '$v' fused fwd+bwd"; python tools/cfg_loop.py C5S 12 2>&1 | tail -1
done
unset PYGLM_HIP_LIB
echo "== default lib, slab form (94=4)"; python tools/cfg_loop.py C5S 12 4 2>&1 | tail -1
done
