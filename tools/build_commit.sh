#!/bin/bash
# Build the library as of a git commit for A/B runs: tools/build_commit.sh <commit> <name> [extra hipcc flags]
# -> theano_pyglm_amd/libpyglm_hip_<name>.so (use with PYGLM_HIP_LIB=...)
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
commit=$1; name=$2; shift 2
tmp=$(mktemp -d)
mkdir -p $tmp/theano_pyglm_amd/csrc $tmp/include
git -C "$root" show $commit:theano_pyglm_amd/csrc/pglm_capi.hip > $tmp/theano_pyglm_amd/csrc/pglm_capi.hip
git -C "$root" show $commit:theano_pyglm_amd/csrc/pglm_kernels.hip.h > $tmp/theano_pyglm_amd/csrc/pglm_kernels.hip.h
git -C "$root" show $commit:include/pyglm_hip.h > $tmp/include/pyglm_hip.h
(cd $tmp/theano_pyglm_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" pglm_capi.hip -o "$root/theano_pyglm_amd/libpyglm_hip_$name.so")
rm -rf $tmp
