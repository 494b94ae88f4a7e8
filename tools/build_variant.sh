#!/bin/bash
# Build a variant of the library for A/B runs: tools/build_variant.sh <name> [extra hipcc flags...]
# -> theano_pyglm_amd/libpyglm_hip_<name>.so (use with PYGLM_HIP_LIB=...)
set -e
cd "$(dirname "$0")/../theano_pyglm_amd/csrc"
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared "$@" pglm_capi.hip -o ../libpyglm_hip_$name.so
