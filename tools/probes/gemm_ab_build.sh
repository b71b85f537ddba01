#!/bin/bash
# stand-alone GEMM libraries for tools/probes/gemm_ab.py: name=flags pairs -> tools/gemm_variants/libgemm_<name>.so (git-ignored)
set -e
cd "$(dirname "$0")/../.."
OUT=tools/gemm_variants
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function -Wno-inline-asm -Iinclude"
for spec in "$@"; do
  name=${spec%%=*}; defs=${spec#*=}
  /opt/rocm/bin/hipcc $FLAGS $defs -o $OUT/libgemm_$name.so sam6d_amd/csrc/s6d_gemm.hip sam6d_amd/csrc/s6d_capi.hip &
done
wait
ls -la $OUT
