#!/bin/bash
# Stand-alone builds of the GEMM kernel with profiling switches (csrc/s6d_gemm.hip header) for tools/gemm_time.py:
#   tools/gemm_variants/libgemm_<name>.so   (git-ignored; travels with the gpurun snapshot)
set -e
cd "$(dirname "$0")/.."
OUT=tools/gemm_variants
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function -Wno-inline-asm"
build() { /opt/rocm/bin/hipcc $FLAGS $2 -o $OUT/libgemm_$1.so sam6d_amd/csrc/s6d_gemm.hip sam6d_amd/csrc/s6d_capi.hip & }
build base ""
build nomfma "-DS6D_GEMM_ABLATE=2"
build lds_stores "-DS6D_GEMM_ABLATE=3"
build dma_lds "-DS6D_GEMM_ABLATE=6"
build ldsonly "-DS6D_GEMM_ABLATE=7"
build mfmaonly "-DS6D_GEMM_ABLATE=5"
wait
ls -la $OUT
