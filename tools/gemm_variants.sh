#!/bin/bash
# Stand-alone builds of the GEMM kernel with profiling switches (csrc/s6d_gemm.hip header) for tools/gemm_time.py:
#   tools/gemm_variants/libgemm_<name>.so   (git-ignored; travels with the gpurun snapshot)
set -e
cd "$(dirname "$0")/.."
OUT=tools/gemm_variants
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function"
build() { /opt/rocm/bin/hipcc $FLAGS $2 -o $OUT/libgemm_$1.so sam6d_amd/csrc/s6d_gemm.hip sam6d_amd/csrc/s6d_capi.hip & }
build base ""
build ph2 "-DS6D_GEMM_PH2=1"
build ph2noprio "-DS6D_GEMM_PH2=1 -DS6D_GEMM_NOPRIO"
build nostore "-DS6D_GEMM_ABLATE=4"
build ph2nostore "-DS6D_GEMM_PH2=1 -DS6D_GEMM_ABLATE=4"
wait
ls -la $OUT
