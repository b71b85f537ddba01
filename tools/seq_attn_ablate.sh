#!/bin/bash
# Where the time of the sequence-attention kernel goes: stand-alone builds of csrc/s6d_attn.hip under the compile-time ablation
# switches (S6D_ATTN_ABLATE: 1 no K / V loads, 2 no tile arithmetic, 4 no softmax arithmetic, 8 no PV, 16 no QK^T), each timed at the
# DINOv2 shape through its own C ABI (tools/seq_attn_ablate.py).  Build here (no GPU needed), run on the GPU box.
set -e
cd "$(dirname "$0")/.."
OUT=tools/attn_variants
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function -Wno-inline-asm"
build() { /opt/rocm/bin/hipcc $FLAGS $2 -o $OUT/libattn_$1.so sam6d_amd/csrc/s6d_attn.hip sam6d_amd/csrc/s6d_capi.hip & }
build seq_base ""
build seq_noload "-DS6D_ATTN_ABLATE=1"
build seq_nomath "-DS6D_ATTN_ABLATE=2"
build seq_nosoftmax "-DS6D_ATTN_ABLATE=4"
wait
build seq_nopv "-DS6D_ATTN_ABLATE=8"
build seq_noqk "-DS6D_ATTN_ABLATE=16"
build seq_mfmaonly "-DS6D_ATTN_ABLATE=4"
build seq_softmaxonly "-DS6D_ATTN_ABLATE=24"
wait
ls $OUT
