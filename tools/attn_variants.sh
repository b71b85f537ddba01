#!/bin/bash
# Stand-alone builds of the attention kernels with the profiling / layout switches of csrc/s6d_attn.hip (header of that file) for
# tools/attn_time.py:   tools/attn_variants/libattn_<name>.so   (git-ignored; travels with the gpurun snapshot)
set -e
cd "$(dirname "$0")/.."
OUT=tools/attn_variants
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function -Wno-inline-asm"
build() { /opt/rocm/bin/hipcc $FLAGS $2 -o $OUT/libattn_$1.so sam6d_amd/csrc/s6d_attn.hip sam6d_amd/csrc/s6d_capi.hip & }
build base ""
build noload "-DS6D_ATTN_ABLATE=1"
build nomath "-DS6D_ATTN_ABLATE=2"
build timing "-DS6D_G64_TIMING=1"
# round 5: the running-maximum arithmetic of round 4 on the global kernel, the exact two-pass window pass, the window kernel without its K chunk swizzle
build g64_max "-DS6D_G64_NOMAX=0"
build win_exact "-DS6D_WIN16_STREAM=0"
build win_noswz "-DS6D_WIN16_KSWZ=0"
build g64_w4s2 "-DS6D_G64_WAVES=4 -DS6D_G64_SLOTS=2"
for x in "$@"; do build "${x%%:*}" "${x#*:}"; done
wait
ls $OUT
