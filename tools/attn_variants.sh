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
build nosoftmax "-DS6D_ATTN_ABLATE=4"
build w4s2 "-DS6D_G64_WAVES=4 -DS6D_G64_SLOTS=2"
build staticprio "-DS6D_G64_STATIC_PRIO=1"
build timing "-DS6D_G64_TIMING=1"
wait
# the register-staged kernel on the 64 x 64 grid: as it was in round 1, with the layout switches, and with 8 waves
build old_r1 "-DS6D_GLB64_DEFAULT_IMPL=1 -DS6D_GLB_KSWZ=0 -DS6D_GLB_THLD=64 -DS6D_GLB_PRIO=0"
build old_all "-DS6D_GLB64_DEFAULT_IMPL=1"
build old_waves8 "-DS6D_GLB64_DEFAULT_IMPL=1 -DS6D_GLB_WAVES=8"
for x in "$@"; do build "${x%%:*}" "${x#*:}"; done
wait
ls $OUT
