#!/bin/bash
# Stand-alone builds of the four-wave GEMM (csrc/s6d_gemm4.hip) with other generated K-loop streams (tools/gen_gemm4_asm.py
# --variant) for tools/gemm4_ab.py variants:  tools/gemm4_variants/libg4_<name>.so  (git-ignored; travels with the gpurun snapshot)
set -e
cd "$(dirname "$0")/.."
OUT=tools/gemm4_variants
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function -Wno-inline-asm"
SRC="sam6d_amd/csrc/s6d_gemm.hip sam6d_amd/csrc/s6d_gemm4.hip sam6d_amd/csrc/s6d_capi.hip"
build() {   # name, generator variant, extra flags
  python3 tools/gen_gemm4_asm.py --variant $2 --out $OUT/g4_$2.inc > /dev/null
  if [ "$2" = buf ]; then X="-DS6D_G4_BUF"; else X=""; fi
  /opt/rocm/bin/hipcc $FLAGS -fno-slp-vectorize $X -DS6D_G4_INC="\"$PWD/$OUT/g4_$2.inc\"" $3 -o $OUT/libg4_$1.so $SRC &
}
for v in ${@:-base buf nodma noreads nobarrier mfmaonly spread dmafirst dmaearly readslate}; do build $v $v ""; done
if [ $# -eq 0 ]; then build noepi base "-DS6D_G4_NOEPI"; fi
wait
ls $OUT/*.so
