#!/bin/bash
# Timing-only ablation builds of geo_embed2_kernel (csrc/s6d_geo.hip, S6D_G2_ABL bits: 1 no weight staging in the loop, 2 no sinusoid
# fragments, 4 every output tile reads the first tile's weight fragments (LDS reads stay, compiler may merge them)) and their
# times beside the product build: bash tools/probes/geo_variants.sh
set -e
cd "$(dirname "$0")/../.."
OUT=tools/probes/geo_variants
mkdir -p $OUT
FLAGS="-O3 -std=c++17 -fPIC -shared --offload-arch=gfx950 -mllvm -amdgpu-mfma-vgpr-form -Wno-unused-function"
for a in ${@:-0 1 2 3 4 7}; do
  /opt/rocm/bin/hipcc $FLAGS -DS6D_G2_ABL=$a -o $OUT/libgeo_$a.so sam6d_amd/csrc/s6d_geo.hip sam6d_amd/csrc/s6d_capi.hip &
done
wait
python tools/probes/geo_variants.py $OUT
