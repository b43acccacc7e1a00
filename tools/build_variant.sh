#!/bin/bash
# builds deltarice_amd/variants/lib_<name>.so with extra -D flags: A/B of kernel variants on one box via DRX_LIB_PATH
# usage: tools/build_variant.sh name -DFOO=1 ...
set -e
name=$1; shift
mkdir -p deltarice_amd/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function "$@" -shared \
  deltarice_amd/csrc/drx_encode_kernels.hip deltarice_amd/csrc/drx_encode_stream.hip deltarice_amd/csrc/drx_decode_kernels.hip deltarice_amd/csrc/drx_blocks.hip deltarice_amd/csrc/drx_pieces.hip deltarice_amd/csrc/drx_iir.hip deltarice_amd/csrc/drx_api.hip -o deltarice_amd/variants/lib_$name.so
