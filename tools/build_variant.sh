#!/bin/bash
# Tuning tool: builds the library with extra compiler flags into lerc_amd/csrc/_var/<name>.so (for tools/bench_variants.sh).
#   tools/build_variant.sh d16 -DLERC_DISC_CHUNKS=16
set -e
cd "$(dirname "$0")/../lerc_amd/csrc"
NAME=$1; shift
D=_var/obj_$NAME
mkdir -p $D
FLAGS="-O3 -std=c++17 -fPIC -pthread -ffp-contract=off -fvisibility=hidden -Wno-unused-value -Wno-unused-result"
for f in tile_encode.hip tile_decode.hip misc_kernels.hip tile_fast.hip tile_fast_decode.hip tile_fast_decode_one.hip tile_fast_decode_scan.hip huffman_kernels.hip fpl_kernels.hip lerc1_kernels.hip rle_kernels.hip \
         codec_common.cpp codec_encode.cpp codec_decode.cpp huffman_host.cpp fpl_host.cpp lerc1_host.cpp gather_rccl.cpp capi.cpp; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS "$@" -c $f -o $D/${f%.*}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o _var/$NAME.so $D/*.o -ldl
rm -rf $D
echo built lerc_amd/csrc/_var/$NAME.so
