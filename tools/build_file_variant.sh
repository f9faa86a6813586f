#!/bin/bash
# Tuning tool: ONE source file built with extra flags, linked against the default build's other objects -> lerc_amd/csrc/_var/<name>.so
#   tools/build_file_variant.sh dec_plain tile_fast_decode_scan -DLERC_DECODE_PLAIN_STORES
set -e
cd "$(dirname "$0")/../lerc_amd/csrc"
NAME=$1; FILE=$2; shift 2
mkdir -p _var/obj_$NAME
FLAGS="-O3 -std=c++17 -fPIC -pthread -ffp-contract=off -fvisibility=hidden -Wno-unused-value -Wno-unused-result"
/opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS "$@" -c $FILE.hip -o _var/obj_$NAME/$FILE.o 2>/dev/null
OTHERS=$(ls *.o | grep -v "^$FILE.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o _var/$NAME.so _var/obj_$NAME/$FILE.o $OTHERS -ldl
rm -rf _var/obj_$NAME
echo built lerc_amd/csrc/_var/$NAME.so
