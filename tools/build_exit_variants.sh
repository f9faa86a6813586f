#!/bin/bash
# Tuning tool: the two headline kernels built to leave at mark n (LERC_ENC_EXIT / LERC_DEC_EXIT), one library per mark, linked
# against the default build's other objects -> lerc_amd/csrc/_var/enc_exit<n>.so, dec_exit<n>.so  (read by tools/pmc_exits.sh)
set -e
cd "$(dirname "$0")/../lerc_amd/csrc"
make -s -j8
mkdir -p _var/obj_exit
FLAGS="-O3 -std=c++17 -fPIC -pthread -ffp-contract=off -fvisibility=hidden -Wno-unused-value -Wno-unused-result"
build() {  # file macro n name
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS -D$2=$3 -c $1.hip -o _var/obj_exit/$4.o 2>/dev/null
  OTHERS=$(ls *.o | grep -v "^$1.o$")
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o _var/$4.so _var/obj_exit/$4.o $OTHERS -ldl
  echo built _var/$4.so
}
N=0
for n in ${ENC_MARKS:-1 2 3 4 5}; do build tile_fast LERC_ENC_EXIT $n enc_exit$n & N=$((N+1)); [ $((N % 4)) = 0 ] && wait; done
for n in ${DEC_MARKS:-0 1 2 3 4 5}; do build tile_fast_decode_scan LERC_DEC_EXIT $n dec_exit$n & N=$((N+1)); [ $((N % 4)) = 0 ] && wait; done
wait
rm -rf _var/obj_exit
