#!/bin/bash
# registers / scratch / LDS of a .hip file's kernels, by the compiler's own remarks:  tools/kernel_regs.sh tile_fast_decode_scan [-DFLAG ...]
cd "$(dirname "$0")/../lerc_amd/csrc"
F=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden --cuda-device-only "$@" -c $F.hip -o /dev/null -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re,sys
name=None; d={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: name=m.group(1); d[name]={}
    m=re.search(r'\b(VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)',l)
    if m and name: d[name][m.group(1).split()[0]]=int(m.group(2))
for k,v in d.items(): print(k[:70].ljust(70), v)
"
