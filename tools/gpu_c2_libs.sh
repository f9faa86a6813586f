#!/bin/bash
# the C2 bench line, twice, for the default build and every library given:  gpurun -- 'bash tools/gpu_c2_libs.sh lib.so ...'
set -u
for rep in 1 2; do
for L in default "$@"; do
  E=""; [ "$L" != default ] && E="LERC_AMD_LIBRARY=$PWD/$L"
  env $E python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c5-anchor --no-other-configs --rotate 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$L', 'c2 ms_per_step', d['ms_per_step'], 'frac', d['roundtrip']['frac_of_hbm_peak_wall'], ' '.join(f\"{k}={v['avg_ms']*1000:.1f}\" for k,v in d['kernels'].items()))
"
done
done
