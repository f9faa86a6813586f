#!/bin/bash
# C2 bench line and C3 timing for the default build and every library given:  gpurun -- 'bash tools/gpu_c3_libs.sh lib.so ...'
set -u
for L in default "$@"; do
  E=""; [ "$L" != default ] && E="LERC_AMD_LIBRARY=$PWD/$L"
  echo "== $L"
  for rep in 1 2; do
  env $E python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c5-anchor --no-other-configs --rotate 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('  c2 ms_per_step', d['ms_per_step'], 'frac', d['roundtrip']['frac_of_hbm_peak_wall'], ' '.join(f\"{k}={v['avg_ms']*1000:.1f}\" for k,v in d['kernels'].items()))
"
  env $E timeout 300 python tools/time_configs.py c3 2>&1 | grep -v amdgpu.ids | tail -3 | tr '\n' ' '; echo
  done
done
