#!/bin/bash
# Times bench.py's kernels for several builds of the library (LERC_AMD_LIBRARY), one line per kernel.
#   gpurun -- 'bash tools/bench_variants.sh lib1.so lib2.so ...'
for L in "$@"; do
  echo "== $L"
  LERC_AMD_LIBRARY=$PWD/$L python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c5-anchor 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'], 'verified', d['config']['verified'])
print(' '.join(f\"{k}={v['avg_ms']*1000:.1f}\" for k,v in d['kernels'].items()))
"
done
