#!/bin/bash
# One GPU-box call while iterating on the masked band's scan (MODE 1): the GPU tests, the C2 bench (the unmasked kernel shares the code),
# the per-workgroup time line of the masked scan, the masked raster's timing with the per-kernel split.
#   gpurun --timeout 1200 -- 'bash tools/gpu_masked.sh'
set -u
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
if [ "${TESTS:-1}" = "1" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q > "$OUT/r5_pytest.txt" 2>&1; tail -n 4 "$OUT/r5_pytest.txt"
fi
timeout 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c5-anchor --no-other-configs --rotate 0 2>"$OUT/r5_bench.err" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('c2 ms_per_step', d['ms_per_step'], 'frac', d['roundtrip']['frac_of_hbm_peak_wall'], 'verified', d['config']['verified'], ' '.join(f\"{k}={v['avg_ms']*1000:.1f}\" for k,v in d['kernels'].items()))
" || tail -n 3 "$OUT/r5_bench.err"
if [ -f lerc_amd/csrc/_var/trace.so ]; then
  PROBE_LIB=$PWD/lerc_amd/csrc/_var/trace.so timeout 200 python tools/trace_masked_scan.py 2>&1 | tail -n 14
fi
timeout 300 python tools/time_configs.py general 2>&1 | tail -n 16
