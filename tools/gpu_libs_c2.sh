#!/bin/bash
# C2 bench, one line per library given (and the default one first, last again): builds / knobs side by side on one box.
#   gpurun --timeout 900 -- 'bash tools/gpu_libs_c2.sh lerc_amd/csrc/_var/a.so ...'
set -u
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
run() {
  local name=$1; shift
  timeout 120 env "$@" python bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-c5-anchor --no-other-configs --rotate 0 2>"$OUT/libs_$name.err" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', 'ms_per_step', d['ms_per_step'], 'frac', d['roundtrip']['frac_of_hbm_peak_wall'], 'verified', d['config']['verified'], ' '.join(f\"{k}={v['avg_ms']*1000:.1f}\" for k,v in d['kernels'].items()))
" || tail -n 3 "$OUT/libs_$name.err"
}
run default A=1
for L in "$@"; do run "$(basename $L .so)" LERC_AMD_LIBRARY=$PWD/$L; done
run default_again A=1
for L in "$@"; do run "$(basename $L .so)_again" LERC_AMD_LIBRARY=$PWD/$L; done
