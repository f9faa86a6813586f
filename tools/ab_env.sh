#!/bin/bash
# A/B of environment settings on one box: the C2 bench line alternately with each setting, N rounds.
#   gpurun -- 'bash tools/ab_env.sh 3 A=1 LERC_AMD_SCAN_GRID=0'
set -u
ROOT=$PWD; OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
N=$1; shift
: > "$OUT/ab_env.txt"
for r in $(seq 1 $N); do
  for E in "$@"; do
    env $(echo $E | tr ',' ' ') timeout 300 python bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-c5-anchor ${OTHERS:---no-other-configs} --rotate 0 2>"$OUT/ab_env.err" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
s='$E ms_per_step %s frac %s verified %s %s' % (d['ms_per_step'], d['roundtrip']['frac_of_hbm_peak_wall'], d['config']['verified'], ' '.join(f\"{k}={v['avg_ms']*1000:.1f}\" for k,v in d['kernels'].items()))
for k in ('c3','c4','c2_masked','c2_ragged','c2_flat'):
    if k in d: s += ' | %s %s q %s' % (k, d[k].get('ms_per_step'), d[k].get('queued',{}).get('ms_per_step'))
print(s)
" | tee -a "$OUT/ab_env.txt" || tail -3 "$OUT/ab_env.err"
  done
done
