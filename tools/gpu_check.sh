#!/bin/bash
# One GPU-box call while iterating: [the gpu suite,] the bench line (C2 only), the two headline kernels' counters.
#   gpurun --timeout 1500 -- 'TESTS=1 bash tools/gpu_check.sh [lib.so ...]'
set -u
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
if [ "${TESTS:-1}" = "1" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > "$OUT/check_pytest.txt" 2>&1; tail -4 "$OUT/check_pytest.txt"
fi
run() {
  local name=$1; shift
  env "$@" timeout 300 python bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-c5-anchor --no-other-configs --rotate 0 2>"$OUT/check_$name.err" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', 'ms_per_step', d['ms_per_step'], 'frac', d['roundtrip']['frac_of_hbm_peak_wall'], 'verified', d['config']['verified'], 'ref', d['config'].get('blob_matches_reference'), ' '.join(f\"{k}={v['avg_ms']*1000:.1f}\" for k,v in d['kernels'].items()))
" | tee -a "$OUT/check_bench.txt" || tail -3 "$OUT/check_$name.err"
}
: > "$OUT/check_bench.txt"
run default A=1
run default_again A=1
for L in "$@"; do run "$(basename $L .so)" LERC_AMD_LIBRARY=$ROOT/$L; done
SETA="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
: > "$OUT/check_pmc.txt"
if [ "${PMC:-1}" = "1" ]; then
  for L in default "$@"; do
    LIBENV="A=1"; [ "$L" != default ] && LIBENV="LERC_AMD_LIBRARY=$ROOT/$L"
    rm -rf /tmp/pv && mkdir -p /tmp/pv && cd /tmp/pv
    env $LIBENV timeout 300 rocprofv3 --kernel-trace --pmc $SETA -d /tmp/pv -o t -- python $ROOT/tools/roundtrip_loop.py 6 > /tmp/pv/log.txt 2>&1
    DB=$(find /tmp/pv -name '*.db' | head -1)
    echo "== pmc $L" | tee -a "$OUT/check_pmc.txt"
    python $ROOT/tools/rocpd_summary.py "$DB" k_fast 2>&1 | grep -E "k_fast|SQ_" | sed -e 's/  */ /g' | cut -c1-120 | tee -a "$OUT/check_pmc.txt"
    cd $ROOT
  done
fi
