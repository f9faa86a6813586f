#!/bin/bash
# One GPU-box call while iterating on the scanning decoder: the GPU tests, then the C2 bench for several builds / knobs (one line per
# run), the per-workgroup time line, instruction counts.
#   gpurun --timeout 1500 -- 'bash tools/gpu_r5.sh [lib.so ...]'
set -u
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
run() {  # name, then env assignments
  local name=$1; shift
  timeout 120 env "$@" python bench.py --steps ${STEPS:-20} --warmup 5 --no-cpu-baseline --no-c5-anchor --no-other-configs --rotate 0 2>"$OUT/r5_$name.err" | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$name', 'ms_per_step', d['ms_per_step'], 'frac', d['roundtrip']['frac_of_hbm_peak_wall'], 'verified', d['config']['verified'], ' '.join(f\"{k}={v['avg_ms']*1000:.1f}\" for k,v in d['kernels'].items()))
" || tail -3 "$OUT/r5_$name.err"
}
if [ "${TESTS:-1}" = "1" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q ${PYTEST_ARGS:-} > "$OUT/r5_pytest.txt" 2>&1; tail -4 "$OUT/r5_pytest.txt"
fi
run scan LERC_AMD_DECODE_SCAN=1
run walk LERC_AMD_DECODE_SCAN=0
run scan_again LERC_AMD_DECODE_SCAN=1
for L in "$@"; do run "$(basename $L .so)" LERC_AMD_LIBRARY=$PWD/$L; done
if [ -f lerc_amd/csrc/_var/trace.so ]; then
  PROBE_LIB=$PWD/lerc_amd/csrc/_var/trace.so timeout 200 python tools/trace_decode_scan.py 2>&1 | tail -12
  PROBE_LIB=$PWD/lerc_amd/csrc/_var/trace.so timeout 200 python tools/trace_decode_scan.py c3 2>&1 | tail -12
fi
if [ "${PMC:-0}" = "1" ]; then
  ROOT=$PWD
  for L in default ${PMC_LIBS:-}; do
    LIB=""; [ "$L" != default ] && LIB="LERC_AMD_LIBRARY=$ROOT/$L"
    for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" ${PMC2:+"SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"}; do
      (cd /tmp && rm -rf /tmp/prof_pmc && TMPDIR=/tmp timeout 300 env $LIB rocprofv3 --kernel-trace --pmc $SET -d /tmp/prof_pmc -o pmc -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-c5-anchor --no-other-configs --rotate 0 > /dev/null 2> "$OUT/r5_pmc.err")
      DB=$(find /tmp/prof_pmc -name '*.db' | head -1)
      echo "== PMC $L"
      python "$ROOT/tools/rocpd_summary.py" "$DB" fast 2>&1 | grep -v "^$" | cut -c1-200 | tee -a "$OUT/r5_pmc.txt"
    done
  done
fi
timeout 300 python tools/time_configs.py c3 2>&1 | tail -8
timeout 300 python bench.py --workload c5 --tiles 8192 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | cut -c1-600
