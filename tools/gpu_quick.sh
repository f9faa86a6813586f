#!/bin/bash
# One GPU-box call while iterating: parity suite, default bench line, kernel trace of the bench.
#   gpurun --timeout 1500 -- 'bash tools/gpu_quick.sh r02a [pytest-args]'
set -u
TAG=${1:-quick}
shift || true
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
timeout 900 python -m pytest tests -m gpu -x -q "$@" > "$OUT/${TAG}_pytest_gpu.txt" 2>&1
tail -5 "$OUT/${TAG}_pytest_gpu.txt"
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
cat "$OUT/${TAG}_bench.json"; tail -3 "$OUT/${TAG}_bench.err"
cd /tmp && rm -rf /tmp/prof_kt
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_kt.err"
DB=$(find /tmp/prof_kt -name '*.db' | head -1)
python "$ROOT/tools/rocpd_summary.py" "$DB" > "$OUT/${TAG}_kernel_trace.txt" 2>&1
grep "lerc::" "$OUT/${TAG}_kernel_trace.txt" | cut -c1-60,90-160
if [ "${PMC:-0}" = "1" ]; then
  i=0
  for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"; do
    i=$((i+1))
    rm -rf /tmp/prof_pmc$i
    timeout 300 rocprofv3 --kernel-trace --pmc $SET -d /tmp/prof_pmc$i -o pmc -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2> "$OUT/${TAG}_pmc$i.err"
    DB=$(find /tmp/prof_pmc$i -name '*.db' | head -1)
    { echo "# counters: $SET"; python "$ROOT/tools/rocpd_summary.py" "$DB" fast; } > "$OUT/${TAG}_pmc$i.txt" 2>&1
    cat "$OUT/${TAG}_pmc$i.txt" | cut -c1-200
  done
fi
