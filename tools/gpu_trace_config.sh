#!/bin/bash
# Kernel trace of one of tools/time_configs.py's configurations on the GPU box:
#   gpurun --timeout 600 -- 'bash tools/gpu_trace_config.sh general gen'
set -u
WHICH=${1:-general}
TAG=${2:-$WHICH}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
timeout 300 python tools/time_configs.py "$WHICH" > "$OUT/${TAG}_time.txt" 2>&1
cat "$OUT/${TAG}_time.txt"
cd /tmp && rm -rf /tmp/prof_cfg
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg -o kt -- python $ROOT/tools/time_configs.py "$WHICH" > "$OUT/${TAG}_trace_run.txt" 2> "$OUT/${TAG}_trace.err"
DB=$(find /tmp/prof_cfg -name '*.db' | head -1)
python "$ROOT/tools/rocpd_summary.py" "$DB" > "$OUT/${TAG}_trace.txt" 2>&1
grep "lerc::\|rocclr" "$OUT/${TAG}_trace.txt" | cut -c1-70,90-160
