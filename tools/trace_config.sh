#!/bin/bash
# Kernel trace (rocprofv3 --kernel-trace --stats) of tools/time_configs.py for the given configurations:
#   gpurun -- 'bash tools/trace_config.sh tag c2lossless [c4 ...]'   ->  gpurun_out/<tag>_trace.txt
set -u
TAG=$1; shift
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
HERE=$PWD
cd /tmp
rm -rf /tmp/prof_cfg
rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg -o kt -- python "$HERE/tools/time_configs.py" "$@" > "$OUT/${TAG}_trace_run.txt" 2> "$OUT/${TAG}_trace.err"
DB=$(find /tmp/prof_cfg -name '*.db' | head -1)
python "$HERE/tools/rocpd_summary.py" "$DB" > "$OUT/${TAG}_trace.txt" 2>&1
