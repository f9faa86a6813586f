#!/bin/bash
# Kernel trace of the C4 configuration (Huffman byte path): gpurun --timeout 600 -- 'bash tools/gpu_c4.sh tag'
set -u
TAG=${1:-c4}
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
timeout 600 python -m pytest tests -m gpu -x -q -k "huff or c4 or lossless or golden or fuzz" > "$OUT/${TAG}_pytest.txt" 2>&1; tail -3 "$OUT/${TAG}_pytest.txt"
timeout 300 python tools/time_configs.py c4 > "$OUT/${TAG}_c4.txt" 2>&1; cat "$OUT/${TAG}_c4.txt"
cd /tmp && rm -rf /tmp/prof_c4
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o kt -- python $ROOT/tools/time_configs.py c4 > /dev/null 2> "$OUT/${TAG}_c4_kt.err"
DB=$(find /tmp/prof_c4 -name '*.db' | head -1)
python "$ROOT/tools/rocpd_summary.py" "$DB" lerc > "$OUT/${TAG}_c4_trace.txt" 2>&1
cut -c1-70,90-160 "$OUT/${TAG}_c4_trace.txt"
