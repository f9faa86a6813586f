#!/bin/bash
# Everything the round's profiles/ files come from, in one GPU-box call:
#   gpurun --timeout 2400 -- 'bash tools/round_refresh.sh r02'
set -u
TAG=${1:-rXX}
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > "$OUT/${TAG}_pytest_gpu.txt" 2>&1; tail -3 "$OUT/${TAG}_pytest_gpu.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/${TAG}_smoke.txt" 2>&1; tail -2 "$OUT/${TAG}_smoke.txt"
bash tools/profile_run.sh $TAG
# (the bench line quotes the traffic of the counter passes just made: it looks for them under profiles/, stamped with these sources' digest)
cp "$OUT/${TAG}_hbm_traffic.json" profiles/ 2>/dev/null
timeout 600 python bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"; cut -c1-400 "$OUT/${TAG}_bench.json"
timeout 300 python tools/time_configs.py c3 c4 general c2lossless > "$OUT/${TAG}_time_configs.txt" 2>&1
timeout 200 python tools/time_tiles.py > "$OUT/${TAG}_time_tiles.txt" 2>&1
timeout 200 python tools/time_small.py > "$OUT/${TAG}_time_small.txt" 2>&1
timeout 200 python tools/time_host_api.py > "$OUT/${TAG}_time_host_api.txt" 2>&1
timeout 200 python tools/time_ragged.py > "$OUT/${TAG}_time_ragged.txt" 2>&1
timeout 400 python bench.py --workload c5 --tiles 8192 --steps 5 --warmup 2 > "$OUT/${TAG}_bench_c5_1gpu.json" 2> "$OUT/${TAG}_bench_c5_1gpu.err"; cut -c1-300 "$OUT/${TAG}_bench_c5_1gpu.json"
ROOT=$PWD
cd /tmp && rm -rf /tmp/prof_c4
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_c4 -o kt -- python $ROOT/tools/time_configs.py c4 > /dev/null 2> "$OUT/${TAG}_c4_kt.err"
DB=$(find /tmp/prof_c4 -name '*.db' | head -1)
python "$ROOT/tools/rocpd_summary.py" "$DB" lerc > "$OUT/${TAG}_kernel_trace_c4.txt" 2>&1
cd $ROOT
bash tools/gpu_trace_config.sh general ${TAG}_masked > /dev/null 2>&1    # kernel trace of the masked raster (general path)
# per-workgroup time lines of the scanning decoder (unmasked: the headline band; masked: the block offsets) from the probe build, if it travelled
if [ -f lerc_amd/csrc/_var/trace.so ]; then
  PROBE_LIB=$PWD/lerc_amd/csrc/_var/trace.so timeout 200 python tools/trace_decode_scan.py > "$OUT/${TAG}_trace_decode_scan.txt" 2>&1
  PROBE_LIB=$PWD/lerc_amd/csrc/_var/trace.so timeout 200 python tools/trace_masked_scan.py > "$OUT/${TAG}_trace_masked_scan.txt" 2>&1
fi
timeout 200 python tools/fuzz_against_oracle.py 5701 100 > "$OUT/${TAG}_fuzz.txt" 2>&1; tail -n 2 "$OUT/${TAG}_fuzz.txt"
ls -la "$OUT" | grep $TAG | wc -l
