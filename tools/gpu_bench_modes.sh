#!/bin/bash
# bench.py in its other modes on a one-GPU box: the C5 mosaic on one GPU (with the all-cores CPU baseline).
#   DRY2=1 adds a dry run of the N = 2 path (both ranks on device 0, gloo instead of RCCL: checks the code path, not the
#   links; gloo moves device tensors slowly -- minutes)
set -u
OUT=$PWD/gpurun_out; mkdir -p "$OUT"; TAG=${1:-modes}
timeout 300 python bench.py --workload c5 --tiles 4096 --steps 5 --warmup 2 > "$OUT/${TAG}_c5_1gpu.json" 2> "$OUT/${TAG}_c5_1gpu.err"; tail -c 1500 "$OUT/${TAG}_c5_1gpu.json"; tail -3 "$OUT/${TAG}_c5_1gpu.err"
if [ "${DRY2:-0}" = "1" ]; then
  LERC_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --tiles 512 --rotate 0 > "$OUT/${TAG}_c5_2rank_dry.json" 2> "$OUT/${TAG}_c5_2rank_dry.err"; tail -c 2500 "$OUT/${TAG}_c5_2rank_dry.json"; tail -5 "$OUT/${TAG}_c5_2rank_dry.err"
fi
