#!/bin/bash
# PMC passes over one of tools/time_configs.py's configurations, for the kernels whose name holds $3:
#   gpurun --timeout 600 -- 'bash tools/gpu_pmc_config.sh general pmcg rank'
set -u
WHICH=${1:-general}; TAG=${2:-pmc}; FLT=${3:-lerc}
OUT=$PWD/gpurun_out; mkdir -p "$OUT"
export TMPDIR=/tmp
ROOT=$PWD
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  cd /tmp && rm -rf /tmp/prof_pc$i
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d /tmp/prof_pc$i -o pmc -- python $ROOT/tools/time_configs.py "$WHICH" > /dev/null 2> "$OUT/${TAG}_$i.err"
  DB=$(find /tmp/prof_pc$i -name '*.db' | head -1)
  { echo "# counters: $SET"; python "$ROOT/tools/rocpd_summary.py" "$DB" "$FLT"; } > "$OUT/${TAG}_$i.txt" 2>&1
  cut -c1-220 "$OUT/${TAG}_$i.txt"
  cd $ROOT
done
