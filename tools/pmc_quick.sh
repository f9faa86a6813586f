#!/bin/bash
# Tuning tool: two PMC passes (instruction counts, activity / waits) over the default bench loop, kernels matching $1 (default "fast").
#   gpurun --timeout 600 -- 'bash tools/pmc_quick.sh fast'
export TMPDIR=/tmp
ROOT=$PWD
PAT=${1:-fast}
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rm -rf /tmp/prof_pmc$i
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d /tmp/prof_pmc$i -o pmc -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-c5-anchor --rotate 0 > /dev/null 2> /tmp/pmc$i.err
  DB=$(find /tmp/prof_pmc$i -name '*.db' | head -1)
  { echo "# counters: $SET"; python "$ROOT/tools/rocpd_summary.py" "$DB" $PAT; } 2>&1 | cut -c1-160
  cd $ROOT
done
