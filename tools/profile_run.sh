#!/bin/bash
# Profiles the bench workload on the GPU box: (1) rocprofv3 --kernel-trace --stats, (2..) PMC passes (counters only,
# never combined with other trace domains).  Summaries land in gpurun_out/<tag>_*.txt; copy them to profiles/.
#   gpurun -- 'bash tools/profile_run.sh r01'
set -u
TAG=${1:-run}
OUT=$PWD/gpurun_out
mkdir -p "$OUT"
export TMPDIR=/tmp
BENCH="python $PWD/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-c5-anchor --no-other-configs --rotate 0"
cd /tmp
rm -rf /tmp/prof_kt /tmp/prof_pmc*
rocprofv3 --kernel-trace --stats -d /tmp/prof_kt -o kt -- $BENCH > "$OUT/${TAG}_bench_under_rocprof.json" 2> "$OUT/${TAG}_kt.err"
DB=$(find /tmp/prof_kt -name '*.db' | head -1)
python "$OLDPWD/tools/rocpd_summary.py" "$DB" > "$OUT/${TAG}_kernel_trace.txt" 2>&1
find /tmp/prof_kt -name '*stats*' -exec cp {} "$OUT/" \; 2>/dev/null
i=0
for SET in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $SET -d /tmp/prof_pmc$i -o pmc -- $BENCH > /dev/null 2> "$OUT/${TAG}_pmc$i.err"
  DB=$(find /tmp/prof_pmc$i -name '*.db' | head -1)
  { echo "# counters: $SET"; python "$OLDPWD/tools/rocpd_summary.py" "$DB" fast; } > "$OUT/${TAG}_pmc$i.txt" 2>&1
  [ "$SET" = "FETCH_SIZE" ] && FETCH_DB=$DB
  [ "$SET" = "WRITE_SIZE" ] && WRITE_DB=$DB
done
python "$OLDPWD/tools/rocpd_summary.py" --traffic "$FETCH_DB" "$WRITE_DB" 8192 "$OUT/${TAG}_hbm_traffic.json"
