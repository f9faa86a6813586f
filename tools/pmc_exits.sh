#!/bin/bash
# Tuning tool: duration + instruction counters of the two headline kernels for the builds of tools/build_exit_variants.sh
# (kernels that leave at mark n: the difference between two marks is what the phase between them issues).
#   gpurun --timeout 1500 -- 'bash tools/pmc_exits.sh'        -> gpurun_out/pmc_exits.txt
export TMPDIR=/tmp
ROOT=$PWD
OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
RES=$OUT/pmc_exits.txt; : > "$RES"
SETA="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
SETB="SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_ANY"
one() {  # label lib pattern counters...
  local label=$1 lib=$2 pat=$3; shift 3
  rm -rf /tmp/pv && mkdir -p /tmp/pv && cd /tmp/pv
  local LIBENV=""; [ "$lib" != default ] && LIBENV="LERC_AMD_LIBRARY=$ROOT/lerc_amd/csrc/_var/$lib.so"
  if [ $# -gt 0 ]; then
    env $LIBENV timeout 300 rocprofv3 --kernel-trace --pmc "$@" -d /tmp/pv -o t -- python $ROOT/tools/roundtrip_loop.py 6 > /tmp/pv/log.txt 2>&1
  else
    env $LIBENV timeout 300 rocprofv3 --kernel-trace -d /tmp/pv -o t -- python $ROOT/tools/roundtrip_loop.py 12 > /tmp/pv/log.txt 2>&1
  fi
  local DB=$(find /tmp/pv -name '*.db' | head -1)
  echo "== $label $lib" | tee -a "$RES"
  if [ -z "$DB" ]; then tail -5 /tmp/pv/log.txt | tee -a "$RES"; else
    python $ROOT/tools/rocpd_summary.py "$DB" $pat 2>&1 | grep -E "$pat|SQ_" | sed -e 's/  */ /g' | cut -c1-150 | tee -a "$RES"; fi
  cd $ROOT
}
for L in default ${ENC_LIBS:-enc_exit1 enc_exit2 enc_exit3 enc_exit4 enc_exit5}; do
  one time $L k_fast_encode1
  one pmcA $L k_fast_encode1 $SETA
  one pmcB $L k_fast_encode1 $SETB
done
for L in default ${DEC_LIBS:-dec_exit0 dec_exit1 dec_exit2 dec_exit3 dec_exit4 dec_exit5}; do
  one time $L k_fast_decode_scan
  one pmcA $L k_fast_decode_scan $SETA
  one pmcB $L k_fast_decode_scan $SETB
done
