#!/bin/bash
# Tuning tool: instruction counts (one PMC pass) of tools/roundtrip_loop.py for several library builds whose results need not be valid.
#   gpurun -- 'bash tools/pmc_variants.sh pattern lib1.so lib2.so ...'
export TMPDIR=/tmp
ROOT=$PWD
PAT=$1; shift
for L in "$@"; do
  echo "== $L"
  rm -rf /tmp/pv && mkdir -p /tmp/pv && cd /tmp/pv
  LERC_AMD_ENCODE_FORM=${FORM:-1} LERC_AMD_LIBRARY=$ROOT/$L timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES -d /tmp/pv -o t -- python $ROOT/tools/roundtrip_loop.py 6 > /tmp/pv/log.txt 2>&1
  DB=$(find /tmp/pv -name '*.db' | head -1)
  python $ROOT/tools/rocpd_summary.py "$DB" $PAT 2>/dev/null | grep -E "avg_us|$PAT|SQ_" | cut -c1-60,90-140
  cd $ROOT
done
