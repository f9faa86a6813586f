#!/bin/bash
# Tuning tool: kernel times (rocprofv3) of tools/roundtrip_loop.py for several library builds.
#   gpurun -- 'bash tools/trace_variants.sh lerc_amd/csrc/_var/a.so ...'
export TMPDIR=/tmp
ROOT=$PWD
for L in "$@"; do
  echo "== $L"
  rm -rf /tmp/tv && mkdir -p /tmp/tv && cd /tmp/tv
  LERC_AMD_LIBRARY=$ROOT/$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tv -o t -- python $ROOT/tools/roundtrip_loop.py 8 > /tmp/tv/log.txt 2>&1
  DB=$(find /tmp/tv -name '*.db' | head -1)
  python $ROOT/tools/rocpd_summary.py "$DB" 2>/dev/null | grep "lerc::" | cut -c1-60,90-130
  cd $ROOT
done
