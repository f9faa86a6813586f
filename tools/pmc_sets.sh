#!/bin/bash
# Tuning tool: arbitrary PMC passes over the default bench loop.  PAT=<kernel name pattern> bash tools/pmc_sets.sh "SET1 counters" "SET2 counters" ...
export TMPDIR=/tmp
ROOT=$PWD
PAT=${PAT:-fast}
i=0
for SET in "$@"; do
  i=$((i+1))
  rm -rf /tmp/prof_pmc$i
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --pmc $SET -d /tmp/prof_pmc$i -o pmc -- python $ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-c5-anchor --rotate 0 > /dev/null 2> /tmp/pmc$i.err
  DB=$(find /tmp/prof_pmc$i -name '*.db' | head -1)
  { echo "# counters: $SET"; python "$ROOT/tools/rocpd_summary.py" "$DB" $PAT; } 2>&1 | cut -c1-160
  cd $ROOT
done
