# tuning: the one-launch decoder against the two-launch form, and its lead (groups of discovery workgroups dispatched first)
cd /root/repo
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "streaming or ragged or give_up or later_bands or elevation or roundtrip or fast" 2>&1 | tail -2
run() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-c5-anchor 2>&1 | tail -1 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print('$1', j['value'], j['ms_per_step'], {k: v['avg_ms'] for k, v in j['sync_per_call']['kernels'].items()} if 'sync_per_call' in j and j['sync_per_call'] and 'kernels' in j['sync_per_call'] else '')"; }
LERC_AMD_DECODE_LAUNCHES=2 run launches=2
for lead in ${LEADS:-24 48 64 96 128 256}; do LERC_AMD_DECODE_LEAD=$lead run lead=$lead; done
