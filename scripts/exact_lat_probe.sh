# one gpurun call: the exact arithmetic's latency variants (round 4) on / off: one-CU calls of the exact and exact-stage tiers, the guard's
# re-run of a few flagged CUs (natural content), small batches of the 64 x 64 model
for mode in on off; do
  if [ $mode = off ]; then export MLT_TUNING=1 MLT_NO_EXACT_LAT=1; else unset MLT_TUNING MLT_NO_EXACT_LAT; fi
  echo "== exact latency variants $mode"
  python scripts/latency_run.py 80 10 1
  python scripts/latency_run.py 80 22 0
  python scripts/latency_run.py 80 12 0
  python bench.py --no-cpu-baseline --cpu-sample 64 --content natural --steps 30 --warmup 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('natural', round(d['value']), d['ms_per_step'], 'reruns', d['config']['arithmetic']['guard_reruns_per_step'], '%.1e'%d['parity']['max_abs_dlogit'])
"
  python bench.py --no-cpu-baseline --cpu-sample 64 --size 64 --batch 64 --steps 200 --warmup 20 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('64x64 batch 64', round(d['value']), d['ms_per_step'], '%.1e'%d['parity']['max_abs_dlogit'])
"
done
