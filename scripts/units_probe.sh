# one gpurun call: the hi+lo-weights tiers at launch-unit granularity against the stage granularity (MLT_NO_W2_UNITS), same box
MLT_CALIB_VERBOSE=1 python scripts/tier_probe.py 11 13 23 24 25 12 21 22 2>&1 | grep -E "^seed"
python scripts/w2_check.py 11 13 23 24 12 2>&1 | grep "^seed"
for s in 13 11 24 23; do
 for mode in units stages; do
  if [ $mode = stages ]; then export MLT_TUNING=1 MLT_NO_W2_UNITS=1; else unset MLT_TUNING MLT_NO_W2_UNITS; fi
  python bench.py --no-cpu-baseline --cpu-sample 64 --weight-seed $s --steps 20 --warmup 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); a=d['config']['arithmetic']; print('seed $s $mode', round(d['value']), d['ms_per_step'], d['dtype'], 'units 0x%x'%a['w2_units'], '%.1e'%d['parity']['max_abs_dlogit'], d['parity'].get('split_mismatch_decisive'), ' | '.join(k['name'][:12]+' %.3f'%k['avg_ms'] for k in d['derived']['kernels']))
"
 done
done
