# one gpurun call: exact arithmetic at launch-unit granularity (tier below exact of the 128 model, half of layer0 for the small models)
MLT_CALIB_VERBOSE=1 python scripts/tier_probe.py 12 21 22 25 2>&1 | grep -E "^seed"
python scripts/w2_check.py 12 21 22 25 2>&1 | grep "^seed"
bash scripts/small_mix_probe.sh 2>&1 | grep -E "^size"
for s in 21 12 22 25; do
  python bench.py --no-cpu-baseline --cpu-sample 64 --weight-seed $s --steps 20 --warmup 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); a=d['config']['arithmetic']; print('seed $s', round(d['value']), d['ms_per_step'], d['dtype'], '%.1e'%d['parity']['max_abs_dlogit'], d['parity'].get('split_mismatch_decisive'))
"
done
