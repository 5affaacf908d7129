MLT_CALIB_VERBOSE=1 python scripts/tier_probe.py 12 21 22 2>&1 | grep -v amdgpu.ids | grep -E "exact in 0x[1-9a-f]|^seed"
python scripts/w2_check.py 12 21 22 2>&1 | grep "^seed"
for s in 12 21 22; do
  python bench.py --no-cpu-baseline --cpu-sample 64 --weight-seed $s --steps 10 --warmup 5 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); a=d['config']['arithmetic']; print('seed $s', round(d['value']), d['ms_per_step'], d['dtype'], '%.1e'%d['parity']['max_abs_dlogit'], d['parity'].get('split_mismatch_decisive'), ' | '.join(k['name'][:14]+' %.3f'%k['avg_ms'] for k in d['derived']['kernels']))
"
done
