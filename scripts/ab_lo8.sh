# A/B inside one gpurun call: the hi+lo-weights tiers with the fp16 lo product against the FP8 lo product (profiles/r04m_lo8_ab.txt).
# lib_base.so = build.build_lib(force=True, out='fastintercu-vvc_amd/_variants/lib_base.so') on a checkout of the commit before the FP8 lo
# product (797e4c3); the variants directory is not tracked.
for v in base lo8 base lo8; do
  if [ $v = base ]; then export MLT_LIB_PATH=$PWD/fastintercu-vvc_amd/_variants/lib_base.so; else unset MLT_LIB_PATH; fi
  for s in 13 11; do
  python bench.py --no-cpu-baseline --cpu-sample 64 --weight-seed $s --steps 20 --warmup 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); a=d['config']['arithmetic']; print('$v seed $s', round(d['value']), d['ms_per_step'], a.get('w2_stages'), '%.1e'%d['parity']['max_abs_dlogit'], ' | '.join(k['name'][:12]+' %.3f'%k['avg_ms'] for k in d['derived']['kernels']))
"
  done
done
