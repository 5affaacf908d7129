# one gpurun call: the fast 32->64 stride-2 kernel on smaller tiles with more than one persistent workgroup per CU
for v in base s2_wp2_un6 s2_wp2_un3 base s2_wp2_un6; do
  for cap in 256 512 768; do
    MLT_TUNING=1 MLT_WG_CAP=$cap MLT_LIB_PATH=$PWD/fastintercu-vvc_amd/_variants/lib_$v.so python bench.py --no-cpu-baseline --cpu-sample 32 --steps 20 --warmup 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v cap $cap', round(d['value']), '%.1e'%d['parity']['max_abs_dlogit'], ' '.join('%.3f'%k['avg_ms'] for k in d['derived']['kernels']))
"
  done
done
