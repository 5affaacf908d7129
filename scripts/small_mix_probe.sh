# one gpurun call: the small models with the calibrated single-pass prefix against the exact arithmetic (MLT_NO_SMALL_MIX)
for size in 64 32 16; do
  for mode in mix exact; do
    if [ $mode = exact ]; then export MLT_TUNING=1 MLT_NO_SMALL_MIX=1; else unset MLT_TUNING MLT_NO_SMALL_MIX; fi
    MLT_CALIB_VERBOSE=1 python bench.py --size $size --no-cpu-baseline --cpu-sample 256 --steps 20 --warmup 10 2>gpurun_out/small_mix_$size_$mode.err | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); a=d['config']['arithmetic']; print('size $size $mode', round(d['value']), d['ms_per_step'], d['dtype'], 'x_stages', a.get('x_stages'), 'x_units', a.get('x_units'), 'calib', a['calib_rms_dlogit'], a['calib_max_dlogit'], 'err %.1e'%d['parity']['max_abs_dlogit'], 'mism', d['parity'].get('split_mismatch_decisive'), 'reruns', a['guard_reruns_per_step'], ' '.join('%.3f'%k['avg_ms'] for k in d['derived']['kernels']))
"
    grep "calibration" gpurun_out/small_mix_$size_$mode.err | cut -c1-260
  done
done
