# GPU box: package power and clocks while the bench loop runs (the "power-limited chip" claim of DESIGN.md, measured): rocm-smi sampled every
# 0.5 s beside a long bench run, for the single-pass tier and for the hi+lo-weights tier of seed 13, plus the idle reading before / after.
out=gpurun_out/power_probe.txt
: > $out
echo "== idle" >> $out
rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)" >> $out
for seed in 10 13; do
  echo "== bench --weight-seed $seed (6000 steps)" >> $out
  python bench.py --no-cpu-baseline --cpu-sample 1 --weight-seed $seed --steps 6000 --warmup 10 > gpurun_out/power_bench_$seed.json 2>/dev/null &
  pid=$!
  sleep 40   # import + load + calibration + the parity sample
  for i in $(seq 1 30); do
    rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" | tr '\n' ' ' >> $out; echo >> $out
    sleep 0.5
  done
  wait $pid
  python -c "
import json
for l in open('gpurun_out/power_bench_$seed.json'):
    if l.startswith('{'):
        d=json.loads(l); print('bench', round(d['value']), 'CU/s', d['ms_per_step'], 'ms/step')
" >> $out
done
echo "== idle after" >> $out
rocm-smi --showpower --showclocks 2>&1 | grep -E "Power|sclk" >> $out
cat $out
