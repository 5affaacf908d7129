#!/bin/bash
out=gpurun_out/${1:-r05e}
mkdir -p $out
timeout 900 python tools/train_synth_weights.py $out/trained1.mltw --steps 300 --batch 16 --seed 1 --threads 64 > $out/train1.log 2>&1; tail -1 $out/train1.log
timeout 900 python scripts/r05_xlite_probe.py $out/trained1.mltw 2>&1 | grep -v amdgpu.ids | tee $out/xlite_probe.txt
for f in 1 65; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain-s 0 --cpu-sample 256 --flags $f > $out/bench_flags$f.json 2>$out/bench_flags$f.err; python - $out/bench_flags$f.json $f <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print('flags',sys.argv[2], d['value'], d['ms_per_step'], d['parity']['max_abs_dlogit'], d['parity']['split_mismatch_decisive'], [(k['name'][9:30],k['avg_ms']) for k in d['derived']['kernels']])
except Exception as e: print('flags', sys.argv[2], 'FAILED', e); print(open(sys.argv[1][:-5]+'.err').read()[-1500:])
PY
done
rm -f $out/trained1.mltw
