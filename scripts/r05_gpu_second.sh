#!/bin/bash
# round 5, second GPU call: RA-toolset evaluation again (N3 fix), the trained (non-i.i.d.) weight family through calibration + tail probe,
# guard cost A/B of the bench line
out=gpurun_out/r05c
mkdir -p $out
timeout 3000 python tools/run_ra_eval.py --mode gpu --out $out/ra_eval > $out/ra_eval.log 2>&1; echo "ra_eval rc $?"
tail -12 $out/ra_eval.log | cut -c1-600
rm -rf $out/ra_eval/*/*.bin $out/ra_eval/torch_model
for s in 1 2; do timeout 900 python tools/train_synth_weights.py $out/trained$s.mltw --steps 300 --batch 16 --seed $s --threads 64 > $out/train$s.log 2>&1; tail -2 $out/train$s.log; done
timeout 1500 python scripts/tail_probe.py --seeds "" --blobs $out/trained1.mltw,$out/trained2.mltw --n 2048 --n-texture 8192 --natural 4096 > $out/tail_probe_trained.txt 2>&1; echo "tail probe rc $?"; grep -v "^generated" $out/tail_probe_trained.txt | cut -c1-400
rm -f $out/trained*.mltw
for f in 0 32; do timeout 600 python bench.py --steps 50 --warmup 20 --no-cpu-baseline --sustain-s 0 --cpu-sample 64 --flags $f > $out/bench_flags$f.json 2>/dev/null; python - $out/bench_flags$f.json $f <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('flags',sys.argv[2], d['value'], d['ms_per_step'], sum(k['avg_ms'] for k in d['derived']['kernels']))
PY
done
du -sh $out
