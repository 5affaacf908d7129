#!/bin/bash
# round 6, sixth GPU call: the RANGE guard (plain-admitted tiers: CUs beyond 1.5 x the largest calibrated magnitude are re-run) -- whole parity suite, re-runs per step of the
# bench batch and of natural content for the seeded sets, the trained families
out=gpurun_out/r06g
mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log; tail -4 $out/pytest_gpu.log
S="--no-cpu-baseline --cpu-sample 4096"
for s in 10 23 24 13 11 21 25 12 22; do python bench.py $S --weight-seed $s --steps 30 --warmup 10 > $out/bench_seed$s.json 2>> $out/bench.err; python bench.py $S --weight-seed $s --steps 30 --warmup 10 --content natural > $out/bench_seed${s}_natural.json 2>> $out/bench.err; done
python bench.py $S --flags 0x28 > $out/bench_no_guards.json 2>> $out/bench.err
for f in tests/data/_blobs/*.mltw; do b=$(basename $f .mltw); python bench.py $S --weights-blob $f --steps 20 --warmup 5 > $out/bench_$b.json 2>> $out/bench.err; python bench.py $S --weights-blob $f --steps 20 --warmup 5 --content natural > $out/bench_${b}_natural.json 2>> $out/bench.err; done
python - $out <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + '/bench_*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1]); a = d['config']['arithmetic']
        print(f.split('/')[-1][6:-5].ljust(22), round(d['value']), 'CU/s | tier', a['mode'][:60], '| magnitude guard', a['magnitude_guard_kind'][:9], '%.3g' % a['magnitude_guard_thr'], '| reruns/step', a['guard_reruns_per_step'], '| max', '%.2e' % d['parity']['max_abs_dlogit'], 'mism', d['parity']['split_mismatch_decisive'])
    except Exception as e:
        print(f, 'FAILED', e)
PY
