#!/bin/bash
# round 6: the trained weight families (tools/train_synth_weights.py; blobs staged under tests/data/_blobs/, git-ignored) and the amplifying stand-in through
# the load-time calibration with the magnitude guard (MLT_CALIB_VERBOSE: one line per candidate), the plain rule beside it, bench legs with ALL 4096 CUs
# against the oracle, and the tail probe (>= 300 k logits of six content classes) of the shipped tiers
out=gpurun_out/${1:-r06b}
mkdir -p $out
for f in tests/data/_blobs/*.mltw; do
  b=$(basename $f .mltw)
  MLT_CALIB_VERBOSE=1 timeout 600 python - $f > $out/calib_$b.txt 2>&1 <<'PY'
import sys, time, numpy as np
sys.path.insert(0, '.')
import mltcnn_pkg
pkg = mltcnn_pkg.load()
blob = open(sys.argv[1], 'rb').read()
t0 = time.time()
m = pkg.MltCnn(device=0, sizes=(128,), blobs={128: blob})
print("ARITH (%.2f s to load + calibrate)" % (time.time() - t0), m.arithmetic(128))
m.close()
t0 = time.time()
p = pkg.MltCnn(device=0, sizes=(128,), blobs={128: blob}, flags=pkg.capi.FLAG_NO_MAGNITUDE_GUARD)
print("ARITH, plain rule only (%.2f s)" % (time.time() - t0), p.arithmetic(128))
p.close()
PY
  grep -E "ARITH|behind the magnitude" $out/calib_$b.txt | cut -c1-420 | tail -12
  timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --cpu-sample 4096 --weights-blob $f > $out/bench_$b.json 2> $out/bench_$b.err; echo "bench $b rc $?"
  timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --cpu-sample 4096 --weights-blob $f --content natural > $out/bench_${b}_natural.json 2>> $out/bench_$b.err; echo "bench $b natural rc $?"
  python - $out/bench_$b.json $out/bench_${b}_natural.json <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        a = d['config']['arithmetic']
        print(f, round(d['value']), 'CU/s', d['ms_per_step'], 'ms |', a['mode'], '| reruns/step', a['guard_reruns_per_step'], '| parity', d['parity']['checked_cus'], 'CUs max', d['parity']['max_abs_dlogit'], 'decisive mismatches', d['parity']['split_mismatch_decisive'])
    except Exception as e:
        print(f, 'FAILED', e)
PY
done
blobs=$(ls tests/data/_blobs/*.mltw | tr '\n' ',' | sed 's/,$//')
timeout 1500 python scripts/tail_probe.py --seeds "" --blobs $blobs --natural 4096 > $out/tail_probe_trained.txt 2>&1; echo "tail probe rc $?"
grep -E "^seed|=>|natural|texture" $out/tail_probe_trained.txt | cut -c1-330
