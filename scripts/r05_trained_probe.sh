#!/bin/bash
# round 5: where does the TRAINED weight family land, tier by tier (MLT_CALIB_VERBOSE), what does it run at, and how far is its shipped tier from the C oracle
out=gpurun_out/${1:-r05d}
mkdir -p $out
timeout 900 python tools/train_synth_weights.py $out/trained1.mltw --steps 300 --batch 16 --seed 1 --threads 64 > $out/train1.log 2>&1; tail -1 $out/train1.log
MLT_CALIB_VERBOSE=1 timeout 600 python - $out/trained1.mltw > $out/calib_verbose.txt 2>&1 <<'PY'
import sys, numpy as np
sys.path.insert(0, '.')
import mltcnn_pkg, oracle
pkg = mltcnn_pkg.load()
blob = open(sys.argv[1], 'rb').read()
m = pkg.MltCnn(device=0, sizes=(128,), blobs={128: blob})
print("ARITH", m.arithmetic(128))
org, pred = pkg.synth.natural_patches(128, 256, 777)
poc, qp = pkg.synth.make_scalars(256, 777)
ref, ref_split = oracle.Oracle(blob).forward(org, pred, poc, qp, threads=64)
s, l = m.predict_batch(org, pred, poc, qp)
print("SHIPPED vs oracle: max|dlogit| %.3e, |logit| max %.1f, split mismatches %d" % (np.abs(l - ref).max(), np.abs(ref).max(), int((s != ref_split).sum())))
for name, fl in (("single pass forced", pkg.capi.FLAG_NO_CALIBRATION | pkg.capi.FLAG_NO_DECISION_GUARD),):
    f = pkg.MltCnn(device=0, sizes=(128,), blobs={128: blob}, flags=fl)
    s2, l2 = f.predict_batch(org, pred, poc, qp)
    d = np.abs(l2 - ref)
    print(name, "vs oracle: max %.3e rms %.3e per head rms" % (d.max(), np.sqrt((d**2).mean())), [float(np.sqrt((d[:, a:b]**2).mean())) for a, b in ((0, 2), (2, 5), (5, 9))], "relative to |logit| rms", float(np.sqrt((ref**2).mean())))
    f.close()
m.close()
PY
grep -v "^$" $out/calib_verbose.txt | cut -c1-330 | tail -60
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --sustain-s 0 --cpu-sample 256 --weights-blob $out/trained1.mltw > $out/bench_trained1.json 2>/dev/null
python - $out/bench_trained1.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('trained1 bench', d['value'], d['ms_per_step'], d['dtype'], d['parity']['max_abs_dlogit'], d['parity']['split_mismatch_decisive'])
PY
rm -f $out/trained1.mltw
