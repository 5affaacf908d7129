#!/usr/bin/env python3
"""GPU box, round 5: the exact-lite arithmetic (MLT_FLAG_EXACT_128 | MLT_FLAG_EXACT_LITE) against the C oracle and the exact arithmetic:
error on golden-style seeded inputs for several weight sets (+ a trained blob when given), bit-identity single-CU vs batch, rate at batch 4096."""
import sys, time
import numpy as np
sys.path.insert(0, '.')
import mltcnn_pkg, oracle
pkg = mltcnn_pkg.load()
F = pkg.capi
size = 128
blobs = [(f"seed {s}", pkg.weights.synthetic_blob(0, s)) for s in (10, 13, 21, 22)] + [(b, open(b, 'rb').read()) for b in sys.argv[1:]]
org, pred = pkg.synth.natural_patches(size, 192, 4242)
o2, p2 = pkg.synth.make_patches_bulk(size, 64, 99)
org, pred = np.concatenate([org, o2]), np.concatenate([pred, p2])
org[5] = 500; pred[5] = 500
poc, qp = pkg.synth.make_scalars(len(org), 4242)
for name, blob in blobs:
    ref, ref_split = oracle.Oracle(blob).forward(org, pred, poc, qp, threads=64)
    out = {}
    for tag, fl in (("exact", F.FLAG_EXACT_128), ("lite", F.FLAG_EXACT_128 | F.FLAG_EXACT_LITE)):
        m = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob}, flags=fl)
        s, l = m.predict_batch(org, pred, poc, qp)
        d = np.abs(l - ref)
        ok1 = all((lambda r: r[0] == s[i] and np.array_equal(r[1], l[i]))(m.predict(org[i], pred[i], int(poc[i]), int(qp[i]))) for i in (0, 5, 100, 200))
        s9, l9 = m.predict_batch(org[:9], pred[:9], poc[:9], qp[:9])
        out[tag] = (float(d.max()), float(np.sqrt((d ** 2).mean())), int((s != ref_split).sum()), ok1 and np.array_equal(l9, l[:9]))
        m.close()
    print(f"{name}: |logit| rms {np.sqrt((ref**2).mean()):.1f}  exact max {out['exact'][0]:.2e} rms {out['exact'][1]:.2e} mism {out['exact'][2]} bitid {out['exact'][3]} | lite max {out['lite'][0]:.2e} rms {out['lite'][1]:.2e} mism {out['lite'][2]} bitid {out['lite'][3]}", flush=True)
