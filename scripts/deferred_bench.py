#!/usr/bin/env python3
"""Per-CU cost of the deferred single-CU API (mlt_submit ... mlt_wait) against synchronous mlt_predict calls, for k CUs
whose mode decision the encoder could postpone together.  GPU box only."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mltcnn_pkg
pkg = mltcnn_pkg.load()
size = 128
blob = pkg.weights.synthetic_blob(0, 10)
m = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob})
org, pred = pkg.synth.make_patches_bulk(size, 64, 3)
poc, qp = pkg.synth.make_scalars(64, 3)
for i in range(4):
    m.predict(org[i], pred[i], int(poc[i]), int(qp[i]))
res = {}
for k in (1, 4, 16, 64):
    ts = []
    for rep in range(12):
        t0 = time.perf_counter()
        tk = [m.submit(org[i], pred[i], int(poc[i]), int(qp[i])) for i in range(k)]
        out = [m.wait(size, t) for t in tk]
        ts.append(time.perf_counter() - t0)
    dfr = float(np.median(ts[2:])) / k * 1e6
    ts = []
    for rep in range(6):
        t0 = time.perf_counter()
        for i in range(k):
            m.predict(org[i], pred[i], int(poc[i]), int(qp[i]))
        ts.append(time.perf_counter() - t0)
    syn = float(np.median(ts[1:])) / k * 1e6
    res[k] = (round(dfr, 1), round(syn, 1))
    print(f"k={k:3d}: deferred {dfr:7.1f} us/CU   synchronous {syn:7.1f} us/CU")
