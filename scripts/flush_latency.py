#!/usr/bin/env python3
"""GPU box: latency of ONE deferred batch -- mlt_flush to the first mlt_wait returning -- at n = 1, 2, 4, 8, 16, 32 CUs of 128 x 128 (the N3 regime: a WPP
anti-diagonal of an 832 x 480 picture carries 1-3 CUs, of a 1080p picture up to 8, of a 4K picture up to 15), next to the synchronous mlt_predict call.
The staging of the CUs (mlt_submit: a strided gather into pinned memory) is outside the timed region; the H2D copy of the batch is inside.
usage: python scripts/flush_latency.py [weight seed = 10] [reps = 40]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mltcnn_pkg  # noqa: E402

pkg = mltcnn_pkg.load()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 10
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
size = 128
m = pkg.MltCnn(device=0, sizes=(size,), blobs={size: pkg.weights.synthetic_blob(0, seed)})
a = m.arithmetic(size)
org, pred = pkg.synth.make_patches_bulk(size, 64, 3)
poc, qp = pkg.synth.make_scalars(64, 3)
ref = [m.predict(org[i], pred[i], int(poc[i]), int(qp[i])) for i in range(32)]
ts = []
for r in range(reps):
    t0 = time.perf_counter()
    m.predict(org[r % 32], pred[r % 32], int(poc[r % 32]), int(qp[r % 32]))
    ts.append(time.perf_counter() - t0)
print(f"weight seed {seed}: tier {a['exact']}, hi+lo units 0x{a['w2_units']:x}, exact units 0x{a['x_units']:x}")
print(f"synchronous mlt_predict (gather + H2D + 16 kernels from a hipGraph + D2H): {np.median(ts[5:]) * 1e6:7.1f} us per call")
for n in (1, 2, 4, 8, 16, 32):
    ts = []
    for r in range(reps):
        tk = [m.submit(org[i], pred[i], int(poc[i]), int(qp[i])) for i in range(n)]
        t0 = time.perf_counter()
        m.flush(size)
        out = m.wait(size, tk[-1])
        ts.append(time.perf_counter() - t0)
        rest = [m.wait(size, t) for t in tk[:-1]] + [out]
        for i in range(n):
            assert rest[i][0] == ref[i][0] and np.array_equal(rest[i][1], ref[i][1]), (n, i)   # the batch's bits are the single call's
    med = float(np.median(ts[5:])) * 1e6
    print(f"flush of {n:2d} CUs -> first wait: {med:7.1f} us  ({med / n:6.1f} us per CU), bit-identical to the one-CU calls")
m.close()
