#!/usr/bin/env python3
"""Race screen of the non-default tiers (round 4: hi+lo-weights kernels with the FP8 lo product and its e4m3 LDS image, exact stages inside a
single-plane network, the exact arithmetic's small-launch variants): the same batch through the shipped tier of a weight set many times,
every run bit-identical to the first, for batch sizes with different tile counts / tails.  GPU box only.
usage: w2_race.py [repeats] [seed ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mltcnn_pkg

pkg = mltcnn_pkg.load()
size = 128
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
seeds = [int(a) for a in sys.argv[2:]] or [13]
bad = 0
for seed in seeds:
    blob = pkg.weights.synthetic_blob(0, seed)
    m = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob}, max_batch=4096)
    a = m.arithmetic(size)
    print(f"seed {seed}: tier {a['exact']} units 0x{a['w2_units']:x} exact units 0x{a['x_units']:x}", flush=True)
    for n in (4096, 1001, 257, 37, 1):
        org, pred = pkg.synth.make_patches_bulk(size, n, 99 + n)
        poc, qp = pkg.synth.make_scalars(n, 99 + n)
        org[n // 2] = 300; pred[n // 2] = 300   # one flagged CU: the guard's exact re-run rides along
        s0, l0 = m.predict_batch(org, pred, poc, qp)
        ok = True
        for r in range(reps):
            s, l = m.predict_batch(org, pred, poc, qp)
            ok &= np.array_equal(l, l0) and np.array_equal(s, s0)
        bad += not ok
        print(f"  batch {n}: {reps} runs identical: {ok}; finite {np.isfinite(l0).all()}", flush=True)
    m.close()
print("RACE SCREEN", "FAILED" if bad else "clean")
