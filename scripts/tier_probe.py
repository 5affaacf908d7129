#!/usr/bin/env python3
"""GPU box: which arithmetic tier the load-time calibration picks for a list of seeded weight sets (MLT_CALIB_VERBOSE=1 prints every
candidate it priced on stderr), how long the load took, and a quick parity check of the chosen tier.  usage: tier_probe.py [seed ...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mltcnn_pkg
import oracle

pkg = mltcnn_pkg.load()
size, n = 128, 24
seeds = [int(a) for a in sys.argv[1:]] or [10, 11, 12, 13, 21, 22, 23, 24, 25]
org, pred = pkg.synth.make_patches_bulk(size, n, 4242)
poc, qp = pkg.synth.make_scalars(n, 4242)
for seed in seeds:
    blob = pkg.weights.synthetic_blob(0, seed)
    t0 = time.perf_counter()
    m = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob})
    dt = time.perf_counter() - t0
    a = m.arithmetic(size)
    s, l = m.predict_batch(org, pred, poc, qp)
    ref, rs = oracle.Oracle(blob).forward(org, pred, poc, qp, threads=8)
    print(f"seed {seed}: tier {a['exact']} rounding {a['rounding']} stages 0x{a['w2_stages']:x} (units 0x{a['w2_units']:x}) exact-stages 0x{a['x_stages']:x} (units 0x{a['x_units']:x}) calib rms {a['calib_rms']:.2e} max {a['calib_max']:.2e} load {dt:.2f} s "
          f"| vs oracle {np.abs(l - ref).max():.2e}", flush=True)
    m.close()
