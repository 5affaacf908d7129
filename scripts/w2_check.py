#!/usr/bin/env python3
"""GPU box: the hi+lo-weights tiers (fused W2 kernels, round 4) on weight sets that land there: large launches (chain_kernel<..., W2>,
stem_block / block32 W2 forms) against small launches (per-conv nsplit-4 forms) bit for bit, both against the oracle, determinism.
usage: w2_check.py [seed ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mltcnn_pkg
import oracle

pkg = mltcnn_pkg.load()
size, n = 128, 301
seeds = [int(a) for a in sys.argv[1:]] or [13, 23]
org, pred = pkg.synth.make_patches_bulk(size, n, 4242)
poc, qp = pkg.synth.make_scalars(n, 4242)
for seed in seeds:
    blob = pkg.weights.synthetic_blob(0, seed)
    m = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob})
    a = m.arithmetic(size)
    s, l = m.predict_batch(org, pred, poc, qp)
    s2, l2 = m.predict_batch(org, pred, poc, qp)
    k = 8
    s1, l1 = m.predict_batch(org[:k], pred[:k], poc[:k], qp[:k])
    sp, lp = m.predict(org[2], pred[2], int(poc[2]), int(qp[2]))
    ref, rs = oracle.Oracle(blob).forward(org[:24], pred[:24], poc[:24], qp[:24], threads=8)
    print(f"seed {seed}: tier {a['exact']} w2 0x{a['w2_stages']:x} (units 0x{a['w2_units']:x}) x 0x{a['x_stages']:x} (units 0x{a['x_units']:x}) calib rms {a['calib_rms']:.2e} max {a['calib_max']:.2e} | deterministic {np.array_equal(l, l2)} | "
          f"large == small {np.array_equal(l[:k], l1)} (max diff {np.abs(l[:k] - l1).max():.2e}) | predict == batch {np.array_equal(lp, l[2])} | "
          f"vs oracle: large {np.abs(l[:24] - ref).max():.2e} small {np.abs(l1 - ref[:k]).max():.2e} | finite {np.isfinite(l).all()}")
    m.close()
