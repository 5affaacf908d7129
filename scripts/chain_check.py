#!/usr/bin/env python3
"""GPU box: the whole-stage / chain kernels (large launches) against the per-conv small-launch variants and the oracle:
the first CUs of a 301-CU batch must come out bit-identical when evaluated alone."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mltcnn_pkg
import oracle

pkg = mltcnn_pkg.load()
size, n = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 301
blob = pkg.weights.synthetic_blob(0, 10)
org, pred = pkg.synth.make_patches_bulk(size, n, 4242)
poc, qp = pkg.synth.make_scalars(n, 4242)
m = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob}, flags=pkg.capi.FLAG_NO_CALIBRATION)
s, l = m.predict_batch(org, pred, poc, qp)
s2, l2 = m.predict_batch(org, pred, poc, qp)
print("deterministic:", np.array_equal(l, l2))
k = 8
s1, l1 = m.predict_batch(org[:k], pred[:k], poc[:k], qp[:k])
print("large launch == small launch (first %d CUs):" % k, np.array_equal(l[:k], l1), " max diff %.3e" % np.abs(l[:k] - l1).max())
ref, rs = oracle.Oracle(blob).forward(org[:24], pred[:24], poc[:24], qp[:24], threads=8)
print("max|dlogit| vs oracle (24 CUs): large %.3e small %.3e" % (np.abs(l[:24] - ref).max(), np.abs(l1 - ref[:k]).max()))
print("finite:", np.isfinite(l).all())
