#!/usr/bin/env python3
"""Round 6, measurement only: what would the CUs the flat-content guard flags on NATURAL content do without it?  For a weight set: the shipped configuration
(flagged CUs carry the exact arithmetic's bits), the same tier without the flat guard (MLT_FLAG_NO_FLAT_GUARD), the exact arithmetic as the reference; the error of
the unguarded single pass on the flagged CUs against its error on the others, per content class.  usage: r06_flat_guard_need_probe.py [seeds] [n]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import mltcnn_pkg

pkg = mltcnn_pkg.load()
C = pkg.capi
seeds = [s for s in (sys.argv[1] if len(sys.argv) > 1 else "10,23,24").split(",") if s]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
size = 128
sets = {"natural": pkg.synth.natural_patches(size, n, 4242),
        "texture": pkg.synth.make_patches_bulk(size, n, 4243),
        "partial_flat": pkg.synth.make_patches(size, 512, 4244, pkg.synth.KIND_PARTIAL_FLAT),
        "flat": pkg.synth.make_patches(size, 256, 4245, pkg.synth.KIND_FLAT),
        "ramp": pkg.synth.make_patches(size, 256, 4246, pkg.synth.KIND_RAMP),
        "dither": pkg.synth.make_patches(size, 512, 4247, pkg.synth.KIND_DITHER),
        "low_contrast": pkg.synth.make_patches(size, 512, 4248, pkg.synth.KIND_LOW_CONTRAST),
        "flat_zero_resi": pkg.synth.make_patches(size, 256, 4249, pkg.synth.KIND_FLAT_ZERO_RESI),
        "partial_near_flat": pkg.synth.make_patches(size, 512, 4250, pkg.synth.KIND_PARTIAL_NEAR_FLAT)}
for sd in seeds:
    blob = open(sd, "rb").read() if os.path.exists(sd) else pkg.weights.synthetic_blob(0, int(sd))
    ship = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob})
    a = ship.arithmetic(size)
    bare = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob}, flags=C.FLAG_NO_FLAT_GUARD | C.FLAG_NO_DECISION_GUARD | C.FLAG_NO_MAGNITUDE_GUARD)
    ab = bare.arithmetic(size)
    ex = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob}, flags=C.FLAG_EXACT_128)
    nodec = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob}, flags=C.FLAG_NO_DECISION_GUARD)
    print(f"weight set {sd}: shipped tier {a['exact']} (w2 units 0x{a['w2_units']:x}, x units 0x{a['x_units']:x}, flat guard {a['flat_guard']}), unguarded tier {ab['exact']} (w2 0x{ab['w2_units']:x}, x 0x{ab['x_units']:x})")
    for name, (org, pred) in sets.items():
        k = len(org)
        poc, qp = pkg.synth.make_scalars(k, 99)
        _, le = ex.predict_batch(org, pred, poc, qp)
        _, lb = bare.predict_batch(org, pred, poc, qp)
        r0 = nodec.arithmetic(size)["guard_reruns"]
        _, ls = nodec.predict_batch(org, pred, poc, qp)
        reruns = nodec.arithmetic(size)["guard_reruns"] - r0
        flagged = (ls == le).all(axis=1)            # (flat- or magnitude-flagged CUs carry the exact bits)
        eb = np.abs(lb - le).max(axis=1)
        es = np.abs(ls - le).max(axis=1)            # the shipped configuration (without the decision guard) itself
        f, u = eb[flagged], eb[~flagged]
        def st(x):
            return f"n {x.size:5d} max {x.max():.2e} p99 {np.percentile(x, 99):.2e} rms {np.sqrt((x ** 2).mean()):.2e}" if x.size else "n     0"
        print(f"  {name:13s} re-runs {reruns:5d} | unguarded single pass on the FLAGGED CUs: {st(f)} | on the others: {st(u)} | flagged CUs beyond 1e-3: {(f > 1e-3).sum()}, beyond 0.65e-3: {(f > 0.65e-3).sum()} | SHIPPED: max {es.max():.2e}, beyond 1e-3: {(es > 1e-3).sum()}")
    for m in (ship, bare, ex, nodec):
        m.close()
