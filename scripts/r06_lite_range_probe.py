#!/usr/bin/env python3
"""GPU box: what does the exact-lite arithmetic do with LARGE activations?  Its e4m3 copy of Xh saturates at 448 and the lo part is clamped at
448 / 4096.  Weight sets with the stem's residual filters scaled by g (weights.amplifying_blob) on uniform-noise content (residuals of hundreds of steps): the
forced exact-lite tier (MLT_FLAG_EXACT_128 | MLT_FLAG_EXACT_LITE) and the exact tier against the C oracle, and what the load-time calibration chooses by itself."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mltcnn_pkg  # noqa: E402
import oracle  # noqa: E402

pkg = mltcnn_pkg.load()
S, size, n = pkg.synth, 128, 48
org, pred = S.make_patches(size, n, 99, S.KIND_UNIFORM)
ot, pt = S.make_patches_bulk(size, n, 98)
org = np.concatenate([org, ot]); pred = np.concatenate([pred, pt])
poc, qp = S.make_scalars(2 * n, 99)
for g in (16.0, 64.0, 256.0, 1024.0):
    blob = pkg.weights.amplifying_blob(0, 10, g, 1.0)
    ref, _ = oracle.Oracle(blob).forward(org, pred, poc, qp, threads=min(os.cpu_count() or 8, 64))
    row = [f"gain {g:6.0f}: |logit| max {np.abs(ref).max():9.1f}"]
    for name, fl in (("exact-lite forced", pkg.capi.FLAG_EXACT_128 | pkg.capi.FLAG_EXACT_LITE), ("exact", pkg.capi.FLAG_EXACT_128), ("shipped", 0)):
        m = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob}, flags=fl)
        a = m.arithmetic(size)
        s, l = m.predict_batch(org, pred, poc, qp)
        e = np.abs(l - ref)
        row.append(f"{name}: uniform {e[:n].max():.2e} texture {e[n:].max():.2e} (relative {(e / np.maximum(np.abs(ref), 1.0)).max():.1e})" + (f" [tier {a['exact']}, magnitude guard {a['mag_guard_thr']:.3g}, reruns {a['guard_reruns']}]" if fl == 0 else ""))
        m.close()
    print(" | ".join(row), flush=True)
