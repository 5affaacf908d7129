import os, sys
import numpy as np
sys.path.insert(0, os.getcwd())
import mltcnn_pkg
pkg = mltcnn_pkg.load()
size = 128
blob = pkg.weights.synthetic_blob(0, 13)
m = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob}, max_batch=4096)
print("tier", m.arithmetic(size)["exact"])
for n in (4096, 1001, 257, 37, 1):
    org, pred = pkg.synth.make_patches_bulk(size, n, 99 + n)
    poc, qp = pkg.synth.make_scalars(n, 99 + n)
    org[n // 2] = 300; pred[n // 2] = 300
    s0, l0 = m.predict_batch(org, pred, poc, qp)
    ok = True
    for r in range(12):
        s, l = m.predict_batch(org, pred, poc, qp)
        ok &= np.array_equal(l, l0) and np.array_equal(s, s0)
    print(f"batch {n}: 12 runs identical: {ok}; finite {np.isfinite(l0).all()}")
print("reruns", m.arithmetic(size)["guard_reruns"])
