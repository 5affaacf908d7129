#!/usr/bin/env python3
"""GPU box: A/B of an environment switch of the runtime (e.g. MLT_NO_CHAIN): the same seeded batch through two child
processes, logits compared bit for bit, plus bench.py timings of both.  usage: ab_env.py MLT_NO_CHAIN [n] [size]"""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 1 and sys.argv[1] == "--child":
    out, n, size = sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    import mltcnn_pkg
    pkg = mltcnn_pkg.load()
    arch = pkg.synth.arch_for_size(size)
    blob = pkg.weights.synthetic_blob(arch, 10)
    org, pred = pkg.synth.make_patches_bulk(size, n, 4242)
    poc, qp = pkg.synth.make_scalars(n, 4242)
    m = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob}, flags=pkg.capi.FLAG_NO_CALIBRATION)
    s, l = m.predict_batch(org, pred, poc, qp)
    s1, l1 = m.predict_batch(org[:3], pred[:3], poc[:3], qp[:3])
    np.save(out, np.concatenate([l.ravel(), l1.ravel(), s.astype(np.float32), s1.astype(np.float32)]))
    sys.exit(0)

var = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 301
size = int(sys.argv[3]) if len(sys.argv) > 3 else 128
res = {}
for val in (None, "1"):
    env = dict(os.environ, MLT_TUNING="1")
    if val:
        env[var] = val
    out = f"/tmp/ab_{var}_{val}.npy"
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", out, str(n), str(size)], env=env)
    res[val] = np.load(out)
    b = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "10", "--size", str(size)], env=env, capture_output=True, text=True).stdout
    line = [l for l in b.splitlines() if l.startswith("{")]
    if line:
        d = json.loads(line[-1])
        print(f"{var}={val}: {d['value']:.0f} CU/s  parity {d['parity']['max_abs_dlogit']:.2e}  " + " | ".join(f"{k['name'][:28]} {k['avg_ms']:.3f}" for k in d["derived"]["kernels"]))
print("bit-identical:", bool(np.array_equal(res[None], res["1"])), " max diff", float(np.abs(res[None] - res["1"]).max()))
