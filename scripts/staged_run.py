#!/usr/bin/env python3
"""mlt_predict_batch from pinned host memory, a few times (for rocprofv3 --kernel-trace --memory-copy-trace)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mltcnn_pkg
pkg = mltcnn_pkg.load()
B = 4096
blob = pkg.weights.synthetic_blob(0, 10)
m = pkg.MltCnn(device=0, sizes=(128,), blobs={128: blob}, max_batch=B)
org, pred = pkg.synth.make_patches_bulk(128, B, 1)
poc, qp = pkg.synth.make_scalars(B, 1)
ho = torch.from_numpy(org).pin_memory().numpy()
hp = torch.from_numpy(pred).pin_memory().numpy()
for i in range(4):
    t0 = time.perf_counter()
    m.predict_batch(ho, hp, poc, qp)
    print("call", i, round((time.perf_counter() - t0) * 1e3, 2), "ms")
