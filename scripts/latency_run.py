#!/usr/bin/env python3
"""GPU box: N synchronous mlt_predict calls (the encoder's real mode: one CU per call, graph replay) for a rocprofv3 --kernel-trace
timeline, plus host-side timing of the call's phases.  usage: latency_run.py [n_calls] [weight_seed] [flags]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mltcnn_pkg

pkg = mltcnn_pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 10
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 0
size = 128
blob = pkg.weights.synthetic_blob(0, seed)
org, pred = pkg.synth.make_patches_bulk(size, 8, 99)
poc, qp = pkg.synth.make_scalars(8, 99)
m = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob}, max_batch=1, flags=flags)
lat = []
for i in range(n):
    t0 = time.perf_counter()
    m.predict(org[i % 8], pred[i % 8], int(poc[i % 8]), int(qp[i % 8]))
    lat.append(time.perf_counter() - t0)
print("tier", m.arithmetic(size)["exact"], "median us", float(np.median(lat[10:]) * 1e6), "min us", float(np.min(lat[10:]) * 1e6))
