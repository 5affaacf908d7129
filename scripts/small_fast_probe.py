#!/usr/bin/env python3
"""GPU box: what the 64 / 32 / 16 models would do in the fast arithmetic (MLT_FLAG_FAST_SMALL): error against the oracle over
several weight seeds, and throughput next to the exact default.  usage: small_fast_probe.py [n]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mltcnn_pkg
import oracle
import torch

pkg = mltcnn_pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = torch.device("cuda:0")
for size in (64, 32, 16):
    arch = pkg.synth.arch_for_size(size)
    for seed in (10, 11, 12, 13, 22):
        blob = pkg.weights.synthetic_blob(arch, seed)
        org, pred = pkg.synth.make_patches_bulk(size, n, 777)
        poc, qp = pkg.synth.make_scalars(n, 777)
        ref, rs = oracle.Oracle(blob).forward(org, pred, poc, qp, threads=8)
        row = []
        for name, flags in (("exact", 0), ("fast", pkg.capi.FLAG_FAST_SMALL)):
            m = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob}, flags=flags, max_batch=4096)
            s, l = m.predict_batch(org, pred, poc, qp)
            err = np.abs(l - ref)
            nb = 4096
            o2, p2 = pkg.synth.make_patches_bulk(size, nb, 5)
            c2, q2 = pkg.synth.make_scalars(nb, 5)
            t = [torch.from_numpy(x).to(dev) for x in (o2, p2, c2, q2)]
            sp = torch.empty(nb, dtype=torch.int32, device=dev)
            lg = torch.empty(nb, l.shape[1], dtype=torch.float32, device=dev)
            for _ in range(3):
                m.predict_batch_device(nb, size, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), sp.data_ptr(), lg.data_ptr())
            m.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                m.predict_batch_device(nb, size, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), sp.data_ptr(), lg.data_ptr())
            m.synchronize()
            dt = (time.perf_counter() - t0) / 10
            row.append(f"{name}: max {err.max():.2e} rms {np.sqrt((err ** 2).mean()):.2e} split mism {int((s != rs).sum())} {nb / dt / 1e6:.2f} M CU/s")
            m.close()
        print(f"size {size} seed {seed}: " + " | ".join(row))
