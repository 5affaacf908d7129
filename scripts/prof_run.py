#!/usr/bin/env python3
"""Small driver for rocprofv3: runs the 128x128 hot path a few times on a fixed batch (no torch.distributed,
no oracle) so per-kernel counters are easy to read.  usage: prof_run.py [batch] [steps] [flags, e.g. 1 = the exact arithmetic]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import mltcnn_pkg

pkg = mltcnn_pkg.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
flags = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
size = 128
dev = torch.device("cuda", 0)
blob = pkg.weights.synthetic_blob(0, 10)
m = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob}, max_batch=B, flags=flags)
org, pred = pkg.synth.make_patches_bulk(size, B, 0xC0FFEE)
poc, qp = pkg.synth.make_scalars(B, 0xC0FFEE)
d = [torch.from_numpy(x).to(dev) for x in (org, pred, poc, qp)]
d_split = torch.zeros(B, dtype=torch.int32, device=dev)
d_logits = torch.zeros((B, 9), dtype=torch.float32, device=dev)
for _ in range(steps):
    m.predict_batch_device(B, size, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d_split.data_ptr(), d_logits.data_ptr())
m.synchronize()
print("done", float(d_logits.abs().sum()))
