#!/usr/bin/env python3
"""Measured machine peaks beside the vendor figures bench.py prices against ("measure both on the box"):
hipBLASLt/rocBLAS fp16 GEMM rate (what a tuned dense kernel sustains on this chip under its loaded clock) and a streaming
copy (HBM read + write).  Uses torch only as a launcher for the vendor kernels; nothing here is on the product path.
GPU box only."""
import json
import sys

import torch

dev = torch.device("cuda", 0)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


out = {}
for n in (4096, 8192):
    a = torch.randn(n, n, device=dev, dtype=torch.float16)
    b = torch.randn(n, n, device=dev, dtype=torch.float16)
    t = timed(lambda: torch.matmul(a, b), 20)
    out[f"gemm_f16_{n}_tflops"] = round(2.0 * n ** 3 / t / 1e12, 1)
for gb in (1, 4):
    x = torch.empty(gb * (1 << 30), device=dev, dtype=torch.uint8)
    y = torch.empty_like(x)
    t = timed(lambda: y.copy_(x), 10)
    out[f"copy_{gb}GiB_read_plus_write_TBps"] = round(2.0 * x.numel() / t / 1e12, 2)
    t = timed(lambda: x.sum(dtype=torch.int64) if False else x.view(torch.int64).sum(), 10)
    out[f"read_{gb}GiB_TBps"] = round(x.numel() / t / 1e12, 2)
print(json.dumps(out))
