#!/usr/bin/env python3
"""Synthetic 10-bit 4:2:0 YUV for the N1 integration test (no sequences exist in the container): a textured background that
moves by (2, 1) luma pixels per frame (so merge / skip predictions are meaningful), one exactly-constant CTU (the flat-content
guard's case), one smooth ramp CTU, and a slowly brightening low-contrast CTU.  Planar, 16-bit little-endian samples, as VTM's
-i expects with InputBitDepth 10.

usage: python tools/make_synth_yuv.py out.yuv [--width 384 --height 256 --frames 3 --seed 7]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_frames(width=384, height=256, frames=3, seed=7):
    import mltcnn_pkg
    synth = mltcnn_pkg.load().synth
    W, H = width + 64, height + 64                      # margin for the motion
    nb = 16
    base = synth.randint(seed, "yuv/base", (H // nb + 1) * (W // nb + 1), 100, 900).reshape(H // nb + 1, W // nb + 1)
    base = np.kron(base, np.ones((nb, nb), np.int64))[:H, :W]
    tex = synth.randint(seed, "yuv/tex", H * W, -40, 40).reshape(H, W)
    world = np.clip(base + tex, 0, 1023)
    cb = np.clip(512 + synth.randint(seed, "yuv/cb", (H // 2) * (W // 2), -60, 60).reshape(H // 2, W // 2), 0, 1023)
    cr = np.clip(512 + synth.randint(seed, "yuv/cr", (H // 2) * (W // 2), -60, 60).reshape(H // 2, W // 2), 0, 1023)
    out = []
    for f in range(frames):
        dx, dy = 2 * f, f
        y = world[dy:dy + height, dx:dx + width].copy()
        u = cb[dy // 2:dy // 2 + height // 2, dx // 2:dx // 2 + width // 2].copy()
        v = cr[dy // 2:dy // 2 + height // 2, dx // 2:dx // 2 + width // 2].copy()
        if width >= 256 and height >= 128:
            y[0:128, 128:256] = 600                                                   # exactly-constant CTU (static)
            u[0:64, 64:128] = 512
            v[0:64, 64:128] = 512
        if width >= 384 and height >= 128:
            yy, xx = np.mgrid[0:128, 0:128]
            y[0:128, 256:384] = np.clip(200 + 3 * xx + 2 * yy + 4 * f, 0, 1023)       # smooth ramp, brightening
        if height >= 256:
            lc = 400 + 6 * f + synth.randint(seed, f"yuv/lc{f}", 128 * 128, -3, 3).reshape(128, 128)
            y[128:256, 0:128] = lc                                                    # low-contrast CTU
        out.append((y.astype("<u2"), u.astype("<u2"), v.astype("<u2")))
    return out


def write_yuv(path, frames):
    with open(path, "wb") as fh:
        for y, u, v in frames:
            fh.write(y.tobytes()); fh.write(u.tobytes()); fh.write(v.tobytes())


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--width", type=int, default=384)
    ap.add_argument("--height", type=int, default=256)
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--seed", type=int, default=7)
    a = ap.parse_args()
    write_yuv(a.out, make_frames(a.width, a.height, a.frames, a.seed))
    print(f"wrote {a.out}: {a.width}x{a.height}, {a.frames} frames, 10-bit 4:2:0")
