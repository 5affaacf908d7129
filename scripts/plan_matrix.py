#!/usr/bin/env python3
"""CPU: dump the dispatcher's launch plans (mlt_plan_describe, detailed records: every launch with its variant and its buffers) over a wide matrix of
(size, batch, tier, launch-unit masks, alignment) -- the regression check of a dispatcher change: `python scripts/plan_matrix.py > before.txt`, change,
rebuild, `python scripts/plan_matrix.py | diff before.txt -`."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mltcnn_pkg

pkg = mltcnn_pkg.load()
lib = pkg.capi.load_library()
lib.mlt_plan_describe.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_int, C.c_char_p, C.c_size_t]
blobs = {a: pkg.weights.synthetic_blob(a, 10) for a in (0, 1)}
NS = (1, 2, 3, 8, 16, 31, 33, 64, 100, 127, 128, 129, 200, 255, 256, 500, 1024, 2048, 4096)
buf = C.create_string_buffer(1 << 18)
count = 0
for size in (128, 64, 32, 16):
    b = blobs[pkg.synth.arch_for_size(size)]
    units = 8 if size == 128 else 10
    cases = [(0, 0, 0), (1, 0, 0), (5, 0, 0)]
    masks = [1 << u for u in range(units)] + [0x3, 0xC, 0x30, 0xC0, 0xF0, 0xFC, 0xFF, 0x0F, 0x3C, 0xAA, 0x55, (1 << units) - 1, ((1 << units) - 1) & ~1, ((1 << units) - 1) & ~3]
    for mk in masks:
        cases.append((2, mk, 0))
        cases.append((4, 0, mk))
    for w2, x in ((0xC, 0xF0), (0x3, 0xC), (0xF0, 0x0C), (0x30, 0xC0), (0xC0, 0x30), (0x0F, 0xF0), (0x2, 0x1), (0x1, 0x2), (0xFC, 0x3)):
        cases.append((4, w2, x))
    for tier, w2, x in cases:
        for n in NS:
            for al in (3, 2):
                k = lib.mlt_plan_describe(b, len(b), size, n, tier, w2, x, al, buf, 1 << 18)
                print(f"== size {size} n {n} tier {tier} w2 0x{w2:x} x 0x{x:x} aligned {al & 1}: {k}")
                if k > 0:
                    print(buf.value.decode(), end="")
                count += 1
print("cases", count, file=sys.stderr)
