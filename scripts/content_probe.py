#!/usr/bin/env python3
"""GPU box: |dlogit| of the 128x128 arithmetics against the C oracle per CONTENT CLASS and weight set (round 3, VERDICT r2
weak #1: near-threshold partial-flat, one-plane-flat, ramps, dither, low contrast).  Prints max / rms per (seed, kind) for
  default  : calibration + guards (what ships)
  nocal    : fast arithmetic forced, guards on (MLT_FLAG_NO_CALIBRATION)
  raw      : fast arithmetic, no guards (measurement only)
  tier, no guard : whatever tier the calibration picks, flat guard off (with MLT_TUNING=1 MLT_W2_MASK=15: hi+lo weights in every stage)
usage: python scripts/content_probe.py [--seeds 10,11] [--n 8] [--kinds 0,4,6,...]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mltcnn_pkg  # noqa: E402
import oracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="10,11,12,21,23")
    ap.add_argument("--kinds", default="0,4,6,7,8,9,10,11,12")
    ap.add_argument("--n", type=int, default=8)
    ap.add_argument("--size", type=int, default=128)
    a = ap.parse_args()
    pkg = mltcnn_pkg.load()
    pkg.build.build_lib()
    S = a.size
    arch = pkg.synth.arch_for_size(S)
    F = pkg.capi
    modes = [("default", 0), ("nocal", F.FLAG_NO_CALIBRATION), ("raw", F.FLAG_NO_CALIBRATION | F.FLAG_NO_FLAT_GUARD),
             ("tier, no guard", F.FLAG_NO_FLAT_GUARD)]  # the calibrated tier (MLT_TUNING=1 MLT_W2_MASK=15: hi+lo weights everywhere) without the flat guard
    if S != 128:
        modes = [("default", 0), ("fast_small", F.FLAG_FAST_SMALL)]
    for seed in [int(v) for v in a.seeds.split(",")]:
        blob = pkg.weights.synthetic_blob(arch, seed)
        orc = oracle.Oracle(blob)
        ctxs = {name: pkg.MltCnn(device=0, sizes=(S,), blobs={S: blob}, flags=fl) for name, fl in modes}
        ar = ctxs["default"].arithmetic(S)
        print(f"seed {seed}: default arithmetic exact={ar['exact']} calib rms {ar['calib_rms']:.2e} max {ar['calib_max']:.2e}", flush=True)
        for kind in [int(v) for v in a.kinds.split(",")]:
            org, pred = pkg.synth.make_patches(S, a.n, 9000 + kind, kind)
            poc, qp = pkg.synth.make_scalars(a.n, 9000 + kind)
            ref, ref_split = orc.forward(org, pred, poc, qp, threads=8)
            row = []
            for name, _ in modes:
                r0 = ctxs[name].arithmetic(S)["guard_reruns"]
                split, lg = ctxs[name].predict_batch(org, pred, poc, qp)
                d = np.abs(lg - ref)
                rr = ctxs[name].arithmetic(S)["guard_reruns"] - r0
                row.append(f"{name}: max {d.max():.2e} rms {np.sqrt((d * d).mean()):.2e} reruns {rr} splitdiff {int((split != ref_split).sum())}")
            print(f"  {pkg.synth.KIND_NAMES[kind]:20s} " + " | ".join(row), flush=True)
        for c in ctxs.values():
            c.close()


if __name__ == "__main__":
    main()
