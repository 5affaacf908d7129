#!/usr/bin/env python3
"""GPU box: the natural-statistics content class (synth.natural_patches) through the single-pass arithmetic WITHOUT guards against the C
oracle, binned by the flat-content guard's own statistic (fraction of coherent quads): how large the error of the CUs the guard flags
really is, what fraction it flags, and what the shipped configuration (guards on) delivers.  usage: natural_probe.py [--seeds 10,11] [--n 512]"""
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
    ap.add_argument("--seeds", default="10,11")
    ap.add_argument("--n", type=int, default=512)
    a = ap.parse_args()
    pkg = mltcnn_pkg.load()
    S, F = 128, pkg.capi
    org, pred = pkg.synth.natural_patches(S, a.n, 0xBEEF)
    poc, qp = pkg.synth.make_scalars(a.n, 0xBEEF)
    frac = pkg.synth.flat_quad_fraction(org, pred)
    sd = org.astype(np.float64).std(axis=(1, 2))
    print(f"natural set: {a.n} CUs, flagged by the guard (coherent quads >= 1/8): {(frac >= 0.125).mean():.3f}; org std quantiles 10/50/90 %: "
          + " ".join(f"{v:.0f}" for v in np.quantile(sd, [.1, .5, .9])))
    bins = [(0, 0.01), (0.01, 0.125), (0.125, 0.5), (0.5, 0.9), (0.9, 1.01)]
    for seed in [int(v) for v in a.seeds.split(",")]:
        blob = pkg.weights.synthetic_blob(0, seed)
        ref, ref_split = oracle.Oracle(blob).forward(org, pred, poc, qp, threads=os.cpu_count())
        for name, fl in (("raw single pass", F.FLAG_NO_CALIBRATION | F.FLAG_NO_FLAT_GUARD), ("shipped", 0), ("shipped + decision guard", F.FLAG_DECISION_GUARD)):
            m = pkg.MltCnn(device=0, sizes=(S,), blobs={S: blob}, flags=fl)
            ar = m.arithmetic(S)
            s, l = m.predict_batch(org, pred, poc, qp)
            e = np.abs(l - ref).max(axis=1)
            rr = m.arithmetic(S)["guard_reruns"]
            row = " | ".join(f"[{lo:.2f},{hi:.2f}) n={int(((frac >= lo) & (frac < hi)).sum())} max {e[(frac >= lo) & (frac < hi)].max() if ((frac >= lo) & (frac < hi)).any() else 0:.1e}"
                            for lo, hi in bins)
            print(f"seed {seed} {name} (tier {ar['exact']} stages 0x{ar['w2_stages']:x}): max {e.max():.2e} rms {np.sqrt((np.abs(l - ref) ** 2).mean()):.2e} "
                  f"reruns {rr} split mismatches {int((s != ref_split).sum())} | by coherent-quad fraction: {row}", flush=True)
            m.close()


if __name__ == "__main__":
    main()
