#!/usr/bin/env python3
"""GPU box: how far out do the tails of |dlogit| go?  The load-time calibration admits an arithmetic on 96 CUs and a Gaussian tail factor
(5.5 sigma); this measures the tail itself: N CUs per content class through the shipped tier of a weight set and through the exact
arithmetic (which sits within 1e-5 of the fp32 oracle), max / rms / count of |dlogit| > 1e-3 and the ratio max / rms per class and head.
usage: python scripts/tail_probe.py [--seeds 10,23,13] [--n 4096] [--n-texture 16384]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mltcnn_pkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="10,23,13")
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--n-texture", type=int, default=16384)
    ap.add_argument("--size", type=int, default=128, help="CU size (round 4: the small models' calibrated tiers too)")
    ap.add_argument("--blobs", default="", help="comma list of MLTW files probed like seeds (round 5: tools/train_synth_weights.py's trained family)")
    ap.add_argument("--natural", type=int, default=0, help="also N CUs of the natural-statistics class (synth.natural_patches)")
    ap.add_argument("--near-flat", type=int, default=0, help="also N CUs each of the classes around the flat guard's near-flat rule (round 6: low-contrast texture -- not "
                    "flagged since MLT_FLAT_RANGE is 4 --, +-1 LSB dither, a near-flat band under 1/2 of the quads)")
    a = ap.parse_args()
    pkg = mltcnn_pkg.load()
    S, size = pkg.synth, a.size
    arch = S.arch_for_size(size)
    classes = [("texture", None), ("uniform", S.KIND_UNIFORM), ("org_flat_pred_tex", S.KIND_ORG_FLAT_PRED_TEX),
               ("org_tex_pred_flat", S.KIND_ORG_TEX_PRED_FLAT), ("partial_flat", S.KIND_PARTIAL_FLAT)]
    if a.near_flat:
        classes += [("low_contrast", S.KIND_LOW_CONTRAST), ("dither", S.KIND_DITHER), ("partial_near_flat", S.KIND_PARTIAL_NEAR_FLAT), ("low_contrast_5_6", -2)]
    data = {}
    if a.natural:
        org, pred = S.natural_patches(size, a.natural, 31337)
        poc, qp = S.make_scalars(a.natural, 31337 + 99)
        data["natural"] = (org, pred, poc, qp)
    for name, kind in classes:
        n = a.n_texture if kind is None else (a.near_flat if name in ("low_contrast", "dither", "partial_near_flat", "low_contrast_5_6") else a.n)
        if kind == -2:   # texture of amplitude 5 / 6 on a constant base (KIND_LOW_CONTRAST is amplitude 4): 45 - 60 % of its quads have a range <= 8, 25 - 40 % <= 6 --
            rng = np.random.default_rng(31337 + 1000)   # the content MLT_FLAT_RANGE = 6 stops flagging for good
            amp = 5 + (np.arange(n) % 2)[:, None, None]
            org = (rng.integers(8, 1016, size=(n, 1, 1)) + rng.integers(-6, 7, size=(n, size, size)).clip(-amp, amp)).clip(0, 1023).astype(np.int16)
            pred = (org + rng.integers(-2, 3, size=(n, size, size))).clip(0, 1023).astype(np.int16)
            poc, qp = S.make_scalars(n, 31337 + 1000)
            data[name] = (org, pred, poc, qp)
            print(f"generated {name}: {n} CUs (near-flat fraction at range 8 / 6: {S.flat_quad_fraction(org[:64], pred[:64], 8).mean():.2f} / {S.flat_quad_fraction(org[:64], pred[:64], 6).mean():.2f})", flush=True)
            continue
        org, pred = S.make_patches_bulk(size, n, 31337) if kind is None else S.make_patches(size, n, 31337 + kind, kind)
        poc, qp = S.make_scalars(n, 31337 + (kind or 0))
        data[name] = (org, pred, poc, qp)
        print(f"generated {name}: {n} CUs", flush=True)
    heads = [slice(0, 2), slice(2, 5), slice(5, 9)] if size == 128 else [slice(0, 2), slice(2, 5), slice(5, 9), slice(9, 15)]
    dec = 2 if size == 128 else 0  # decision head (EncCu.cpp:913-919)
    if a.natural:
        classes = classes + [("natural", -1)]
    for seed in [int(v) for v in a.seeds.split(",") if v] + [b for b in a.blobs.split(",") if b]:
        blob = open(seed, "rb").read() if isinstance(seed, str) else pkg.weights.synthetic_blob(arch, seed)
        m = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob})
        e = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob}, flags=pkg.capi.FLAG_EXACT_128 if size == 128 else pkg.capi.FLAG_NO_CALIBRATION)
        ar = m.arithmetic(size)
        print(f"seed {seed}: tier {ar['exact']} (0 fast, 3 hi+lo weights in stages 0x{ar['w2_stages']:x}, 2 hi+lo weights everywhere, 4 exact in stages 0x{ar['x_stages']:x} + hi+lo weights in 0x{ar['w2_stages']:x}, 1 exact), exact units 0x{ar['x_units']:x}, hi+lo units 0x{ar['w2_units']:x}; calibration worst rms {ar['calib_rms']:.2e} max {ar['calib_max']:.2e}; magnitude guard threshold {ar['mag_guard_thr']:.3g} (0 = none; {100 * ar['mag_guard_flagged']:.1f} % of the in-distribution calibration CUs above it)", flush=True)
        tot_n = tot_bad = 0
        worst = 0.0
        for name, _ in classes:
            org, pred, poc, qp = data[name]
            r0 = m.arithmetic(size)["guard_reruns"]
            s1, l1 = m.predict_batch(org, pred, poc, qp)
            s2, l2 = e.predict_batch(org, pred, poc, qp)
            d = np.abs(l1.astype(np.float64) - l2)
            rms = np.sqrt((d * d).mean())
            per_head = [np.sqrt((d[:, h] ** 2).mean()) for h in heads]
            bad = int((d > 1e-3).sum())
            srt = np.sort(l2[:, heads[dec]].astype(np.float64), axis=1)
            decisive = (srt[:, -1] - srt[:, -2]) > 2e-3
            flips = int(((s1 != s2) & decisive).sum())
            tot_n += d.size; tot_bad += bad; worst = max(worst, float(d.max()))
            print(f"  {name:18s} n {len(org):6d}  max {d.max():.2e}  rms {rms:.2e}  max/rms {d.max() / rms:4.1f}  per head rms " +
                  " ".join(f"{v:.2e}" for v in per_head) + f"  |d|>1e-3: {bad}  decisive split flips {flips}  guard re-runs {m.arithmetic(size)['guard_reruns'] - r0}", flush=True)
        print(f"  => {tot_n} logits, {tot_bad} beyond 1e-3, worst {worst:.2e}", flush=True)
        m.close(); e.close()


if __name__ == "__main__":
    main()
