#!/usr/bin/env python3
"""Where does the single pass's logit error of a weight set come from?  

CPU only.  Runs the library's OWN synthetic calibration set (the 560 CUs mlt_load_weights prices; host-only hook mlt_calibration_set_copy) --
or natural-statistics CUs -- through scripts/emul_fast.py's emulation of the single pass (every rounding point of the kernels) and through the
same graph without any rounding, and attributes the difference:

  1. per content class: rms / max |dlogit|, and the worst CUs by name (class, index, head, logit);
  2. for the worst CUs, per ROUNDING SITE (the 18 fp16 activation tensors + the weights): the logit error with ONLY that site rounded, and with
     every site BUT that one rounded (non-additivity shows as a gap between the two views);
  3. for the worst CU and its heaviest sites: the rounding error's spatial statistics per channel -- |mean over pixels| against rms / sqrt(HW),
     i.e. how COHERENT the error is (global pooling averages incoherent errors away, coherent ones go straight into the features) -- and the
     fraction of pixels whose value sits in the same fp16 binade / rounds in the same direction.

  python tools/attribute_error.py BLOB.mltw [--set calib|natural] [--n N] [--worst K] [--strategies base,dither,...] [--threads T]

Strategies (scripts/emul_fast.py STRATEGIES + those defined here) are candidate cures priced on the same CUs: prints their rms / max per class."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

CLASS_NAMES = ["texture", "uniform", "org const", "pred const", "const band", "near-flat band", "caller"]
HEADS = ((0, 2), (2, 5), (5, 9))


def calibration_set(pkg, size=128):
    lib = pkg.capi.load_library()
    lib.mlt_calibration_set_copy.argtypes = [C.c_int] + [C.c_void_p] * 5
    n = 560
    org = np.zeros((n, size, size), np.int16); pred = np.zeros_like(org)
    poc = np.zeros(n, np.int32); qp = np.zeros(n, np.int32); cls = np.zeros(n, np.int32)
    assert lib.mlt_calibration_set_copy(size, org.ctypes.data, pred.ctypes.data, poc.ctypes.data, qp.ctypes.data, cls.ctypes.data) == n
    return org, pred, poc, qp, cls


def batched(em, org, pred, poc, qp, bs=80):
    out = []
    for i in range(0, len(org), bs):
        out.append(em.forward(org[i:i + bs], pred[i:i + bs], poc[i:i + bs], qp[i:i + bs]))
    return np.concatenate(out)


def summarise(name, d, cls):
    rows = []
    for c in sorted(set(cls.tolist())):
        e = d[cls == c]
        rows.append(f"{CLASS_NAMES[c]}: rms {np.sqrt((e ** 2).mean()):.2e} max {np.abs(e).max():.2e}")
    rms_all = np.sqrt((d ** 2).mean())
    per_head = " ".join(f"{np.sqrt((d[:, a:b] ** 2).mean()):.2e}" for a, b in HEADS)
    print(f"[{name}] all: rms {rms_all:.2e} max {np.abs(d).max():.2e} = {np.abs(d).max() / max(rms_all, 1e-30):.1f} x rms | per head rms {per_head}")
    print("    " + " | ".join(rows))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("blob")
    ap.add_argument("--set", default="calib", choices=["calib", "natural"])
    ap.add_argument("--n", type=int, default=0, help="CUs (calib: the first n of every class when > 0)")
    ap.add_argument("--worst", type=int, default=6)
    ap.add_argument("--strategies", default="")
    ap.add_argument("--threads", type=int, default=4)
    ap.add_argument("--no-sites", action="store_true")
    a = ap.parse_args()
    import torch
    import emul_fast as E
    torch.set_num_threads(a.threads)
    pkg = E.pkg
    blob = open(a.blob, "rb").read()
    if a.set == "calib":
        org, pred, poc, qp, cls = calibration_set(pkg)
        if a.n:
            keep = np.concatenate([np.flatnonzero(cls == c)[:a.n] for c in range(6)])
            org, pred, poc, qp, cls = org[keep], pred[keep], poc[keep], qp[keep], cls[keep]
    else:
        n = a.n or 256
        org, pred = pkg.synth.natural_patches(128, n, 777)
        poc, qp = pkg.synth.make_scalars(n, 777)
        cls = np.full(n, 0, np.int32)
    print(f"{a.blob}: {len(org)} CUs ({a.set})")
    t0 = time.time()
    ref = batched(E.Emul(blob, {"act_default": "f32", "w_hilo": E.CONVS}), org, pred, poc, qp)
    print(f"reference (no activation rounding, hi+lo weights): |logit| max {np.abs(ref).max():.1f} rms {np.sqrt((ref ** 2).mean()):.2f}  ({time.time() - t0:.0f} s)")
    try:
        import oracle
        k = min(16, len(org))
        oref, _ = oracle.Oracle(blob).forward(org[:k], pred[:k], poc[:k], qp[:k], threads=a.threads)
        print(f"reference vs the C oracle on {k} CUs: max {np.abs(ref[:k] - oref).max():.2e}")
    except Exception as e:  # the oracle library may not be built
        print("oracle check skipped:", e)
    base = batched(E.Emul(blob, {}), org, pred, poc, qp)
    d = base - ref
    summarise("single pass", d, cls)
    for nm, opts in (("hi+lo weights everywhere", {"w_hilo": E.CONVS}), ("no activation rounding, single-pass weights", {"act_default": "f32"})):
        summarise(nm, batched(E.Emul(blob, opts), org, pred, poc, qp) - ref, cls)
    for nm in [s for s in a.strategies.split(",") if s]:
        summarise(nm, batched(E.Emul(blob, E.STRATEGIES[nm]), org, pred, poc, qp) - ref, cls)
    # worst CUs
    per_cu = np.abs(d).max(axis=1)
    order = np.argsort(-per_cu)[:a.worst]
    print("\nworst CUs (single pass):")
    for i in order:
        j = int(np.abs(d[i]).argmax())
        print(f"  CU {i:4d} class {CLASS_NAMES[cls[i]]:15s} |d| {per_cu[i]:.2e} at logit {j} (head {[h for h, (x, y) in enumerate(HEADS) if x <= j < y][0] + 1}), ref {ref[i, j]:+.3f}; "
              f"org span {int(org[i].max()) - int(org[i].min())}, |org-pred| mean {np.abs(org[i].astype(int) - pred[i].astype(int)).mean():.1f}")
    if a.no_sites:
        return
    w = order
    o, p, pc, q = org[w], pred[w], poc[w], qp[w]
    rw = ref[w]
    print("\nper rounding site, worst CUs: max |dlogit| with ONLY the site rounded / with all BUT the site rounded (weights hi+lo in both views)")
    all_act = batched(E.Emul(blob, {"w_hilo": E.CONVS}), o, p, pc, q) - rw
    print(f"  {'all activation sites':14s} " + " ".join(f"{np.abs(all_act[k]).max():.2e}" for k in range(len(w))))
    for s in E.ACT_SITES:
        only = batched(E.Emul(blob, {"act_default": "f32", "act": {s: "rn"}, "w_hilo": E.CONVS}), o, p, pc, q) - rw
        but = batched(E.Emul(blob, {"act": {s: "f32"}, "w_hilo": E.CONVS}), o, p, pc, q) - rw
        print(f"  {s:14s} " + " ".join(f"{np.abs(only[k]).max():.1e}/{np.abs(but[k]).max():.1e}" for k in range(len(w))))
    # spatial statistics of the rounding error at every site, worst CU
    i0 = int(order[0])
    print(f"\nrounding error statistics per site, CU {i0} (class {CLASS_NAMES[cls[i0]]}): per channel, coherent part |mean_px e| vs incoherent floor rms_px e / sqrt(HW)")
    rec = {}
    E.Emul(blob, {"w_hilo": E.CONVS, "record": rec}).forward(org[i0:i0 + 1], pred[i0:i0 + 1], poc[i0:i0 + 1], qp[i0:i0 + 1])
    for s in E.ACT_SITES:
        x, y = rec[s]
        e = (y - x)[0].numpy()                     # [C, H, W]
        hw = e.shape[1] * e.shape[2]
        mean = np.abs(e.mean(axis=(1, 2)))
        rms = np.sqrt((e ** 2).mean(axis=(1, 2)))
        floor = rms / np.sqrt(hw)
        xs = x[0].numpy()
        live = (xs > 0).mean()
        # channels whose live pixels all sit in ONE fp16 binade (same ulp) and whose spread is below a few ulp: "spatially constant" activations
        spread = []
        for c in range(xs.shape[0]):
            v = xs[c][xs[c] > 0]
            if v.size < 16:
                continue
            ulp = 2.0 ** (np.floor(np.log2(np.median(v))) - 10)
            spread.append((np.percentile(v, 90) - np.percentile(v, 10)) / ulp)
        spread = np.array(spread) if spread else np.array([np.inf])
        print(f"  {s:10s} C {e.shape[0]:3d} HW {hw:5d} live {live:.2f} | coherent rms_c {np.sqrt((mean ** 2).mean()):.2e} (max {mean.max():.2e}) vs floor {np.sqrt((floor ** 2).mean()):.2e}"
              f" -> ratio {np.sqrt((mean ** 2).mean()) / max(np.sqrt((floor ** 2).mean()), 1e-30):5.1f} | channels with 10-90 % spread < 4 ulp: {(spread < 4).sum()} of {spread.size}")


if __name__ == "__main__":
    main()
