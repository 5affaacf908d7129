#!/usr/bin/env python3
"""Numerics study aid (CPU, scripts/emul_fast.py): for weight sets the single pass and the hi+lo-weights tiers do not admit (seeds 12, 22,
21), WHICH stages' activation rounding carries the error?  Reference = hi+lo weights everywhere + unrounded activations.  For every subset
of stages: hi+lo weights everywhere + unrounded activations inside the stages of the subset (a stage "in the exact arithmetic"); prints
rms / max |dlogit| over textured + uniform CUs.  Not part of the product or the tests."""
import itertools
import sys

import numpy as np

from emul_fast import ACT_SITES, CONVS, Emul, pkg


def sites_of(stage):
    return [s for s in ACT_SITES if s.startswith(f"l{stage}.")]


def convs_of(stage):
    return [c for c in CONVS if c.startswith(f"l{stage}.")] + (["stem"] if stage == 0 else [])


def main():
    seeds = [int(a) for a in sys.argv[1:]] or [12, 22, 21]
    n = 12
    org, pred = pkg.synth.make_patches(128, n, 7001, pkg.synth.KIND_TEXTURE)
    o2, p2 = pkg.synth.make_patches(128, 4, 7002, pkg.synth.KIND_UNIFORM)
    org = np.concatenate([org, o2]); pred = np.concatenate([pred, p2])
    poc, qp = pkg.synth.make_scalars(n + 4, 7001)
    for seed in seeds:
        blob = pkg.weights.synthetic_blob(0, seed)
        ref = Emul(blob, {"act_default": "f32", "w_hilo": CONVS}).forward(org, pred, poc, qp)
        print(f"seed {seed}: |logit| max {np.abs(ref).max():.1f}")
        rows = []
        for k in range(0, 5):
            for sub in itertools.combinations(range(4), k):
                act = {s: "f32" for st in sub for s in sites_of(st)}
                # W: hi+lo everywhere (upper bound of what weights can do) / only in the stages of the subset
                for wmode, wh in (("w2 all", CONVS), ("w2 sub", [c for st in sub for c in convs_of(st)])):
                    got = Emul(blob, {"act": act, "w_hilo": wh}).forward(org, pred, poc, qp)
                    d = got - ref
                    rows.append((sub, wmode, float(np.sqrt((d ** 2).mean())), float(np.abs(d).max())))
        for sub, wmode, rms, mx in rows:
            ok = 5.5 * rms <= 1e-3 and mx <= 0.75e-3
            print(f"  exact stages {str(sub):14s} {wmode}: rms {rms:.2e} max {mx:.2e} {'ADMIT' if ok else ''}")


if __name__ == "__main__":
    main()
