"""Shared helpers for the parity tests: golden fixtures -> regenerated inputs."""
import hashlib
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SIZES = (128, 64, 32, 16)


def load_golden(size):
    with open(os.path.join(GOLDEN_DIR, f"golden_{size}.json")) as f:
        return json.load(f)


def variant_state_dict(pkg, arch, weight_seed, variant, size, param=0.0):
    """Same construction as tools/gen_golden.py (kept in sync through blob_sha256)."""
    sd = pkg.synth.make_state_dict(arch, weight_seed)
    if variant == "tie":
        h = 3 if size == 128 else 1
        w, b = sd[f"branch{h}.weight"], sd[f"branch{h}.bias"]
        w[1] = w[0]
        b[1] = b[0]
        b[2:] -= 1000.0
    if variant == "near_tie":
        # decision head: row 1 = row 0 + a small feature-dependent perturbation, the other rows far below -> the top-2 margin of every CU
        # is a small random number (|margin| ~ 1e-5 ... 2e-3, either sign): the CUs a decision guard exists for (round 4)
        h = 3 if size == 128 else 1
        w, b = sd[f"branch{h}.weight"], sd[f"branch{h}.bias"]
        c = w.shape[1] - 2
        u = pkg.synth.uniform(weight_seed, "near_tie", c).astype(np.float32)
        w[1] = w[0]
        w[1, :c] += np.float32(1e-3) * (2.0 * u - 1.0)
        b[1] = b[0] + np.float32(param)   # param: the shift that centres the margins of the case's CUs on zero (found by gen_golden.py, stored in the fixture)
        b[2:] -= 1000.0
    return sd


def materialise(pkg, golden, case):
    """-> (blob, org, pred, poc, qp, expected_logits, expected_argmax) for one golden case."""
    size, arch = golden["size"], golden["arch"]
    sd = variant_state_dict(pkg, arch, case["weight_seed"], case["variant"], size, case.get("variant_param", 0.0))
    blob = pkg.weights.pack_blob(arch, sd)
    assert hashlib.sha256(blob).hexdigest() == case["blob_sha256"], "weight generator drifted from the fixtures"
    org, pred = pkg.synth.make_patches(size, case["n"], case["input_seed"], case["kind"])
    assert hashlib.sha256(org.tobytes() + pred.tobytes()).hexdigest() == case["input_sha256"], "input generator drifted"
    poc = np.array(case["poc"], np.int32)
    qp = np.array(case["qp"], np.int32)
    return blob, org, pred, poc, qp, np.array(case["logits"], np.float32), case["argmax"]


def head_slices(head_classes):
    out, lo = [], 0
    for c in head_classes:
        out.append(slice(lo, lo + c))
        lo += c
    return out


def decisive(logits_row, sl, margin):
    """True when the top-2 gap of this head exceeds `margin` (argmax comparison is meaningful)."""
    v = np.sort(np.asarray(logits_row[sl], np.float64))[::-1]
    return (v[0] - v[1]) > margin


EXACT_NOISE = 2e-5  # |dlogit| of the exact (hi, lo) arithmetic against the fp32 reference / oracle (measured <= 1.5e-5 on every fixture)


def check_splits(split, ref_logits, ref_split, sl, guarded, tol=1e-3, what=""):
    """EVERY CU's split mode against the reference's -- no CU is skipped silently; returns the number of CUs that cannot be decided.

    guarded = True: the decision guard is on (or the arithmetic is the exact one).  A CU whose fast top-2 margin is >= 2 x tol keeps
        its fast split -- two logits within tol of the reference cannot swap over such a margin -- and every other CU is re-evaluated
        exactly (|dlogit| <= EXACT_NOISE).  So the split must equal the reference's for EVERY CU whose reference margin exceeds
        2 x EXACT_NOISE; the CUs below that are ties of the reference's own fp32 arithmetic and are counted, not skipped.
    guarded = False: an arithmetic with errors up to tol and no guard: equality is guaranteed only above a reference margin of 2 x tol."""
    floor = 2 * EXACT_NOISE if guarded else 2 * tol
    undecided = 0
    for i in range(len(split)):
        v = np.sort(np.asarray(ref_logits[i][sl], np.float64))[::-1]
        if v[0] - v[1] > floor:
            assert split[i] == ref_split[i], (what, i, int(split[i]), int(ref_split[i]), float(v[0] - v[1]))
        else:
            undecided += 1
    return undecided
