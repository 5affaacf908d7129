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


def variant_state_dict(pkg, arch, weight_seed, variant, size):
    """Same construction as tools/gen_golden.py (kept in sync through blob_sha256)."""
    sd = pkg.synth.make_state_dict(arch, weight_seed)
    if variant == "tie":
        h = 3 if size == 128 else 1
        w, b = sd[f"branch{h}.weight"], sd[f"branch{h}.bias"]
        w[1] = w[0]
        b[1] = b[0]
        b[2:] -= 1000.0
    return sd


def materialise(pkg, golden, case):
    """-> (blob, org, pred, poc, qp, expected_logits, expected_argmax) for one golden case."""
    size, arch = golden["size"], golden["arch"]
    sd = variant_state_dict(pkg, arch, case["weight_seed"], case["variant"], size)
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
