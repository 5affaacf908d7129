#!/usr/bin/env python3
"""Generate tests/golden/golden_<S>.json by running the REFERENCE's own PyTorch modules.

Runs ONLY in the build container (needs /root/reference); the reference never travels to
the GPU box -- only the small JSON fixtures (inputs are regenerated from seeds, expected
logits are stored) do.  The reference arch files are imported by path because importing
the `models.archs` package drags in cv2 (absent here):
    mlt-cnn-python/codes/models/archs/mlt_ctu_or_pq_arch.py  -> GapBigMltCtuORPQ  (128)
    mlt-cnn-python/codes/models/archs/mlt_cu_or_pq_arch.py   -> GapBigMltCuORPQ   (64/32/16)
Input preparation mirrors vtm-mlt-cpp/source/Lib/EncoderLib/EncCu.cpp:810-887 (u16 cast,
absdiff, *(float)(1/1023), clip, channels [org, resi], poc/qp as int64 tensors).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tools/gen_golden.py
"""
import hashlib
import importlib.util
import json
import os
import sys

os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
sys.dont_write_bytecode = True

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mltcnn_pkg  # noqa: E402

pkg = mltcnn_pkg.load()
synth, weights = pkg.synth, pkg.weights

REF_ARCH_DIR = "/root/reference/mlt-cnn-python/codes/models/archs"


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def reference_model(arch: int):
    if arch == synth.ARCH_CTU:
        return _load(os.path.join(REF_ARCH_DIR, "mlt_ctu_or_pq_arch.py"), "ref_ctu").GapBigMltCtuORPQ()
    return _load(os.path.join(REF_ARCH_DIR, "mlt_cu_or_pq_arch.py"), "ref_cu").GapBigMltCuORPQ()


def variant_state_dict(arch, weight_seed, variant, size, param=0.0):
    sd = synth.make_state_dict(arch, weight_seed)
    if variant == "tie":
        # decision head (EncCu.cpp:913-919): rows 0 and 1 identical and dominant -> exact tie,
        # torch.argmax must return the FIRST maximal index.
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
        u = synth.uniform(weight_seed, "near_tie", c).astype(np.float32)
        w[1] = w[0]
        w[1, :c] += np.float32(1e-3) * (2.0 * u - 1.0)
        b[1] = b[0] + np.float32(param)   # param: the shift that centres the margins of the case's CUs on zero (found by gen_golden.py, stored in the fixture)
        b[2:] -= 1000.0
    return sd


def prep_input(org, pred):
    o = torch.from_numpy(org.astype(np.uint16).astype(np.float32))
    p = torch.from_numpy(pred.astype(np.uint16).astype(np.float32))
    c = torch.tensor(np.float32(1.0 / 1023))
    x0 = (o * c).clamp(0.0, 1.0)
    x1 = ((o - p).abs() * c).clamp(0.0, 1.0)
    return torch.stack([x0, x1], dim=1)  # [n,2,S,S], ch0 = org, ch1 = resi


CASES = [
    # name, weight_seed, variant, input_seed, kind, n, scalars override
    ("texture", 10, "plain", 1000, synth.KIND_TEXTURE, 6, None),
    ("uniform", 11, "plain", 1001, synth.KIND_UNIFORM, 3, None),
    ("zero_resi", 10, "plain", 1002, synth.KIND_ZERO_RESI, 2, None),
    ("saturated", 10, "plain", 1003, synth.KIND_SATURATED, 1, None),
    ("flat", 12, "plain", 1004, synth.KIND_FLAT, 2, None),
    ("poc_qp_min", 10, "plain", 1005, synth.KIND_TEXTURE, 1, ([0], [17])),
    ("poc_qp_max", 10, "plain", 1006, synth.KIND_TEXTURE, 1, ([1023], [47])),
    ("poc0_features_decide", 13, "plain", 1007, synth.KIND_TEXTURE, 4, ([0, 0, 0, 0], [0, 0, 0, 0])),
    ("argmax_tie", 10, "tie", 1008, synth.KIND_TEXTURE, 2, None),
    # further weight sets / contents (added later; the cases above are unchanged)
    ("texture_w21", 21, "plain", 1010, synth.KIND_TEXTURE, 4, None),
    ("texture_w22", 22, "plain", 1011, synth.KIND_TEXTURE, 4, None),
    ("uniform_w23", 23, "plain", 1012, synth.KIND_UNIFORM, 3, None),
    ("zero_resi_w24", 24, "plain", 1013, synth.KIND_ZERO_RESI, 2, None),
    ("out_of_range_pels", 25, "plain", 1014, synth.KIND_OUT_OF_RANGE, 3, None),
    # round 3: content the flat-content guard used not to see -- near-threshold partial-flat, one plane flat,
    # ramps, +-1 LSB dither, low contrast, zero residual on flat org -- on the bench weight set and on a second one
    ("partial_flat", 10, "plain", 1020, synth.KIND_PARTIAL_FLAT, 3, None),
    ("org_flat_pred_tex", 10, "plain", 1021, synth.KIND_ORG_FLAT_PRED_TEX, 3, None),
    ("org_tex_pred_flat", 10, "plain", 1022, synth.KIND_ORG_TEX_PRED_FLAT, 3, None),
    ("ramp", 10, "plain", 1023, synth.KIND_RAMP, 4, None),
    ("dither", 10, "plain", 1024, synth.KIND_DITHER, 3, None),
    ("low_contrast", 10, "plain", 1025, synth.KIND_LOW_CONTRAST, 3, None),
    ("flat_zero_resi", 10, "plain", 1026, synth.KIND_FLAT_ZERO_RESI, 2, None),
    ("partial_flat_w21", 21, "plain", 1030, synth.KIND_PARTIAL_FLAT, 3, None),
    ("org_flat_pred_tex_w21", 21, "plain", 1031, synth.KIND_ORG_FLAT_PRED_TEX, 3, None),
    ("org_tex_pred_flat_w21", 21, "plain", 1032, synth.KIND_ORG_TEX_PRED_FLAT, 3, None),
    ("ramp_w21", 21, "plain", 1033, synth.KIND_RAMP, 4, None),
    ("dither_w21", 21, "plain", 1034, synth.KIND_DITHER, 3, None),
    ("low_contrast_w21", 21, "plain", 1035, synth.KIND_LOW_CONTRAST, 3, None),
    ("dither_w11", 11, "plain", 1036, synth.KIND_DITHER, 3, None),
    # round 4: near-ties on the decision head (top-2 margins of 1e-5 ... 2e-3, either sign) on the bench weight set (single pass) and on
    # two sets of the hi+lo-weights tiers; the new content classes of the two-level flat guard
    ("near_tie", 10, "near_tie", 1040, synth.KIND_TEXTURE, 16, None),
    ("near_tie_w13", 13, "near_tie", 1041, synth.KIND_TEXTURE, 12, None),
    ("near_tie_w23", 23, "near_tie", 1042, synth.KIND_UNIFORM, 8, None),
    ("partial_near_flat", 10, "plain", 1043, synth.KIND_PARTIAL_NEAR_FLAT, 4, None),
    ("partial_near_flat_w13", 13, "plain", 1044, synth.KIND_PARTIAL_NEAR_FLAT, 4, None),
]


def main():
    torch.set_num_threads(8)
    torch.manual_seed(10)  # EncCu.cpp:890-891 (no effect in eval; kept for fidelity)
    outdir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(outdir, exist_ok=True)
    for size in (128, 64, 32, 16):
        arch = synth.arch_for_size(size)
        model = reference_model(arch).eval()
        ref_keys = list(model.state_dict().keys())
        assert ref_keys == list(synth.state_dict_spec(arch).keys()), "state_dict key order differs from reference"
        for k, v in model.state_dict().items():
            assert tuple(v.shape) == tuple(synth.state_dict_spec(arch)[k]), k
        out = {"size": size, "arch": arch, "generator": "tools/gen_golden.py",
               "reference": "GapBigMltCtuORPQ" if arch == 0 else "GapBigMltCuORPQ",
               "torch": torch.__version__, "cases": []}
        for name, wseed, variant, iseed, kind, n, override in CASES:
            org, pred = synth.make_patches(size, n, iseed, kind)
            if override is None:
                poc, qp = synth.make_scalars(n, iseed)
            else:
                poc, qp = np.array(override[0], np.int32), np.array(override[1], np.int32)
            x = prep_input(org, pred)
            param = 0.0
            if variant == "near_tie":  # centre the top-2 margins of this case's CUs on zero: what is left is +- the feature-dependent part
                sd0 = variant_state_dict(arch, wseed, variant, size, 0.0)
                model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd0.items()}, strict=True)
                model.eval()
                with torch.no_grad():
                    hd = model(x, torch.from_numpy(poc.astype(np.int64)), torch.from_numpy(qp.astype(np.int64)))[2 if size == 128 else 0].numpy()
                param = float(np.float32(-np.median(hd[:, 1] - hd[:, 0])))
            sd = variant_state_dict(arch, wseed, variant, size, param)
            model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
            model.eval()
            with torch.no_grad():
                # one CU per forward, exactly like the encoder (batch 1, EncCu.cpp:869-909)
                per_cu = [model(x[i:i + 1], torch.tensor([int(poc[i])]), torch.tensor([int(qp[i])])) for i in range(n)]
                batched = model(x, torch.from_numpy(poc.astype(np.int64)), torch.from_numpy(qp.astype(np.int64)))
            logits = np.stack([np.concatenate([h[0].numpy() for h in cu]) for cu in per_cu]).astype(np.float32)
            logits_b = np.concatenate([h.numpy() for h in batched], axis=1)
            assert np.abs(logits - logits_b).max() <= 1e-4 * max(1.0, np.abs(logits).max()), "batched vs single"
            argmax = [[int(h[0].argmax().item()) for h in cu] for cu in per_cu]
            out["cases"].append({
                "name": name, "weight_seed": wseed, "variant": variant, **({"variant_param": param} if variant == "near_tie" else {}),
                "input_seed": iseed, "kind": kind, "n": n,
                "poc": [int(v) for v in poc], "qp": [int(v) for v in qp],
                "blob_sha256": hashlib.sha256(weights.pack_blob(arch, sd)).hexdigest(),
                "input_sha256": hashlib.sha256(org.tobytes() + pred.tobytes()).hexdigest(),
                "logits": [[float(v) for v in row] for row in logits],
                "argmax": argmax,
            })
            print(size, name, "max|logit| %.2f" % np.abs(logits).max(), "argmax", argmax)
        with open(os.path.join(outdir, f"golden_{size}.json"), "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
