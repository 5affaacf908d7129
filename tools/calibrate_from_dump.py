#!/usr/bin/env python3
"""Calibrate the arithmetic of a weight set on the INTEGRATOR's content: feed the CUs an encoder run recorded (host/mlt_split_predictor.hpp's
call dump, MLTCNN_CALL_DUMP_FILE in a -DMLTCNN_TEST_HOOKS build: the org / pred planes out of VTM's picture buffers, POC, CU QP of every
predictSplitMode call) to mlt_calibrate and report the tier before and after (mlt_arithmetic).  GPU box only -- there is no CPU fallback.

  python tools/calibrate_from_dump.py calls.bin --weights-dir torch_model [--size 128] [--replace] [--max 4096] [--flags 0]

An integrator runs this once per weight set and content family; the encoder then calls mlt_calibrate(ctx, size, ...) with the same CUs after
mlt_init (or ships the outcome as a policy: MLT_FLAG_EXACT_128 when the content does not admit the faster tiers)."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def calibrate(pkg, dump_path, weights_dir, size=128, replace=False, max_cus=4096, flags=0):
    from run_ra_eval import read_call_dump
    calls = [c for c in read_call_dump(dump_path) if c["cuw"] == size]
    if not calls:
        raise SystemExit(f"{dump_path} holds no {size}x{size} call")
    if len(calls) > max_cus:   # an even sample over the run (all pictures, all QPs)
        calls = [calls[i] for i in np.linspace(0, len(calls) - 1, max_cus).astype(int)]
    org = np.stack([c["org"] for c in calls]); pred = np.stack([c["pred"] for c in calls])
    poc = np.array([c["poc"] for c in calls], np.int32); qp = np.array([c["qp"] for c in calls], np.int32)
    m = pkg.MltCnn(device=0, sizes=(size,), weights_dir=weights_dir, flags=flags)
    before = m.arithmetic(size)
    m.calibrate(size, org, pred, poc, qp, replace=replace)
    after = m.arithmetic(size)
    s, lg = m.predict_batch(org, pred, poc, qp)
    m.close()
    return {"dump": os.path.basename(dump_path), "size": size, "cus_in_dump": len(calls), "mode": "replace" if replace else "append", "before": before, "after": after,
            "tier_changed": (before["exact"], before["w2_units"], before["x_units"], before["rounding"]) != (after["exact"], after["w2_units"], after["x_units"], after["rounding"])}, (org, pred, poc, qp, s, lg)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dump")
    ap.add_argument("--weights-dir", required=True)
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--replace", action="store_true")
    ap.add_argument("--max", type=int, default=4096)
    ap.add_argument("--flags", type=lambda v: int(v, 0), default=0)
    a = ap.parse_args()
    import mltcnn_pkg
    pkg = mltcnn_pkg.load()
    rep, _ = calibrate(pkg, a.dump, a.weights_dir, a.size, a.replace, a.max, a.flags)
    print(json.dumps(rep, indent=1))


if __name__ == "__main__":
    main()
