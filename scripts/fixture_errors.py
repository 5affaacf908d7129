#!/usr/bin/env python3
"""GPU box: per-fixture |dlogit| of the 128x128 model against the committed reference fixtures, per arithmetic
(flags), optionally dumping every layer output of selected fixtures (MLT_DEBUG_DUMP_DIR) for scripts/emul_fast.py.

    python scripts/fixture_errors.py [--flags 0] [--dump saturated,flat --out gpurun_out/dumps]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", type=int, nargs="*", default=[0])
    ap.add_argument("--dump", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "dumps"))
    ap.add_argument("--size", type=int, default=128)
    args = ap.parse_args()
    dump = [d for d in args.dump.split(",") if d]
    import mltcnn_pkg
    from helpers import load_golden, materialise
    pkg = mltcnn_pkg.load()
    pkg.build.build_lib()
    golden = load_golden(args.size)
    for flags in args.flags:
        print(f"== size {args.size} flags {flags}")
        for case in golden["cases"]:
            blob, org, pred, poc, qp, exp, exp_arg = materialise(pkg, golden, case)
            m = pkg.MltCnn(device=0, sizes=(args.size,), blobs={args.size: blob}, flags=flags)
            split, logits = m.predict_batch(org, pred, poc, qp)
            d = np.abs(logits - exp)
            print(f"{case['name']:24s} max {d.max():.2e}  per-logit-max {np.array2string(d.max(axis=0), precision=1, max_line_width=200)}  |logit|max {np.abs(exp).max():.1f}")
            m.close()
    # layer dumps need the env var set before the library's first debug_dump call: run in a child per fixture
    if dump and not os.environ.get("MLT_DEBUG_DUMP_DIR"):
        import subprocess
        for name in dump:
            d = os.path.join(args.out, name)
            os.makedirs(d, exist_ok=True)
            env = dict(os.environ, MLT_TUNING="1", MLT_DEBUG_DUMP_DIR=d, MLT_DUMP_CASE=name)
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--flags", "--size", str(args.size)], env=env)
    if os.environ.get("MLT_DUMP_CASE"):
        case = [c for c in golden["cases"] if c["name"] == os.environ["MLT_DUMP_CASE"]][0]
        blob, org, pred, poc, qp, exp, exp_arg = materialise(pkg, golden, case)
        m = pkg.MltCnn(device=0, sizes=(args.size,), blobs={args.size: blob})
        split, logits = m.predict_batch(org, pred, poc, qp)
        np.save(os.path.join(os.environ["MLT_DEBUG_DUMP_DIR"], "logits.npy"), logits)
        m.close()


if __name__ == "__main__":
    main()
