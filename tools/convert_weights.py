#!/usr/bin/env python3
"""Checkpoint -> MLTW blob converter (the counterpart of the reference's model2torchScript.py).

The reference turns `net_<iter>.pth` ({'params': state_dict}, optional 'module.' prefixes,
mlt-cnn-python/codes/model2torchScript.py:23-32) into a TorchScript `.pt` that the encoder re-loads per CU.
This tool turns the same checkpoint -- or a TorchScript file saved by that script -- into
`MLTORPQ_splitMode_<S>.mltw`, read once by mlt_init().

  python tools/convert_weights.py --size 128 net_310000.pth  out_dir/
  python tools/convert_weights.py --size 64 MLTORPQ_splitMode_64.pt out_dir/
  python tools/convert_weights.py --size 128 --synthetic 10 out_dir/       (seeded synthetic weights, tests/bench)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mltcnn_pkg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, required=True, choices=(128, 64, 32, 16))
    ap.add_argument("--synthetic", type=int, default=None, metavar="SEED")
    ap.add_argument("paths", nargs="+", help="[checkpoint] out_dir")
    args = ap.parse_args()
    pkg = mltcnn_pkg.load()
    arch = pkg.synth.arch_for_size(args.size)
    out_dir = args.paths[-1]
    if args.synthetic is not None:
        blob = pkg.weights.synthetic_blob(arch, args.synthetic)
    else:
        import torch
        src = args.paths[0]
        # `.pth` = {'params': state_dict} of plain tensors: weights_only=True never unpickles arbitrary objects from a
        # third-party checkpoint.  A TorchScript archive (model2torchScript.py:46-48) is refused by it and goes to jit.load.
        try:
            obj = torch.load(src, map_location="cpu", weights_only=True)
        except Exception:
            obj = torch.jit.load(src, map_location="cpu")
        if hasattr(obj, "state_dict"):
            obj = obj.state_dict()
        blob = pkg.weights.from_checkpoint(obj, arch)
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"MLTORPQ_splitMode_{args.size}.mltw")
    with open(path, "wb") as f:
        f.write(blob)
    print(f"wrote {path} ({len(blob)} bytes)")


if __name__ == "__main__":
    main()
