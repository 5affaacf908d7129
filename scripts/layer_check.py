#!/usr/bin/env python3
"""Bring-up aid (GPU box): run one small batch with MLT_DEBUG_DUMP_DIR set and compare every dumped
activation tensor against a torch-CPU emulation of the SAME pipeline (fp32 math on the folded fp32
weights), printing max abs error per layer.  Runs the UNFUSED layer0 path (MLT_NO_BLOCK_FUSION) so that every
conv output exists in HBM.  Not part of the product or the tests."""
import os
import sys
import tempfile

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mltcnn_pkg  # noqa: E402


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    dump = tempfile.mkdtemp(prefix="mltdump_")
    os.environ["MLT_TUNING"] = "1"
    os.environ["MLT_DEBUG_DUMP_DIR"] = dump
    os.environ["MLT_NO_BLOCK_FUSION"] = "1"  # one launch (and one dumped tensor) per conv: the fused layer0 kernels keep t / sc on chip
    pkg = mltcnn_pkg.load()
    pkg.build.build_lib()
    synth = pkg.synth
    arch = synth.arch_for_size(size)
    sd = synth.make_state_dict(arch, 10)
    blob = pkg.weights.pack_blob(arch, sd)
    org, pred = synth.make_patches(size, n, 1234)
    poc, qp = synth.make_scalars(n, 1234)
    m = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob})
    split, logits = m.predict_batch(org, pred, poc, qp)
    files = sorted(os.listdir(dump))
    T = lambda k: torch.from_numpy(np.asarray(sd[k]))

    def fold(w, p):
        s = T(p + ".weight") / torch.sqrt(T(p + ".running_var") + 1e-5)
        return w * s.view(-1, 1, 1, 1), T(p + ".bias") - T(p + ".running_mean") * s

    def load(i, c, h):
        a = np.fromfile(os.path.join(dump, files[i]), dtype=np.float16).reshape(n, h, h, c)
        return torch.from_numpy(a.astype(np.float32)).permute(0, 3, 1, 2)

    o = torch.from_numpy(org.astype(np.float32)); p = torch.from_numpy(pred.astype(np.float32))
    x = torch.stack([o, (o - p).abs()], 1) * np.float32(1.0 / 1023)
    # the stem activation lives only in LDS (fp16); emulate that rounding
    cur = F.conv2d(x, T("conv1.weight"), padding=1).half().float()  # (exact-integer inputs, 1/1023 applied in fp32)
    idx = 0
    planes = synth.STAGE_PLANES[arch]
    h = size
    for li, c in enumerate(planes):
        last = li == len(planes) - 1
        pfx = f"layer{li}.0"
        ho = max(h // 2, 1)
        w1, b1 = fold(T(pfx + ".conv1.weight"), pfx + ".bn1")
        t_ref = F.relu(F.conv2d(cur, w1, b1, stride=2, padding=1)); t = load(idx, c, ho)
        print(f"{files[idx]:44s} max|d| {float((t - t_ref).abs().max()):.3e}"); idx += 1
        ws, bs = fold(T(pfx + ".shortcut.0.weight"), pfx + ".shortcut.1")
        s_ref = F.conv2d(cur, ws, bs, stride=2); sc = load(idx, c, ho)
        print(f"{files[idx]:44s} max|d| {float((sc - s_ref).abs().max()):.3e}"); idx += 1
        w2, b2 = fold(T(pfx + ".conv2.weight"), pfx + ".bn2")
        b0_ref = F.relu(F.conv2d(t, w2, b2, padding=1) + sc); b0 = load(idx, c, ho)
        print(f"{files[idx]:44s} max|d| {float((b0 - b0_ref).abs().max()):.3e}"); idx += 1
        pfx = f"layer{li}.1"
        w1, b1 = fold(T(pfx + ".conv1.weight"), pfx + ".bn1")
        t_ref = F.relu(F.conv2d(b0, w1, b1, padding=1)); t = load(idx, c, ho)
        print(f"{files[idx]:44s} max|d| {float((t - t_ref).abs().max()):.3e}"); idx += 1
        if not last:
            w2, b2 = fold(T(pfx + ".conv2.weight"), pfx + ".bn2")
            o_ref = F.relu(F.conv2d(t, w2, b2, padding=1) + b0); out = load(idx, c, ho)
            print(f"{files[idx]:44s} max|d| {float((out - o_ref).abs().max()):.3e}  (|ref|max {float(o_ref.abs().max()):.2f})"); idx += 1
            cur = out
        h = ho
    from oracle import Oracle
    ref_logits, ref_split = Oracle(blob).forward(org, pred, poc, qp)
    print("logits max|d| vs oracle:", float(np.abs(logits - ref_logits).max()), "split", split, ref_split)


if __name__ == "__main__":
    main()
