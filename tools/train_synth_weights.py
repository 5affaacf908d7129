#!/usr/bin/env python3
"""A NON-i.i.d. weight family for the arithmetic probes: the 128 x 128 graph (this repository's own restatement of
mlt_ctu_or_pq_arch.py:239-299 -- torch.nn.functional calls, no reference module or trainer is imported) trained for a few hundred Adam
steps on SYNTHETIC labels, so that the weights carry what training leaves behind and Kaiming-random sets do not: correlated filters,
BatchNorm statistics that come from data, heads that use the features.  No trained checkpoint of the authors is distributed
(.MISSING_LARGE_BLOBS:7); this is the closest stand-in the container allows.

Task (learnable from the two input planes): residual energy per quadrant decides the label of the lvl3 head --
  0 no split  : mean |org - pred| below a QP-dependent threshold        1 QT   : energy spread over both axes
  2 BT_H      : top / bottom halves differ most                          3 BT_V : left / right halves differ most
(class ids as mlt_ctu_or_pq_dataset.py:17); lvl1 = split or not, lvl2 = none / QT / BT.  Content: synth.natural_patches (1/f scenes, motion-
shifted prediction) with the residual re-weighted per quadrant.

  python tools/train_synth_weights.py out.mltw [--steps 300] [--batch 16] [--seed 1] [--threads N]

Writes an MLTW blob (fastintercu-vvc_amd/weights.py).  CPU only; ~3 minutes on 8 cores.  The blob is NOT bit-reproducible across machines
(thread count changes fp32 summation order) -- it is a probe input, not a fixture: the oracle is always evaluated on the same blob."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_batch(S, size, n, seed, first):
    org, pred = S.natural_patches(size, n, seed, first)
    rng = np.random.default_rng((seed << 20) + first)
    o = org.astype(np.int32)
    r = pred.astype(np.int32) - o
    gains = rng.choice(np.array([0.0, 0.25, 1.0, 3.0]), size=(n, 2, 2))
    hs = size // 2
    g = np.repeat(np.repeat(gains, hs, axis=1), hs, axis=2)
    p = np.clip(o + np.rint(r * g).astype(np.int32), 0, 1023)
    a = np.abs(o - p).astype(np.float32)
    q = a.reshape(n, 2, hs, 2, hs).mean(axis=(2, 4))                    # quadrant energies [n][row][col]
    tot = q.mean(axis=(1, 2))
    qp = rng.integers(22, 38, size=n).astype(np.int32)
    poc = rng.integers(0, 65, size=n).astype(np.int32)
    thr = 0.6 + 0.12 * (qp - 22)                                         # coarser quantisation tolerates more residual
    dv = np.abs(q[:, 0, :].mean(axis=1) - q[:, 1, :].mean(axis=1))       # top vs bottom
    dh = np.abs(q[:, :, 0].mean(axis=1) - q[:, :, 1].mean(axis=1))       # left vs right
    lab3 = np.where(tot < thr, 0, np.where((dv > 1.5 * dh) & (dv > 0.25 * tot), 2, np.where((dh > 1.5 * dv) & (dh > 0.25 * tot), 3, 1)))
    lab1 = (lab3 != 0).astype(np.int64)
    lab2 = np.where(lab3 == 0, 0, np.where(lab3 == 1, 1, 2)).astype(np.int64)
    return o.astype(np.int16), p.astype(np.int16), poc, qp, (lab1, lab2, lab3.astype(np.int64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    import torch
    import torch.nn.functional as F
    import mltcnn_pkg
    pkg = mltcnn_pkg.load()
    S = pkg.synth
    if a.threads:
        torch.set_num_threads(a.threads)
    torch.manual_seed(a.seed)
    arch, size = S.ARCH_CTU, 128
    init = S.make_state_dict(arch, 1000 + a.seed)
    P, B = {}, {}                                                         # parameters / BatchNorm buffers
    for k, v in init.items():
        if k.endswith("num_batches_tracked") or k.startswith("bn1."):
            continue
        if k.endswith("running_mean"):
            B[k] = torch.zeros(v.shape)
        elif k.endswith("running_var"):
            B[k] = torch.ones(v.shape)
        elif (".bn" in k or "shortcut.1" in k):
            P[k] = torch.nn.Parameter(torch.ones(v.shape) if k.endswith(".weight") else torch.zeros(v.shape))   # a fresh network's BatchNorm
        else:
            P[k] = torch.nn.Parameter(torch.from_numpy(v.copy()))

    def bn(x, p, train):
        return F.batch_norm(x, B[p + ".running_mean"], B[p + ".running_var"], P[p + ".weight"], P[p + ".bias"], train, 0.1, 1e-5)

    def block(x, p, stride, train):
        t = F.relu(bn(F.conv2d(x, P[p + ".conv1.weight"], stride=stride, padding=1), p + ".bn1", train))
        u = bn(F.conv2d(t, P[p + ".conv2.weight"], padding=1), p + ".bn2", train)
        if stride != 1:
            x = bn(F.conv2d(x, P[p + ".shortcut.0.weight"], stride=stride), p + ".shortcut.1", train)
        return F.relu(u + x)

    def forward(org, pred, poc, qp, train):
        o = torch.from_numpy(org.astype(np.float32))
        p = torch.from_numpy(pred.astype(np.float32))
        c = np.float32(1.0 / 1023)
        x = torch.stack([(o * c).clamp(0, 1), ((o - p).abs() * c).clamp(0, 1)], dim=1)
        extra = torch.stack([torch.from_numpy(poc.astype(np.float32)), torch.from_numpy(qp.astype(np.float32))], dim=1)
        cur = F.conv2d(x, P["conv1.weight"], padding=1)
        outs = []
        for s in range(4):
            cur = block(cur, f"layer{s}.0", 2, train)
            cur = block(cur, f"layer{s}.1", 1, train)
            if s >= 1:
                outs.append(F.linear(torch.cat([cur.mean(dim=(2, 3)), extra], dim=1), P[f"branch{s}.weight"], P[f"branch{s}.bias"]))
        return outs

    opt = torch.optim.Adam(list(P.values()), lr=2e-3)
    t0 = time.time()
    for step in range(a.steps):
        org, pred, poc, qp, labs = make_batch(S, size, a.batch, a.seed, step * a.batch)
        outs = forward(org, pred, poc, qp, True)
        loss = sum(F.cross_entropy(o, torch.from_numpy(l)) for o, l in zip(outs, labs))
        opt.zero_grad()
        loss.backward()
        opt.step()
        if step % 25 == 0 or step == a.steps - 1:
            acc = [float((o.argmax(1) == torch.from_numpy(l)).float().mean()) for o, l in zip(outs, labs)]
            print(f"step {step:4d}  loss {loss.item():.3f}  batch accuracy lvl1/2/3 {acc[0]:.2f} {acc[1]:.2f} {acc[2]:.2f}  ({time.time() - t0:.0f} s)", flush=True)
    with torch.no_grad():  # held-out accuracy in eval mode (running statistics), the mode the encoder runs
        org, pred, poc, qp, labs = make_batch(S, size, 128, a.seed + 77, 0)
        outs = forward(org, pred, poc, qp, False)
        acc = [float((o.argmax(1) == torch.from_numpy(l)).float().mean()) for o, l in zip(outs, labs)]
        print(f"held-out (eval mode, 128 CUs): accuracy lvl1/2/3 {acc[0]:.2f} {acc[1]:.2f} {acc[2]:.2f}; label histogram lvl3 {np.bincount(labs[2], minlength=4).tolist()}")
    sd = {k: v.detach().numpy() for k, v in P.items()}
    sd.update({k: v.numpy() for k, v in B.items()})
    blob = pkg.weights.pack_blob(arch, sd)
    open(a.out, "wb").write(blob)
    print(f"wrote {a.out}: {len(blob)} bytes")


if __name__ == "__main__":
    main()
