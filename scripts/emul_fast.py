#!/usr/bin/env python3
"""CPU emulation of the 128x128 "fast" arithmetic (fp16 operands, fp32 accumulate) -- numerics study aid.

Mirrors, with torch CPU ops on fp32 tensors that hold fp16-representable values, every rounding point of the HIP
pipeline (mlt_kernels.hip / mlt_model.cpp): BN folded in double, tap-diffused fp16 weights, composed first layer
with its border-correction slots, fp16 activations between convs, fp32 shortcut inside the fused layer0.0, fp32 GAP
and heads.  Lets candidate arithmetic changes (rounding modes, hi/lo splits per layer) be priced against the
reference fixtures WITHOUT a GPU.  Not part of the product or the tests.

    python scripts/emul_fast.py [strategy ...]
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mltcnn_pkg  # noqa: E402
from helpers import load_golden, materialise  # noqa: E402

pkg = mltcnn_pkg.load()
torch.set_num_threads(8)


_DITHER = {}


def rn16(x):
    return x.to(torch.float16).to(torch.float32)


def diffuse_round(w):
    """w: float64 [..., taps]; tap-diffused fp16 rounding along the last axis (mlt_model.cpp pack_conv)."""
    out = np.empty_like(w, dtype=np.float32)
    err = np.zeros(w.shape[:-1], np.float64)
    for t in range(w.shape[-1]):
        q = (w[..., t] - err).astype(np.float32).astype(np.float16).astype(np.float32)
        err += q.astype(np.float64) - w[..., t]
        out[..., t] = q
    return out


def split_hi_lo(w):
    hi = w.astype(np.float32).astype(np.float16)
    lo = (w - hi.astype(np.float64)).astype(np.float32).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


class Emul:
    def __init__(self, blob, opts):
        self.opts = opts
        arch, sd = pkg.weights.unpack_blob(blob)
        assert arch == 0
        self.sd = {k: np.asarray(v, np.float64) for k, v in sd.items()}
        self.planes = (32, 64, 128, 256)

    def fold(self, wname, bn):
        sd = self.sd
        s = sd[bn + ".weight"] / np.sqrt(sd[bn + ".running_var"] + 1e-5)
        w = sd[wname] * s.reshape(-1, 1, 1, 1)
        b = sd[bn + ".bias"] - sd[bn + ".running_mean"] * s
        return w, b.astype(np.float32)

    def qw(self, w, name):
        """fp16 weight tensor(s) for conv `name` (w: float64 [co,ci,kh,kw]) -> list of fp32 tensors to be summed."""
        co, ci, kh, kw = w.shape
        if name in self.opts.get("w_hilo", ()):
            hi, lo = split_hi_lo(w)
            return [torch.from_numpy(hi), torch.from_numpy(lo)]
        if self.opts.get("w_rn"):
            return [torch.from_numpy(w.astype(np.float32).astype(np.float16).astype(np.float32))]
        q = diffuse_round(w.reshape(co, ci, kh * kw)).reshape(co, ci, kh, kw)
        return [torch.from_numpy(q)]

    def act(self, x, name):
        """Round an activation tensor for storage (fp16).  Strategies:
        rn (default) | hilo (keep x - rn16(x) as a second fp16 plane => returns hi+lo as fp32) | dither"""
        mode = self.opts.get("act", {}).get(name, self.opts.get("act_default", "rn"))
        rec = self.opts.get("record")
        if rec is not None:  # tools/attribute_error.py: the unrounded tensor and what the site stores
            y = self._act(x, name, mode)
            rec[name] = (x.clone(), y.clone())
            return y
        return self._act(x, name, mode)

    def _act(self, x, name, mode):
        if mode == "rn":
            return rn16(x)
        if mode == "pdither":
            # position-keyed sub-ulp dither: a deterministic offset in [-1/2, 1/2) ulp, a function of (channel, y, x) only --
            # the same for every CU and every kernel form -- added before the round-to-nearest.  Spatially constant activations then round up at
            # some pixels and down at others, in proportion to their position inside the fp16 interval: the pooled mean is unbiased.
            amp = float(self.opts.get("dither_amp", 1.0))
            n, c, h, w = x.shape
            key = (self.opts.get("dither_key", 0), c, h, w)
            if key not in _DITHER:
                ci = torch.arange(c, dtype=torch.int64).view(c, 1, 1)
                yi = torch.arange(h, dtype=torch.int64).view(1, h, 1)
                xi = torch.arange(w, dtype=torch.int64).view(1, 1, w)
                hsh = (ci * 0x9E3779B1 + yi * 0x85EBCA77 + xi * 0xC2B2AE3D + key[0] * 0x27D4EB2F) & 0xFFFFFFFF
                hsh = ((hsh ^ (hsh >> 15)) * 0x2C1B3C6D) & 0xFFFFFFFF
                hsh = ((hsh ^ (hsh >> 12)) * 0x297A2D39) & 0xFFFFFFFF
                hsh = hsh ^ (hsh >> 15)
                _DITHER[key] = ((hsh & 0xFFFF).to(torch.float32) / 65536.0 - 0.5).view(1, c, h, w)
            u = _DITHER[key] * amp
            # ulp of fp16 at |x| (normals: 2^(e - 10); below 2^-14: 2^-24)
            e = torch.floor(torch.log2(x.abs().clamp(min=2.0 ** -14)))
            ulp = torch.pow(2.0, e - 10.0)
            return rn16(x + u * ulp)
        if mode == "hilo":
            hi = rn16(x)
            return hi + rn16(x - hi)
        if mode == "f32":
            return x
        if mode == "dither":
            # stochastic rounding with a deterministic per-element hash: add uniform [-0.5,0.5) ulp16 then RN... emulate by
            # adding u * ulp before truncation toward -inf on the fp16 grid
            g = torch.Generator().manual_seed(hash(name) & 0xFFFF)
            bits = x.view(torch.int32)
            r = torch.randint(0, 1 << 13, x.shape, generator=g, dtype=torch.int32)
            y = ((bits + r) & ~0x1FFF).view(torch.float32)  # magnitude-wise stochastic rounding on the 10-bit mantissa grid (normals)
            return y.to(torch.float16).to(torch.float32)
        raise ValueError(mode)

    def conv(self, x, ws, **kw):
        y = None
        for w in ws:
            t = F.conv2d(x, w, **kw)
            y = t if y is None else y + t
        return y

    def forward(self, org, pred, poc, qp):
        o = torch.from_numpy(org.view(np.uint16).astype(np.float32))
        p = torch.from_numpy(pred.view(np.uint16).astype(np.float32))
        x = torch.stack([o.clamp(max=1023.0), (o - p).abs().clamp(max=1023.0)], dim=1)  # exact integers
        n = x.shape[0]
        sd = self.sd
        # ---- composed first layer (pack_stem5) ----
        Ws = sd["conv1.weight"]  # [32,2,3,3]
        W1, b1 = self.fold("layer0.0.conv1.weight", "layer0.0.bn1")
        Wsc, bsc = self.fold("layer0.0.shortcut.0.weight", "layer0.0.shortcut.1")
        mul = float(np.float32(1.0 / 1023)) * 4096.0
        w5 = np.zeros((32, 2, 5, 5))
        top = np.zeros((32, 2, 5)); left = np.zeros((32, 2, 5)); corner = np.zeros((32, 2)); sc3 = np.zeros((32, 2, 3, 3))
        for ay in range(3):
            for ax in range(3):
                for by in range(3):
                    for bx in range(3):
                        w5[:, :, ay + by, ax + bx] += np.einsum("om,mc->oc", W1[:, :, ay, ax], Ws[:, :, by, bx])
        for ax in range(3):
            for bx in range(3):
                top[:, :, ax + bx] -= np.einsum("om,mc->oc", W1[:, :, 0, ax], Ws[:, :, 2, bx])
        for ay in range(3):
            for by in range(3):
                left[:, :, ay + by] -= np.einsum("om,mc->oc", W1[:, :, ay, 0], Ws[:, :, by, 2])
        corner += np.einsum("om,mc->oc", W1[:, :, 0, 0], Ws[:, :, 2, 2])
        for by in range(3):
            for bx in range(3):
                sc3[:, :, by, bx] += np.einsum("om,mc->oc", Wsc[:, :, 0, 0], Ws[:, :, by, bx])
        stem_hilo = "stem" in self.opts.get("w_hilo", ())

        def q_plain(a):
            if stem_hilo:
                hi, lo = split_hi_lo(a * mul)
                return torch.from_numpy(hi + lo)  # two MFMAs with the same exact-integer B operand
            return torch.from_numpy((a * mul).astype(np.float32).astype(np.float16).astype(np.float32))

        def q_diff(a):
            if stem_hilo:
                return q_plain(a)
            sh = a.shape
            return torch.from_numpy(diffuse_round((a * mul).reshape(sh[0], sh[1], -1)).reshape(sh))

        w5q, sc3q = q_diff(w5), q_diff(sc3)
        topq, leftq, cornq = q_plain(top), q_plain(left), q_plain(corner)
        acc = F.conv2d(x, w5q, stride=2, padding=2)
        acc[:, :, 0:1, :] += F.conv2d(x[:, :, 0:1, :], topq.reshape(32, 2, 1, 5), stride=(1, 2), padding=(0, 2))
        acc[:, :, :, 0:1] += F.conv2d(x[:, :, :, 0:1], leftq.reshape(32, 2, 5, 1), stride=(2, 1), padding=(2, 0))
        acc[:, :, 0, 0] += torch.einsum("oc,nc->no", cornq, x[:, :, 0, 0])
        t = F.relu(acc * (1.0 / 4096.0) + torch.from_numpy(b1).view(1, -1, 1, 1))
        t = self.act(t, "l0.0.t")
        sc = F.conv2d(x, sc3q, stride=2, padding=1) * (1.0 / 4096.0)  # fp32, never rounded (fused kernel)
        W2, b2 = self.fold("layer0.0.conv2.weight", "layer0.0.bn2")
        u = self.conv(t, self.qw(W2, "l0.0.c2"), padding=1) + (sc + torch.from_numpy(b2 + bsc).view(1, -1, 1, 1))
        cur = self.act(F.relu(u), "l0.0.out")
        # layer0.1
        W1, b1 = self.fold("layer0.1.conv1.weight", "layer0.1.bn1")
        t = self.act(F.relu(self.conv(cur, self.qw(W1, "l0.1.c1"), padding=1) + torch.from_numpy(b1).view(1, -1, 1, 1)), "l0.1.t")
        W2, b2 = self.fold("layer0.1.conv2.weight", "layer0.1.bn2")
        cur = self.act(F.relu(self.conv(t, self.qw(W2, "l0.1.c2"), padding=1) + torch.from_numpy(b2).view(1, -1, 1, 1) + cur), "l0.1.out")
        feats = []
        for s in (1, 2, 3):
            pf = f"layer{s}.0"
            W1, b1 = self.fold(pf + ".conv1.weight", pf + ".bn1")
            Wsc, bsc = self.fold(pf + ".shortcut.0.weight", pf + ".shortcut.1")
            t = self.act(F.relu(self.conv(cur, self.qw(W1, f"l{s}.0.c1"), stride=2, padding=1) + torch.from_numpy(b1).view(1, -1, 1, 1)), f"l{s}.0.t")
            scq = [torch.from_numpy(w_) for w_ in (split_hi_lo(Wsc) if f"l{s}.0.sc" in self.opts.get("w_hilo", ()) else [Wsc.astype(np.float32).astype(np.float16).astype(np.float32)])]
            sc = self.act(self.conv(cur, scq, stride=2) + torch.from_numpy(bsc).view(1, -1, 1, 1), f"l{s}.0.sc")
            W2, b2 = self.fold(pf + ".conv2.weight", pf + ".bn2")
            b0 = self.act(F.relu(self.conv(t, self.qw(W2, f"l{s}.0.c2"), padding=1) + torch.from_numpy(b2).view(1, -1, 1, 1) + sc), f"l{s}.0.out")
            pf = f"layer{s}.1"
            W1, b1 = self.fold(pf + ".conv1.weight", pf + ".bn1")
            t = self.act(F.relu(self.conv(b0, self.qw(W1, f"l{s}.1.c1"), padding=1) + torch.from_numpy(b1).view(1, -1, 1, 1)), f"l{s}.1.t")
            W2, b2 = self.fold(pf + ".conv2.weight", pf + ".bn2")
            outf = F.relu(self.conv(t, self.qw(W2, f"l{s}.1.c2"), padding=1) + torch.from_numpy(b2).view(1, -1, 1, 1) + b0)
            feats.append(outf.mean(dim=(2, 3)))
            cur = self.act(outf, f"l{s}.1.out")
        extra = torch.tensor(np.stack([poc, qp], 1).astype(np.float32))
        outs, mags = [], []
        for s in (1, 2, 3):
            w = torch.from_numpy(sd[f"branch{s}.weight"].astype(np.float32)); b = torch.from_numpy(sd[f"branch{s}.bias"].astype(np.float32))
            outs.append(F.linear(torch.cat([feats[s - 1], extra], 1), w, b))
            mags.append(F.linear(feats[s - 1].abs(), w[:, :-2].abs()))   # sum_k |w_ck gap_k|: the magnitude the fp16 pipeline's relative error acts on
        self.last_mag = torch.cat(mags, 1).numpy()
        return torch.cat(outs, 1).numpy()


ACT_SITES = ["l0.0.t", "l0.0.out", "l0.1.t", "l0.1.out"] + [f"l{s}.{b}.{k}" for s in (1, 2, 3) for b, ks in ((0, ("t", "sc", "out")), (1, ("t", "out"))) for k in ks]
CONVS = ["stem", "l0.0.c2", "l0.1.c1", "l0.1.c2"] + [f"l{s}.{b}.{c}" for s in (1, 2, 3) for b, cs in ((0, ("c1", "sc", "c2")), (1, ("c1", "c2"))) for c in cs]

STRATEGIES = {
    "base": {},
    "all_f32_act": {"act_default": "f32"},
    "all_hilo_w": {"w_hilo": CONVS},
    "both": {"act_default": "f32", "w_hilo": CONVS},
    "stem_hilo": {"w_hilo": ["stem"]},
    "dither": {"act_default": "dither"},
    "dither+stem": {"act_default": "dither", "w_hilo": ["stem"]},
    "pdither": {"act_default": "pdither"},
    "pdither_l01": {"act": {k: "pdither" for k in ("l0.0.t", "l0.0.out", "l0.1.t", "l0.1.out", "l1.0.t", "l1.0.sc", "l1.0.out", "l1.1.t", "l1.1.out")}},
    "pdither_half": {"act_default": "pdither", "dither_amp": 0.5},
}


def run(names, cases=None):
    golden = load_golden(128)
    rows = {}
    for case in golden["cases"]:
        if cases and case["name"] not in cases:
            continue
        blob, org, pred, poc, qp, exp, _ = materialise(pkg, golden, case)
        for nm in names:
            got = Emul(blob, STRATEGIES[nm]).forward(org, pred, poc, qp)
            rows.setdefault(case["name"], {})[nm] = float(np.abs(got - exp).max())
    print(f"{'case':24s}" + "".join(f"{n:>14s}" for n in names))
    for c, r in rows.items():
        print(f"{c:24s}" + "".join(f"{r[n]:14.2e}" for n in names))
    return rows


if __name__ == "__main__":
    names = sys.argv[1:] or ["base", "all_f32_act", "all_hilo_w", "both"]
    run(names)
