"""Second CPU restatement of the path, on PyTorch's CPU operators (oneDNN convolutions) -- TEST INFRASTRUCTURE ONLY.

Same role and same import rules as oracle.py / mlt_oracle.c: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may use it.  It is an independent cross-check of the C oracle (different operators, different summation
order, NCHW) and the restatement closest to what the reference's CPU path would execute (LibTorch CPU backend,
EncCu.cpp:869-921).  It IS bench.py's cpu_baseline (oracle/cpu_baseline.py times it in child processes: one CU at a time with a
thread sweep, batch 64, and the batch sharded over one single-threaded process per physical core -- the best row, 977 CU/s on the
GPU box's 128 cores in round 4, is `cpu_baseline.value`; the OpenMP-over-CUs C oracle, 220 CU/s, is one more row of the same table).
Restates, with torch.nn.functional only (no reference module is imported):
  preprocessing  EncCu.cpp:816,827 (uint16 cast), :833 (absdiff), :836,838 (* (float)(1/1023)), :848-867 (clip)
  network        mlt_ctu_or_pq_arch.py:32-57 (BasicBlock), :273-299 (forward: stem without bn/relu, GAP, heads with
                 [features, poc, qp]), mlt_cu_or_pq_arch.py:96-128 (five stages, four heads)
  decision       EncCu.cpp:913-921 (head [2] for 128, [0] otherwise; torch.argmax = first maximum)
"""
from __future__ import annotations

import struct

import numpy as np
import torch
import torch.nn.functional as F

_HEADER = struct.Struct("<4sIII")
_ENTRY = struct.Struct("<64sI4IQQ")


def _state_dict(blob: bytes):
    magic, version, arch, n = _HEADER.unpack_from(blob, 0)
    if magic != b"MLTW" or version != 1:
        raise ValueError("not an MLTW v1 blob")
    pos = _HEADER.size
    data = np.frombuffer(blob, dtype=np.float32, offset=pos + n * _ENTRY.size)
    sd = {}
    for _ in range(n):
        name, ndim, d0, d1, d2, d3, off, numel = _ENTRY.unpack_from(blob, pos)
        pos += _ENTRY.size
        sd[name.rstrip(b"\0").decode()] = torch.from_numpy(data[off:off + numel].reshape((d0, d1, d2, d3)[:ndim]).copy())
    return arch, sd


class TorchPort:
    def __init__(self, blob: bytes):
        self.arch, self.sd = _state_dict(blob)
        self.n_stages = 4 if self.arch == 0 else 5
        self.head_classes = [self.sd[f"branch{i}.bias"].numel() for i in range(1, self.n_stages)]
        self.n_logits = sum(self.head_classes)

    def _bn(self, x, p):
        sd = self.sd
        return F.batch_norm(x, sd[p + ".running_mean"], sd[p + ".running_var"], sd[p + ".weight"], sd[p + ".bias"], False, 0.0, 1e-5)

    def _block(self, x, p, stride):
        sd = self.sd
        t = F.relu(self._bn(F.conv2d(x, sd[p + ".conv1.weight"], stride=stride, padding=1), p + ".bn1"))
        u = self._bn(F.conv2d(t, sd[p + ".conv2.weight"], padding=1), p + ".bn2")
        if stride != 1:  # projection shortcut on the first block of every stage (arch:44-50)
            x = self._bn(F.conv2d(x, sd[p + ".shortcut.0.weight"], stride=stride), p + ".shortcut.1")
        return F.relu(u + x)

    @torch.no_grad()
    def forward(self, org, pred, poc, qp, head_index: int = -1, threads: int = 0, chunk: int = 64):
        if threads:
            torch.set_num_threads(threads)
        o = torch.from_numpy(np.ascontiguousarray(org).view(np.uint16).astype(np.float32))
        p = torch.from_numpy(np.ascontiguousarray(pred).view(np.uint16).astype(np.float32))
        c = torch.tensor(np.float32(1.0 / 1023))
        n, S = o.shape[0], o.shape[1]
        dec = head_index if head_index >= 0 else (2 if S == 128 else 0)
        logits = torch.empty((n, self.n_logits), dtype=torch.float32)
        for i0 in range(0, n, chunk):
            oc, pc = o[i0:i0 + chunk], p[i0:i0 + chunk]
            x = torch.stack([(oc * c).clamp(0.0, 1.0), ((oc - pc).abs() * c).clamp(0.0, 1.0)], dim=1)
            extra = torch.stack([torch.from_numpy(np.asarray(poc[i0:i0 + chunk], np.float32)),
                                 torch.from_numpy(np.asarray(qp[i0:i0 + chunk], np.float32))], dim=1)
            cur = F.conv2d(x, self.sd["conv1.weight"], padding=1)  # stem: no BN, no ReLU (arch:277-278)
            outs = []
            for s in range(self.n_stages):
                cur = self._block(cur, f"layer{s}.0", 2)
                cur = self._block(cur, f"layer{s}.1", 1)
                if s >= 1:
                    feat = torch.cat([cur.mean(dim=(2, 3)), extra], dim=1)
                    outs.append(F.linear(feat, self.sd[f"branch{s}.weight"], self.sd[f"branch{s}.bias"]))
            logits[i0:i0 + chunk] = torch.cat(outs, dim=1)
        lo = sum(self.head_classes[:dec])
        split = logits[:, lo:lo + self.head_classes[dec]].argmax(dim=1).to(torch.int32)
        return logits.numpy(), split.numpy()
