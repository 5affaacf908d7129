"""ctypes wrapper of oracle/libmlt_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
The product path (fastintercu-vvc_amd/) never does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libmlt_oracle.so")
    src = os.path.join(_HERE, "mlt_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libmlt_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        lib = C.CDLL(build())
        lib.mlto_load.restype = C.c_void_p
        lib.mlto_load.argtypes = [C.c_void_p, C.c_size_t]
        lib.mlto_free.argtypes = [C.c_void_p]
        lib.mlto_num_logits.argtypes = [C.c_void_p]
        lib.mlto_num_heads.argtypes = [C.c_void_p]
        lib.mlto_head_classes.argtypes = [C.c_void_p, C.c_int]
        lib.mlto_forward.argtypes = [C.c_void_p, C.c_int, C.c_int,
                                     C.c_void_p, C.c_long, C.c_long, C.c_void_p, C.c_long, C.c_long,
                                     C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        _LIB = lib
    return _LIB


class Oracle:
    """fp32 CPU restatement of the reference network for one weight blob."""

    def __init__(self, blob: bytes):
        lib = _lib()
        self._buf = C.create_string_buffer(blob, len(blob))
        self._h = lib.mlto_load(self._buf, len(blob))
        if not self._h:
            raise ValueError("oracle: bad weight blob")
        self.n_logits = lib.mlto_num_logits(self._h)
        self.head_classes = [lib.mlto_head_classes(self._h, i) for i in range(lib.mlto_num_heads(self._h))]

    def __del__(self):
        if getattr(self, "_h", None):
            _lib().mlto_free(self._h)
            self._h = None

    def forward(self, org, pred, poc, qp, head_index: int = -1, threads: int = 0):
        """org/pred: int16 [n,S,S] (may be non-contiguous views with positive strides).
        Returns (logits float32 [n, n_logits], split int32 [n])."""
        org = np.asarray(org)
        pred = np.asarray(pred)
        assert org.dtype == np.int16 and pred.dtype == np.int16 and org.ndim == 3 and org.shape == pred.shape
        n, S, _ = org.shape
        for a in (org, pred):
            assert a.strides[2] == 2 and a.strides[1] % 2 == 0 and (n <= 1 or a.strides[0] % 2 == 0)
        poc = np.ascontiguousarray(poc, dtype=np.int32)
        qp = np.ascontiguousarray(qp, dtype=np.int32)
        logits = np.empty((n, self.n_logits), dtype=np.float32)
        split = np.empty((n,), dtype=np.int32)
        rc = _lib().mlto_forward(self._h, n, S,
                                 org.ctypes.data, org.strides[1] // 2, org.strides[0] // 2,
                                 pred.ctypes.data, pred.strides[1] // 2, pred.strides[0] // 2,
                                 poc.ctypes.data, qp.ctypes.data, head_index,
                                 logits.ctypes.data, split.ctypes.data, threads)
        if rc != 0:
            raise RuntimeError(f"mlto_forward failed rc={rc}")
        return logits, split
