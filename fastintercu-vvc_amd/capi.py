"""ctypes binding of libmltcnn_hip.so (include/mltcnn.h) + the host-side mirror of the reference call site.

`MltCnn.predict(org, pred, poc, qp)` takes exactly what EncCu.cpp:806-830 reads at the call site
(two Pel planes with their strides, the slice POC and the CU QP) and returns what :921 produces
(`predictedSplitMode`).  There is NO fallback: a missing library raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build as _build

MLT_OK = 0
ERR_NAMES = {1: "MLT_ERR_ARG", 2: "MLT_ERR_NO_DEVICE", 3: "MLT_ERR_WEIGHTS", 4: "MLT_ERR_SIZE_DISABLED",
             5: "MLT_ERR_HIP", 6: "MLT_ERR_NOMEM"}
SIZE_BITS = {128: 1, 64: 2, 32: 4, 16: 8}
FLAG_EXACT_128 = 0x1   # 128x128 in exact (fp16 hi+lo, 3-pass) arithmetic instead of fast
FLAG_FAST_SMALL = 0x2  # 64/32/16 in fast arithmetic instead of exact
FLAG_DECISION_GUARD = 0x4  # ABI <= 3 opt-in; since ABI 4 the decision guard is the default (accepted, no effect)
FLAG_NO_FLAT_GUARD = 0x8   # fast arithmetic without the flat-content guard (measurement only)
FLAG_NO_CALIBRATION = 0x10  # keep the fast arithmetic whatever the weight set (measurement only)
FLAG_EXACT_LITE = 0x40  # round 5 (measurement): exact-configured sizes run the exact-lite arithmetic (FP8 cross terms)
FLAG_NO_MAGNITUDE_GUARD = 0x80  # round 6 (measurement): never admit a tier behind the magnitude guard
FLAG_NO_DECISION_GUARD = 0x20  # ABI 4: no exact re-evaluation of CUs with a near-tie on the decision head (measurement only)
EXPORTS = ["mlt_abi_version", "mlt_build_signature", "mlt_init", "mlt_num_devices", "mlt_device_ctx", "mlt_load_weights", "mlt_calibrate", "mlt_arithmetic", "mlt_predict", "mlt_predict_batch",
           "mlt_predict_batch_device", "mlt_submit", "mlt_flush", "mlt_wait", "mlt_synchronize", "mlt_set_stream", "mlt_alloc_pinned", "mlt_free_pinned",
           "mlt_num_logits", "mlt_profile_enable", "mlt_profile_read", "mlt_last_error", "mlt_shutdown"]


class MltConfig(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("weights_dir", C.c_char_p),
                ("size_mask", C.c_uint32), ("head_index", C.c_int32 * 4), ("max_batch", C.c_int32),
                ("flags", C.c_uint32), ("guard_margin", C.c_float), ("tolerance", C.c_float), ("reserved", C.c_uint32),
                ("n_devices", C.c_int32), ("devices", C.c_int32 * 8)]


class MltArithInfo(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("exact", C.c_int32), ("calibrated", C.c_int32), ("calib_rms", C.c_float), ("calib_max", C.c_float),
                ("flat_guard", C.c_int32), ("decision_guard", C.c_int32), ("guard_reruns", C.c_uint64),
                ("w2_stages", C.c_int32), ("guard_margin", C.c_float), ("x_stages", C.c_int32), ("w2_units", C.c_int32), ("x_units", C.c_int32), ("rounding", C.c_int32),
                ("calib_cus", C.c_int32), ("calib_caller_cus", C.c_int32), ("mag_guard_thr", C.c_float), ("mag_guard_flagged", C.c_float),
                ("mag_guard_kind", C.c_int32)]


class MltKernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 48), ("launches", C.c_uint32), ("total_ms", C.c_float),
                ("flops", C.c_double), ("bytes", C.c_double)]


class MltError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


_LIB = None


def lib_path() -> str:
    return os.environ.get("MLT_LIB_PATH") or _build.LIB  # MLT_LIB_PATH: tuning variants only (scripts/sweep_cfg.py)


def load_library():
    """Loads the in-tree HIP library; raises if it has not been built (no silent fallback).

    torch is imported FIRST on purpose: in this image the working HIP runtime is the libamdhip64.so.7 bundled
    with PyTorch-ROCm; loading ours first would bind the system copy of the same soname and HIP init fails."""
    import torch  # noqa: F401
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(f"{path} is missing: run __graft_entry__.build() (hipcc --offload-arch=gfx950) first")
    lib = C.CDLL(path)
    vp, i32 = C.c_void_p, C.c_int
    lib.mlt_abi_version.restype = i32
    lib.mlt_build_signature.restype = C.c_char_p
    if "MLT_LIB_PATH" not in os.environ and os.path.isdir(_build.CSRC):
        # the binary travels to the GPU box with the snapshot: one built from other sources than the tree's must not pass for them
        have, want = lib.mlt_build_signature().decode(), _build.source_signature()
        if have != want:
            raise RuntimeError(f"{path} was built from sources {have}, the tree is {want}: rebuild (python __graft_entry__.py)")
    lib.mlt_init.argtypes = [C.POINTER(MltConfig), C.POINTER(vp)]
    lib.mlt_num_devices.argtypes = [vp]
    lib.mlt_device_ctx.argtypes = [vp, i32]
    lib.mlt_device_ctx.restype = vp
    lib.mlt_load_weights.argtypes = [vp, i32, vp, C.c_size_t]
    lib.mlt_arithmetic.argtypes = [vp, i32, C.POINTER(MltArithInfo)]
    lib.mlt_calibrate.argtypes = [vp, i32, vp, vp, vp, vp, i32, i32]
    lib.mlt_predict.argtypes = [vp, vp, i32, vp, i32, i32, C.c_int32, C.c_int32, vp, vp]
    lib.mlt_predict_batch.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.mlt_predict_batch_device.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.mlt_submit.argtypes = [vp, vp, i32, vp, i32, i32, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)]
    lib.mlt_flush.argtypes = [vp, i32]
    lib.mlt_wait.argtypes = [vp, i32, C.c_uint64, vp, vp]
    lib.mlt_synchronize.argtypes = [vp]
    lib.mlt_set_stream.argtypes = [vp, vp]
    lib.mlt_alloc_pinned.restype = vp
    lib.mlt_alloc_pinned.argtypes = [C.c_size_t]
    lib.mlt_free_pinned.argtypes = [vp]
    lib.mlt_num_logits.argtypes = [i32]
    lib.mlt_profile_enable.argtypes = [vp, i32]
    lib.mlt_profile_read.argtypes = [vp, C.POINTER(MltKernelTime), i32]
    lib.mlt_last_error.restype = C.c_char_p
    lib.mlt_last_error.argtypes = [vp]
    lib.mlt_shutdown.argtypes = [vp]
    lib.mlt_shutdown.restype = None
    _LIB = lib
    return lib


class MltCnn:
    """One context = one HIP device + one stream (one per encoder thread / EncCu instance)."""

    def __init__(self, device: int = 0, sizes=(128,), weights_dir: str | None = None, blobs: dict | None = None,
                 head_index: dict | None = None, max_batch: int = 4096, flags: int = 0, guard_margin: float = 0.0,
                 tolerance: float = 0.0, devices=None):
        """devices: list of HIP ordinals -> ONE context serving several GPUs (mlt_config.n_devices / devices[]); None: `device`."""
        self._lib = load_library()
        cfg = MltConfig()
        cfg.struct_size = C.sizeof(MltConfig)
        cfg.device = device
        if devices:
            cfg.n_devices = len(devices)
            for i, d in enumerate(devices):
                cfg.devices[i] = d
        cfg.weights_dir = weights_dir.encode() if weights_dir else None
        cfg.size_mask = sum(SIZE_BITS[s] for s in sizes)
        for i, s in enumerate((128, 64, 32, 16)):
            cfg.head_index[i] = (head_index or {}).get(s, -1)
        cfg.max_batch = max_batch
        cfg.flags = flags
        cfg.guard_margin = guard_margin
        cfg.tolerance = tolerance
        self._h = C.c_void_p()
        rc = self._lib.mlt_init(C.byref(cfg), C.byref(self._h))
        if rc != MLT_OK:
            raise MltError(rc, self._lib.mlt_last_error(None).decode())
        for s, blob in (blobs or {}).items():
            self.load_weights(s, blob)

    # -- lifecycle ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._lib.mlt_shutdown(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != MLT_OK:
            raise MltError(rc, self._lib.mlt_last_error(self._h).decode())

    def load_weights(self, size: int, blob: bytes):
        buf = C.create_string_buffer(blob, len(blob))
        self._check(self._lib.mlt_load_weights(self._h, size, buf, len(blob)))

    def calibrate(self, size: int, org: np.ndarray, pred: np.ndarray, poc, qp, replace: bool = False):
        """Repeat the load-time calibration of `size` with the caller's CUs appended to (or replacing) the synthetic set (mlt_calibrate)."""
        org = np.ascontiguousarray(org, np.int16)
        pred = np.ascontiguousarray(pred, np.int16)
        poc = np.ascontiguousarray(poc, np.int32)
        qp = np.ascontiguousarray(qp, np.int32)
        n = org.shape[0]
        assert org.shape == (n, size, size) and pred.shape == org.shape and poc.shape == (n,) and qp.shape == (n,)
        self._check(self._lib.mlt_calibrate(self._h, size, org.ctypes.data, pred.ctypes.data, poc.ctypes.data, qp.ctypes.data, n, 1 if replace else 0))

    def arithmetic(self, size: int) -> dict:
        """Arithmetic the size runs after loading (fast / exact), what the calibration measured, guard activity."""
        info = MltArithInfo()
        info.struct_size = C.sizeof(MltArithInfo)
        self._check(self._lib.mlt_arithmetic(self._h, size, C.byref(info)))
        return {k: getattr(info, k) for k, _ in MltArithInfo._fields_ if k != "struct_size"}

    def num_devices(self) -> int:
        return self._lib.mlt_num_devices(self._h)

    def arithmetic_of_device(self, index: int, size: int) -> dict:
        info = MltArithInfo()
        info.struct_size = C.sizeof(MltArithInfo)
        h = self._lib.mlt_device_ctx(self._h, index)
        if not h:
            raise MltError(1, "no such device index")
        rc = self._lib.mlt_arithmetic(h, size, C.byref(info))
        if rc != MLT_OK:
            raise MltError(rc, self._lib.mlt_last_error(h).decode())
        return {k: getattr(info, k) for k, _ in MltArithInfo._fields_ if k != "struct_size"}

    def num_logits(self, size: int) -> int:
        return self._lib.mlt_num_logits(size)

    # -- the call site (EncCu.cpp:806-921) ---------------------------------------------------
    def predict(self, org: np.ndarray, pred: np.ndarray, poc: int, qp: int):
        """org / pred: int16 2-D views [S, S] with arbitrary positive row stride (picture buffers)."""
        assert org.dtype == np.int16 and pred.dtype == np.int16 and org.ndim == 2 and org.shape == pred.shape
        S = org.shape[0]
        assert org.shape[1] == S and org.strides[1] == 2 and pred.strides[1] == 2
        split = C.c_int32(-1)  # the reference's "inference failed" value (EncModeCtrl.cpp:147-148)
        logits = np.zeros(self.num_logits(S) or 1, np.float32)
        self._check(self._lib.mlt_predict(self._h, org.ctypes.data, org.strides[0] // 2, pred.ctypes.data,
                                          pred.strides[0] // 2, S, int(poc), int(qp), C.byref(split), logits.ctypes.data))
        return int(split.value), logits

    def predict_batch(self, org: np.ndarray, pred: np.ndarray, poc, qp, want_logits: bool = True):
        org = np.ascontiguousarray(org, np.int16)
        pred = np.ascontiguousarray(pred, np.int16)
        n, S, _ = org.shape
        poc = np.ascontiguousarray(poc, np.int32)
        qp = np.ascontiguousarray(qp, np.int32)
        split = np.full((n,), -1, np.int32)
        logits = np.zeros((n, self.num_logits(S) or 1), np.float32) if want_logits else None
        self._check(self._lib.mlt_predict_batch(self._h, n, S, org.ctypes.data, pred.ctypes.data, poc.ctypes.data,
                                                qp.ctypes.data, split.ctypes.data, logits.ctypes.data if want_logits else None))
        return split, logits

    def submit(self, org: np.ndarray, pred: np.ndarray, poc: int, qp: int) -> int:
        """Deferred single-CU prediction: stage one CU, return a ticket (see mlt_submit in include/mltcnn.h)."""
        assert org.dtype == np.int16 and pred.dtype == np.int16 and org.ndim == 2 and org.shape == pred.shape
        S = org.shape[0]
        assert org.shape[1] == S and org.strides[1] == 2 and pred.strides[1] == 2
        t = C.c_uint64(0)
        self._check(self._lib.mlt_submit(self._h, org.ctypes.data, org.strides[0] // 2, pred.ctypes.data, pred.strides[0] // 2,
                                         S, int(poc), int(qp), C.byref(t)))
        return int(t.value)

    def flush(self, size: int):
        self._check(self._lib.mlt_flush(self._h, size))

    def wait(self, size: int, ticket: int):
        split = C.c_int32(-1)
        logits = np.zeros(self.num_logits(size) or 1, np.float32)
        self._check(self._lib.mlt_wait(self._h, size, C.c_uint64(ticket), C.byref(split), logits.ctypes.data))
        return int(split.value), logits

    def predict_batch_device(self, n: int, size: int, d_org: int, d_pred: int, d_poc: int, d_qp: int, d_split: int,
                             d_logits: int | None):
        """Raw device pointers (e.g. torch tensor .data_ptr()); asynchronous on the context's stream."""
        self._check(self._lib.mlt_predict_batch_device(self._h, n, size, d_org, d_pred, d_poc, d_qp, d_split, d_logits))

    def synchronize(self):
        self._check(self._lib.mlt_synchronize(self._h))

    def set_stream(self, hip_stream: int):
        self._check(self._lib.mlt_set_stream(self._h, hip_stream))

    def profile_enable(self, on: bool = True):
        self._check(self._lib.mlt_profile_enable(self._h, 1 if on else 0))

    def profile_read(self):
        arr = (MltKernelTime * 64)()
        k = self._lib.mlt_profile_read(self._h, arr, 64)
        return [dict(name=arr[i].name.decode(), launches=arr[i].launches, total_ms=arr[i].total_ms,
                     flops=arr[i].flops, bytes=arr[i].bytes) for i in range(max(k, 0))]
