"""MLTW weight blob: the hand-off format between the training side and the HIP runtime.

The reference hands weights to the encoder as a TorchScript file produced from a
`.pth` checkpoint (`mlt-cnn-python/codes/model2torchScript.py:22-48`: `load_net['params']`,
strip `module.`, `load_state_dict(strict=False)`, trace, save) and the encoder
deserialises it on EVERY call (`EncCu.cpp:894-900`).  Here the hand-off is a flat,
self-describing blob of the raw fp32 state_dict (same key names as the reference
module), read ONCE by `mlt_init`; BN folding / fp16 packing happen inside the runtime.

Layout (little endian):
  char  magic[4] = "MLTW"; u32 version = 1; u32 arch (0 = CTU/128, 1 = CU/64-32-16); u32 n_tensors
  n_tensors x { char name[64]; u32 ndim; u32 dims[4]; u64 offset_floats; u64 numel }
  float data[]   (concatenated, offset_floats indexes into it)
`num_batches_tracked` entries are dropped (integer bookkeeping, unused in eval).
"""
from __future__ import annotations

import struct
from collections import OrderedDict

import numpy as np

from . import synth

MAGIC = b"MLTW"
VERSION = 1
_ENTRY = struct.Struct("<64sI4IQQ")
_HEADER = struct.Struct("<4sIII")


def pack_blob(arch: int, state_dict) -> bytes:
    """state_dict: mapping name -> array-like (numpy or torch tensor)."""
    spec = synth.state_dict_spec(arch)
    entries = []
    chunks = []
    off = 0
    for key, shape in spec.items():
        if key.endswith("num_batches_tracked"):
            continue
        if key not in state_dict:
            if key.startswith("bn1."):  # top-level bn1 is dead weight (arch:247,277-278)
                continue
            raise KeyError(f"state_dict is missing '{key}'")
        v = state_dict[key]
        if hasattr(v, "detach"):
            v = v.detach().cpu().numpy()
        a = np.ascontiguousarray(np.asarray(v, dtype=np.float32))
        if tuple(a.shape) != tuple(shape):
            raise ValueError(f"'{key}': shape {a.shape} != expected {shape}")
        dims = list(a.shape) + [1] * (4 - a.ndim)
        entries.append(_ENTRY.pack(key.encode(), a.ndim, *dims, off, a.size))
        chunks.append(a.tobytes())
        off += a.size
    head = _HEADER.pack(MAGIC, VERSION, arch, len(entries))
    return head + b"".join(entries) + b"".join(chunks)


def unpack_blob(blob: bytes):
    magic, version, arch, n = _HEADER.unpack_from(blob, 0)
    if magic != MAGIC or version != VERSION:
        raise ValueError("not an MLTW v1 blob")
    pos = _HEADER.size
    data_start = pos + n * _ENTRY.size
    data = np.frombuffer(blob, dtype=np.float32, offset=data_start)
    sd = OrderedDict()
    for _ in range(n):
        name, ndim, d0, d1, d2, d3, off, numel = _ENTRY.unpack_from(blob, pos)
        pos += _ENTRY.size
        shape = (d0, d1, d2, d3)[:ndim]
        sd[name.rstrip(b"\0").decode()] = data[off:off + numel].reshape(shape)
    return arch, sd


def from_checkpoint(obj, arch: int) -> bytes:
    """`.pth` (dict with 'params') or a bare state_dict -> blob, following
    model2torchScript.py:23-32 ('params' key, optional 'module.' prefix)."""
    sd = obj["params"] if isinstance(obj, dict) and "params" in obj else obj
    clean = OrderedDict()
    for k, v in sd.items():
        clean[k[7:] if k.startswith("module.") else k] = v
    return pack_blob(arch, clean)


def synthetic_blob(arch: int, weight_seed: int) -> bytes:
    return pack_blob(arch, synth.make_state_dict(arch, weight_seed))


def amplifying_blob(arch: int, weight_seed: int, resi_gain: float = 16.0, org_gain: float = 0.25) -> bytes:
    """A deterministic stand-in for what TRAINING does to the first layer (round 6; tools/attribute_error.py): the seeded set with the stem's
    residual-plane filters scaled up and its org-plane filters scaled down.  Like a trained set (tools/train_synth_weights.py) it amplifies the
    residual plane, so content with residuals of hundreds of ten-bit steps -- the synthetic calibration classes "uniform", "constant org / pred"
    -- produces features, logits and absolute fp16 errors many times those of ordinary content: the weight family the MAGNITUDE guard exists for
    (mlt_arith_info.mag_guard_thr), available to tests and probes without a training run."""
    sd = {k: np.array(v, copy=True) for k, v in synth.make_state_dict(arch, weight_seed).items()}
    sd["conv1.weight"][:, 1] *= np.float32(resi_gain)
    sd["conv1.weight"][:, 0] *= np.float32(org_gain)
    return pack_blob(arch, sd)
