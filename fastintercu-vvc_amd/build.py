"""Build recipe for the HIP extension (in-tree, gfx950 only): csrc/*.hip + *.cpp -> libmltcnn_hip.so."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmltcnn_hip.so")
SOURCES = ["mlt_kernels.hip", "mlt_model.cpp", "mlt_api.cpp"]
HEADERS = ["mlt_kernels.h", "mlt_model.h", "mlt_tier_search.h", "mlt_conv_kernels.inc", "mlt_chain_kernel.inc", "mlt_front_kernels.inc", "mlt_tail_kernels.inc",
           os.path.join("..", "..", "include", "mltcnn.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False, defines=(), out: str | None = None) -> str:
    """Cross-compiles without a GPU (hipcc only needs the gfx950 target).
    `defines` / `out`: tuning variants (scripts/sweep_cfg.py); the product is always LIB with no defines."""
    target = out or LIB
    if force or out or stale():
        cmd = [hipcc()] + FLAGS + [f"-D{d}" for d in defines] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", target]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return target
