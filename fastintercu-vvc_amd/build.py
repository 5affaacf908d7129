"""Build recipe for the HIP extension (in-tree, gfx950 only): csrc/*.hip + *.cpp -> libmltcnn_hip.so.

The library carries the signature of the sources it was built from (`MLTCNN_SOURCE_SIG=<16 hex>` in its read-only data, also returned by
`mlt_build_signature()`): `stale()` and `capi.load_library()` compare it with the sources in the tree, so an edit to ANY file of csrc/ --
kernel includes as much as translation units -- can neither be skipped by the build nor run against an older binary."""
from __future__ import annotations

import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmltcnn_hip.so")
ABI_HEADER = os.path.normpath(os.path.join(HERE, "..", "include", "mltcnn.h"))
SOURCES = ["mlt_kernels.hip", "mlt_model.cpp", "mlt_api.cpp"]
SOURCE_EXTS = (".hip", ".inc", ".cpp", ".h")
SIG_MARKER = b"MLTCNN_SOURCE_SIG="
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def dependencies() -> list[str]:
    """Every file the library is compiled from: all of csrc/ (translation units, headers, kernel includes) + the C ABI header."""
    return [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(SOURCE_EXTS)] + [ABI_HEADER]


def signature_files() -> list[str]:
    """What `source_signature()` hashes (bench.py keys profiles/pmc_traffic.json on it): the files of csrc/."""
    return [d for d in dependencies() if d != ABI_HEADER]


def source_signature() -> str:
    """sha256 over the kernel + runtime sources (every file of csrc/, sorted by name), first 16 hex digits."""
    h = hashlib.sha256()
    for path in signature_files():
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def built_signature(path: str | None = None) -> str | None:
    """The signature a built library carries (None: no library, or one from before the marker existed)."""
    path = path or LIB
    if not os.path.exists(path):
        return None
    with open(path, "rb") as fh:
        data = fh.read()
    i = data.find(SIG_MARKER)
    if i < 0:
        return None
    sig = data[i + len(SIG_MARKER): i + len(SIG_MARKER) + 16]
    return sig.decode("ascii", "replace")


def stale() -> bool:
    if not os.path.exists(LIB):
        return True
    if built_signature() != source_signature():
        return True
    return os.path.getmtime(ABI_HEADER) > os.path.getmtime(LIB)


def _unit_key(src: str, defines) -> str:
    """Cache key of one translation unit's object file: its own bytes, every header / kernel include of csrc/ (they are few and a unit sees
    most of them), the C ABI header, the flags and the defines."""
    h = hashlib.sha256()
    h.update(" ".join(FLAGS + list(defines)).encode())
    for path in [os.path.join(CSRC, src)] + [d for d in dependencies() if d.endswith((".h", ".inc"))]:
        if path.endswith(".inc") and not src.endswith(".hip"):
            continue  # the kernel includes belong to mlt_kernels.hip alone
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:20]


def build_lib(force: bool = False, verbose: bool = False, defines=(), out: str | None = None) -> str:
    """Cross-compiles without a GPU (hipcc only needs the gfx950 target).  One object per translation unit, cached under _build/ by the hash of
    everything the unit includes (an edit to the host runtime does not recompile the kernels: 85 s -> 12 s); the source signature is a
    define of mlt_api.cpp alone.
    `defines` / `out`: tuning variants (scripts/sweep_cfg.py); the product is always LIB with no defines."""
    target = out or LIB
    if force or out or stale():
        cache = os.path.join(HERE, "_build")
        os.makedirs(cache, exist_ok=True)
        cflags = [f for f in FLAGS if f != "-shared"]
        objs, procs = [], []
        for src in SOURCES:
            sig_def = [f'-DMLT_SOURCE_SIG="{source_signature()}"'] if src == "mlt_api.cpp" else []
            obj = os.path.join(cache, f"{os.path.splitext(src)[0]}.{_unit_key(src, list(defines) + sig_def)}.o")
            objs.append(obj)
            if force or not os.path.exists(obj):
                cmd = [hipcc()] + cflags + sig_def + [f"-D{d}" for d in defines] + ["-c", os.path.join(CSRC, src), "-o", obj + ".tmp"]
                if verbose:
                    print(" ".join(cmd))
                procs.append((obj, subprocess.Popen(cmd)))
        for obj, p in procs:  # the units compile side by side
            if p.wait() != 0:
                raise subprocess.CalledProcessError(p.returncode, "hipcc -c (see its diagnostics above)")
            os.replace(obj + ".tmp", obj)
        cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-fvisibility=hidden"] + objs + ["-o", target + ".tmp"]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
        os.replace(target + ".tmp", target)
        keep = set(objs)
        for f in os.listdir(cache):  # objects of older sources (variants excepted: they come and go with their sweep)
            if not out and f.endswith(".o") and os.path.join(cache, f) not in keep:
                os.remove(os.path.join(cache, f))
    return target
