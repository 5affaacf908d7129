"""CPU: the C-ABI library loads and exports every symbol include/mltcnn.h declares; host logic that
needs no GPU (argument checks, the no-device error path).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib(pkg):
    pkg.build.build_lib()
    return pkg.capi.load_library()


def test_exports_match_header(pkg, lib):
    header = open(os.path.join(ROOT, "include", "mltcnn.h")).read()
    declared = set(re.findall(r"\b(mlt_[a-z_0-9]+)\s*\(", header))
    declared -= {"mlt_config", "mlt_ctx", "mlt_kernel_time", "mlt_arith_info"}
    assert declared == set(pkg.capi.EXPORTS), declared ^ set(pkg.capi.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in mltcnn.h but not exported"


def test_abi_version_and_logit_counts(lib):
    assert lib.mlt_abi_version() == 4
    assert [lib.mlt_num_logits(s) for s in (128, 64, 32, 16, 8)] == [9, 15, 15, 15, 0]


def test_config_struct_layout(pkg):
    # must match `struct mlt_config` in include/mltcnn.h (x86-64 SysV)
    assert C.sizeof(pkg.capi.MltConfig) == 96   # ABI 3: the 56-byte ABI-2 struct + n_devices + devices[8] (+ tail padding)
    assert pkg.capi.MltConfig.n_devices.offset == 56 and pkg.capi.MltConfig.devices.offset == 60
    assert pkg.capi.MltConfig.guard_margin.offset == 44
    assert pkg.capi.MltConfig.tolerance.offset == 48
    assert C.sizeof(pkg.capi.MltArithInfo) == 88   # ABI 4 (72 bytes: struct_size in front of the ABI-3 fields) + round 6's magnitude-guard fields (the library writes only what fits)
    assert pkg.capi.MltArithInfo.mag_guard_thr.offset == 72 and pkg.capi.MltArithInfo.mag_guard_flagged.offset == 76 and pkg.capi.MltArithInfo.mag_guard_kind.offset == 80
    assert pkg.capi.MltArithInfo.struct_size.offset == 0 and pkg.capi.MltArithInfo.guard_reruns.offset == 32
    assert pkg.capi.MltArithInfo.w2_stages.offset == 40 and pkg.capi.MltArithInfo.guard_margin.offset == 44 and pkg.capi.MltArithInfo.rounding.offset == 60 and pkg.capi.MltArithInfo.calib_caller_cus.offset == 68
    assert pkg.capi.MltConfig.weights_dir.offset == 8
    assert pkg.capi.MltConfig.head_index.offset == 20
    assert C.sizeof(pkg.capi.MltKernelTime) == 72


def test_init_rejects_bad_config(pkg, lib):
    h = C.c_void_p()
    cfg = pkg.capi.MltConfig()
    cfg.struct_size = 4
    assert lib.mlt_init(C.byref(cfg), C.byref(h)) == 1  # MLT_ERR_ARG
    assert lib.mlt_init(None, C.byref(h)) == 1


def test_device_list_and_abi2_struct_are_validated_before_any_device_is_touched(pkg, lib):
    """mlt_config carries a device list (ABI 3).  ABI 4 is a hard break: the 56-byte ABI-2 struct is REJECTED (its binary would also hand
    mlt_arithmetic a 32-byte mlt_arith_info), a device count beyond MLT_MAX_DEVICES is an argument error, and -- without a GPU -- a
    well-formed list fails with MLT_ERR_NO_DEVICE, not a crash."""
    import torch
    h = C.c_void_p()
    cfg = pkg.capi.MltConfig()
    cfg.struct_size = C.sizeof(pkg.capi.MltConfig)
    cfg.n_devices = 9
    assert lib.mlt_init(C.byref(cfg), C.byref(h)) == 1        # MLT_ERR_ARG
    cfg.n_devices = -1
    assert lib.mlt_init(C.byref(cfg), C.byref(h)) == 1
    assert lib.mlt_num_devices(None) == 0 and not lib.mlt_device_ctx(None, 0)
    cfg.struct_size = 56                                      # an ABI-2 caller
    cfg.n_devices = 0
    assert lib.mlt_init(C.byref(cfg), C.byref(h)) == 1        # MLT_ERR_ARG, before any device is touched
    assert b"ABI 4" in lib.mlt_last_error(None)
    cfg.struct_size = C.sizeof(pkg.capi.MltConfig)
    if not torch.cuda.is_available():
        cfg.n_devices = 2
        cfg.devices[0], cfg.devices[1] = 0, 1
        assert lib.mlt_init(C.byref(cfg), C.byref(h)) == 2    # MLT_ERR_NO_DEVICE


def test_no_gpu_fails_loudly(pkg):
    """No CPU fallback: without a HIP device the product path must refuse to run."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.MltError) as ei:
        pkg.MltCnn(device=0, sizes=(128,))
    assert ei.value.code == 2  # MLT_ERR_NO_DEVICE


def test_null_ctx_calls_are_errors_not_crashes(lib):
    assert lib.mlt_synchronize(None) == 1
    assert lib.mlt_predict_batch_device(None, 1, 128, None, None, None, None, None, None) == 1
    lib.mlt_shutdown(None)
    lib.mlt_free_pinned(None)


def test_bench_gpus_flag_starts_its_own_ranks(monkeypatch, capsys):
    """`python bench.py --gpus N` outside torchrun must start N ranks as a CHILD process (never exec, nothing GPU-related touched
    first), relay rank 0's single JSON line and return the child's exit code; inside torchrun (WORLD_SIZE set) it must not."""
    import importlib
    import sys
    import types
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    calls = {}

    def fake_run(cmd, env=None, stdout=None, text=None):
        calls["cmd"], calls["env"] = cmd, env
        return types.SimpleNamespace(returncode=0, stdout='noise\n{"metric": "m", "n_gpus": 4}\n')

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    args = bench.parse_args()
    assert bench.launch_ranks(args) == 0
    cmd = calls["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert calls["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    out = capsys.readouterr()
    assert out.out.strip() == '{"metric": "m", "n_gpus": 4}' and "noise" in out.err
    # a failing child (or a child that printed no / several JSON lines) is a failure of the command, never a 1-GPU fallback
    monkeypatch.setattr(bench.subprocess, "run", lambda *a, **k: types.SimpleNamespace(returncode=0, stdout="no json here\n"))
    assert bench.launch_ranks(args) != 0
    monkeypatch.setattr(bench.subprocess, "run", lambda *a, **k: types.SimpleNamespace(returncode=7, stdout='{"a": 1}\n'))
    assert bench.launch_ranks(args) == 7


def test_bench_refuses_a_world_size_that_is_not_gpus():
    """Inside a torchrun launch (WORLD_SIZE set) `--gpus N` must equal the world size: bench.py exits 2 with a message BEFORE any GPU call
    (this container has no GPU -- reaching torch.cuda would trip the `no CPU fallback` assertion instead) and prints no JSON line."""
    import subprocess
    import sys
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--warmup", "0"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 2, (r.returncode, r.stderr[-400:])
    assert "--gpus 8 but WORLD_SIZE=2" in r.stderr and "{" not in r.stdout


def test_library_carries_the_signature_of_the_sources_in_the_tree(pkg, lib, tmp_path):
    """VERDICT r5 item 3 / ADVICE r5: the build's dependency list used to omit the two hottest kernel files, so an edit to them could run against
    an older libmltcnn_hip.so (which travels to the GPU box).  Now: every file bench.py's source signature hashes is a dependency of the
    build, the binary carries the signature it was built from (mlt_build_signature = the marker build.stale() greps for), the host side refuses
    a library whose signature differs from the tree's, and an edit to ANY csrc file makes the build stale."""
    import bench
    b = pkg.build
    deps = set(b.dependencies())
    hashed = set(b.signature_files())
    assert hashed and hashed <= deps
    csrc_files = {os.path.join(b.CSRC, f) for f in os.listdir(b.CSRC) if f.endswith((".hip", ".inc", ".cpp", ".h"))}
    assert csrc_files == hashed, csrc_files ^ hashed
    for inc in re.findall(r'#include "([^"]+\.inc)"', open(os.path.join(b.CSRC, "mlt_kernels.hip")).read()):
        assert os.path.join(b.CSRC, inc) in deps, inc          # mlt_layer0_kernel.inc, mlt_layer1_kernel.inc, ...
    assert b.ABI_HEADER in deps
    sig = b.source_signature()
    assert bench.source_signature() == sig and re.fullmatch(r"[0-9a-f]{16}", sig)
    assert lib.mlt_build_signature().decode() == sig == b.built_signature()
    assert not b.stale()
    # an edit to a kernel include changes the signature -> stale; and a library built from other sources is refused when it is loaded
    probe = os.path.join(b.CSRC, "mlt_layer1_kernel.inc")
    orig = open(probe, "rb").read()
    try:
        open(probe, "ab").write(b"\n// edit\n")
        assert b.source_signature() != sig and b.stale()
        saved, pkg.capi._LIB = pkg.capi._LIB, None
        try:
            with pytest.raises(RuntimeError, match="was built from sources"):
                pkg.capi.load_library()
        finally:
            pkg.capi._LIB = saved
    finally:
        open(probe, "wb").write(orig)
    assert not b.stale()


def test_calibration_set_hook(lib):
    """The synthetic calibration set the load-time calibration prices (host-only hook, used by tools/attribute_error.py): 560 CUs in six content
    classes, 10-bit Pels, POC / QP in the encoder's range, deterministic."""
    import numpy as np
    S = 128
    lib.mlt_calibration_set_copy.argtypes = [C.c_int] + [C.c_void_p] * 5
    assert lib.mlt_calibration_set_copy(8, None, None, None, None, None) == -1
    org = np.zeros((560, S, S), np.int16); pred = np.zeros_like(org)
    poc = np.zeros(560, np.int32); qp = np.zeros(560, np.int32); cls = np.zeros(560, np.int32)
    assert lib.mlt_calibration_set_copy(S, org.ctypes.data, pred.ctypes.data, poc.ctypes.data, qp.ctypes.data, cls.ctypes.data) == 560
    assert np.bincount(cls).tolist() == [160, 80, 80, 80, 80, 80]
    assert org.min() >= 0 and org.max() <= 1023 and pred.min() >= 0 and pred.max() <= 1023
    assert poc.min() >= 0 and poc.max() <= 600 and qp.min() >= 17 and qp.max() <= 47
    assert all(len(np.unique(org[i])) == 1 for i in np.flatnonzero(cls == 2)[:8])     # class 2: constant org
    assert all(len(np.unique(pred[i])) == 1 for i in np.flatnonzero(cls == 3)[:8])    # class 3: constant pred
    o2 = np.zeros_like(org)
    assert lib.mlt_calibration_set_copy(S, o2.ctypes.data, None, None, None, None) == 560 and np.array_equal(o2, org)


def test_host_restatement_of_the_flat_statistic_uses_the_device_range(pkg):
    """synth.flat_quad_fraction is the host restatement of the flat-content guard's statistic (probes, flag-rate estimates); its default range must be the
    device's MLT_FLAT_RANGE (round 6: 6; the near-flat classes of the tests are built around it: amplitude-4 texture 69 - 73 % near-flat, +-1 LSB dither 100 %,
    natural scenes below 1/2)."""
    import inspect
    import os
    import re
    import numpy as np
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "fastintercu-vvc_amd", "csrc", "mlt_kernels.h")).read()
    dev = int(re.search(r"^#define MLT_FLAT_RANGE (\d+)", hdr, re.M).group(1))
    host = inspect.signature(pkg.synth.flat_quad_fraction).parameters["flat_range"].default
    assert dev == host == 6
    S = pkg.synth
    o, p = S.make_patches(128, 8, 5, S.KIND_LOW_CONTRAST)
    f = S.flat_quad_fraction(o, p)
    assert f.min() >= 0.6 and f.max() <= 0.8                      # still flagged (>= 1/2), no longer 100 %
    o, p = S.make_patches(128, 8, 5, S.KIND_DITHER)
    assert S.flat_quad_fraction(o, p).min() >= 0.99
    o, p = S.natural_patches(128, 256, 4242)
    assert (S.flat_quad_fraction(o, p) >= 0.5).sum() == 0          # (range 8 flagged 2 - 3 % of this class)
    o, p = S.make_patches(128, 8, 5, S.KIND_PARTIAL_NEAR_FLAT)
    assert S.flat_quad_fraction(o, p).max() < 0.5
