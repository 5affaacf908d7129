"""CPU: weight converter round trip and the C++ host mirror (builds with plain g++, links the C ABI)."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_blob_round_trip_and_checkpoint_conventions(pkg, tmp_path):
    sd = pkg.synth.make_state_dict(1, 3)
    blob = pkg.weights.pack_blob(1, sd)
    arch, back = pkg.weights.unpack_blob(blob)
    assert arch == 1
    for k, v in sd.items():
        if k.endswith("num_batches_tracked") or k.startswith("bn1."):
            continue
        assert np.array_equal(back[k], v), k
    # model2torchScript.py:23-32 conventions: {'params': ...} wrapper and 'module.' prefixes
    wrapped = {"params": {"module." + k: v for k, v in sd.items()}}
    assert pkg.weights.from_checkpoint(wrapped, 1) == blob


def test_convert_weights_cli(pkg, tmp_path):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "convert_weights.py"), "--size", "32", "--synthetic", "10", str(tmp_path)],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    data = open(tmp_path / "MLTORPQ_splitMode_32.mltw", "rb").read()
    assert data == pkg.weights.synthetic_blob(1, 10)


def test_cpp_host_mirror_builds_and_links(pkg, tmp_path):
    lib = pkg.build.build_lib()
    exe = str(tmp_path / "callsite_demo")
    cmd = ["g++", "-std=c++17", "-Wall", "-Werror", os.path.join(ROOT, "host", "callsite_demo.cpp"), "-o", exe,
           "-L" + os.path.dirname(lib), "-lmltcnn_hip", "-Wl,-rpath," + os.path.dirname(lib)]
    assert subprocess.run(cmd, capture_output=True, text=True).returncode == 0
    out = subprocess.run([exe], capture_output=True, text=True, env=dict(os.environ, LD_LIBRARY_PATH="/opt/rocm/lib"))
    assert out.returncode == 0 and "abi 2, logits(128) 9, logits(32) 15" in out.stdout
