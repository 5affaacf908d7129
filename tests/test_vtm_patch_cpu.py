"""CPU, build container only: the SURVEY 8(f) N1 integration patch (patches/vtm-mlt-cpp-mltcnn.patch) applies to the reference
encoder tree with zero fuzz, removes LibTorch / OpenCV from the call site and the build files, leaves
EncModeCtrl::setNewModeList's call untouched, and the patched EncCu.cpp passes g++ -fsyntax-only against the reference's own
headers plus this repository's host/ and include/ (nothing of the reference is copied into the repo or shipped to the GPU box)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/vtm-mlt-cpp"
PATCH = os.path.join(ROOT, "patches", "vtm-mlt-cpp-mltcnn.patch")
FILES = ["source/Lib/EncoderLib/EncCu.cpp", "source/Lib/EncoderLib/EncCu.h", "CMakeLists.txt", "source/Lib/EncoderLib/CMakeLists.txt",
         "source/Lib/EncoderLib/EncSlice.cpp"]

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not mounted (GPU box)")


@pytest.fixture(scope="module")
def patched(tmp_path_factory):
    d = tmp_path_factory.mktemp("vtm")
    for f in FILES:
        os.makedirs(os.path.dirname(d / f), exist_ok=True)
        shutil.copy(os.path.join(REF, f), d / f)
    r = subprocess.run(["patch", "-p1", "--fuzz=0", "--no-backup-if-mismatch", "-i", PATCH], cwd=d, capture_output=True, text=True)
    assert r.returncode == 0 and "fuzz" not in r.stdout and "FAILED" not in r.stdout, r.stdout + r.stderr
    return d


def test_patch_is_what_the_generator_writes(tmp_path):
    out = tmp_path / "regen.patch"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_vtm_patch.py"), "--out", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == open(PATCH, "rb").read(), "patches/vtm-mlt-cpp-mltcnn.patch is stale: run tools/make_vtm_patch.py"


def test_patch_replaces_torch_and_opencv_and_keeps_the_mode_list_call(patched):
    src = (patched / FILES[0]).read_text()
    ref = open(os.path.join(REF, FILES[0])).read()
    for gone in ("torch::", "torch/script.h", "opencv2", "cv::", "xMalloc(uint16_t", "torch_model/MLTORPQ_splitMode_", "c10::"):
        assert gone not in src, gone
    call = "m_modeCtrl->setNewModeList(*tempCS, partitioner, predictedSplitMode, currTestMode.qp);"
    assert src.count(call) == 1 and ref.count(call) == 1                      # EncCu.cpp:928 untouched
    assert src.count("int predictedSplitMode = -1;") == 1                     # :694, the failure value the error contract relies on
    assert "m_cnnSplitPredictor->predictSplitMode(orgY.buf, orgY.stride, predY.buf, predY.stride, cuw, poc, cuQP)" in src
    assert "m_cnnSplitPredictor->gate(partitioner.chType" in src
    for f in (FILES[2], FILES[3]):
        cm = (patched / f).read_text()
        assert "find_package(Torch" not in cm and "TORCH_LIBRARIES" not in cm and "find_package(OpenCV" not in cm and "OpenCV_LIBRARIES" not in cm
        assert "mltcnn" in cm.lower()


def test_patched_enccu_compiles_against_the_reference_headers(patched):
    inc = [str(patched / "source/Lib/EncoderLib"), str(patched / "source/Lib"), REF + "/source/Lib/EncoderLib", REF + "/source/Lib",
           REF + "/source/Lib/CommonLib", os.path.join(ROOT, "host"), os.path.join(ROOT, "include")]
    cmd = ["g++", "-std=c++14", "-fsyntax-only", "-w"] + [x for i in inc for x in ("-I", i)] + [str(patched / FILES[0])]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
