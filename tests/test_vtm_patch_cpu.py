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
PATCH_N3 = os.path.join(ROOT, "patches", "vtm-mlt-cpp-mltcnn-n3.patch")   # opt-in encoder-side batching, applies on top of PATCH
N3_EXTRA = ["source/Lib/EncoderLib/EncModeCtrl.h"]
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


def test_n3_patch_applies_on_top_with_zero_fuzz_and_compiles(patched, tmp_path):
    """SURVEY 8(f) N3, encoder half (round 4): patches/vtm-mlt-cpp-mltcnn-n3.patch -- probe and replay over WPP anti-diagonals inside
    EncSlice::encodeCtus -- is what the generator writes, applies to the N1-patched tree with zero fuzz, keeps the setNewModeList call where
    it was, and the files it touches pass g++ -fsyntax-only against the reference's headers."""
    out = tmp_path / "regen_n3.patch"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_vtm_patch.py"), "--n3", "--out", str(out)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert out.read_bytes() == open(PATCH_N3, "rb").read(), "patches/vtm-mlt-cpp-mltcnn-n3.patch is stale: run tools/make_vtm_patch.py --n3"
    d = tmp_path / "tree"
    shutil.copytree(patched, d)
    for f in N3_EXTRA:
        os.makedirs(os.path.dirname(d / f), exist_ok=True)
        shutil.copy(os.path.join(REF, f), d / f)
    # (a header reached through another header of the same directory is looked up next to THAT header first: the untouched EncoderLib
    # headers must sit beside the patched ones for the syntax check)
    for h in os.listdir(os.path.join(REF, "source/Lib/EncoderLib")):
        if h.endswith(".h") and not (d / "source/Lib/EncoderLib" / h).exists():
            shutil.copy(os.path.join(REF, "source/Lib/EncoderLib", h), d / "source/Lib/EncoderLib" / h)
    r = subprocess.run(["patch", "-p1", "--fuzz=0", "--no-backup-if-mismatch", "-i", PATCH_N3], cwd=d, capture_output=True, text=True)
    assert r.returncode == 0 and "fuzz" not in r.stdout and "FAILED" not in r.stdout, r.stdout + r.stderr
    src = (d / FILES[0]).read_text()
    assert src.count("m_modeCtrl->setNewModeList(*tempCS, partitioner, predictedSplitMode, currTestMode.qp);") == 1
    assert "m_cnnSplitPredictor->submitSplitMode(" in src and "m_cnnSplitPredictor->waitSplitMode(" in src and "m_modeCtrl->abortCTU();" in src
    sl = (d / FILES[4]).read_text()
    assert "MLTCNN_BATCH" in sl and "rowCtx[ctuYPosInCtus] = pCABACWriter->getCtx();" in sl
    inc = [str(d / "source/Lib/EncoderLib"), str(d / "source/Lib"), REF + "/source/Lib/EncoderLib", REF + "/source/Lib",
           REF + "/source/Lib/CommonLib", os.path.join(ROOT, "host"), os.path.join(ROOT, "include")]
    for f in (FILES[0], FILES[4]):
        cmd = ["g++", "-std=c++14", "-fsyntax-only", "-w"] + [x for i in inc for x in ("-I", i)] + [str(d / f)]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, (f, r.stderr[-3000:])
