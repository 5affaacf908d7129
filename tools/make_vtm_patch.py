#!/usr/bin/env python3
"""Generate patches/vtm-mlt-cpp-mltcnn.patch: the SURVEY.md 8(f) N1 integration patch against the reference encoder
(/root/reference/vtm-mlt-cpp, the authors' VTM-11.0 tree).  Build container only (needs the reference tree); the patch
file it writes is what ships.

What the patch does (reference lines):
  source/Lib/EncoderLib/EncCu.cpp:60-65     drop <torch/script.h> / OpenCV includes, include host/mlt_split_predictor.hpp
  source/Lib/EncoderLib/EncCu.cpp:160-206   EncCu::destroy  -> delete the predictor (mlt_shutdown)
  source/Lib/EncoderLib/EncCu.cpp:233-259   EncCu::init     -> create the predictor ONCE (weights dir / device / size mask from
                                            MLTCNN_WEIGHTS_DIR / MLTCNN_DEVICE / MLTCNN_SIZE_MASK / MLTCNN_FLAGS; the reference re-loads the .pt per CU, :894-900)
  source/Lib/EncoderLib/EncCu.cpp:746-756   gate            -> SplitPredictor::gate (same condition, size mask instead of the commented-out clauses)
  source/Lib/EncoderLib/EncCu.cpp:801-927   gather / absdiff / normalise / tensor assembly / jit::load / forward / argmax
                                            -> ONE call: predictSplitMode(org buf+stride, pred buf+stride, cuw, poc, qp)
  source/Lib/EncoderLib/EncCu.cpp:928       m_modeCtrl->setNewModeList(...)  UNTOUCHED
  source/Lib/EncoderLib/EncCu.h             forward declaration + member
  CMakeLists.txt:58-66, source/Lib/EncoderLib/CMakeLists.txt:35-40   Torch / OpenCV stanzas -> MLTCNN_ROOT include dirs + libmltcnn_hip.so
  source/Lib/EncoderLib/EncSlice.cpp:51-54  four unused OpenCV includes removed

usage: python tools/make_vtm_patch.py [--ref /root/reference/vtm-mlt-cpp] [--out patches/vtm-mlt-cpp-mltcnn.patch]
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["source/Lib/EncoderLib/EncCu.cpp", "source/Lib/EncoderLib/EncCu.h", "CMakeLists.txt", "source/Lib/EncoderLib/CMakeLists.txt",
         "source/Lib/EncoderLib/EncSlice.cpp"]


def replace_once(text, old, new, what):
    if text.count(old) != 1:
        raise SystemExit(f"make_vtm_patch: anchor for '{what}' found {text.count(old)} times (reference tree differs from the surveyed one)")
    return text.replace(old, new)


def cut(text, first, last, new, what):
    """Replace everything from the line containing `first` through the line containing `last` (both unique) by `new`."""
    if text.count(first) != 1 or text.count(last) != 1:
        raise SystemExit(f"make_vtm_patch: anchors for '{what}' not unique")
    a = text.rfind("\n", 0, text.index(first)) + 1
    b = text.index("\n", text.index(last)) + 1
    if b <= a:
        raise SystemExit(f"make_vtm_patch: anchors for '{what}' out of order")
    return text[:a] + new + text[b:]


def patch_enccu_cpp(t):
    t = cut(t, "#include <torch/script.h>", "#include <opencv2/highgui.hpp>",
            '#include <cstdlib>\n#include <string>\n\n'
            '// MI355X-native MLT-CNN split predictor behind a C ABI (replaces LibTorch + OpenCV)\n'
            '#include "mlt_split_predictor.hpp"\n', "includes")
    t = replace_once(t, "    delete m_modeCtrl;\n    m_modeCtrl = nullptr;\n",
                     "    delete m_modeCtrl;\n    m_modeCtrl = nullptr;\n\n"
                     "    delete m_cnnSplitPredictor;   // mlt_shutdown: weights, workspaces, streams\n"
                     "    m_cnnSplitPredictor = nullptr;\n", "destroy")
    t = replace_once(t, "    m_pcIntraSearch->setModeCtrl( m_modeCtrl );\n\n}\n",
                     "    m_pcIntraSearch->setModeCtrl( m_modeCtrl );\n\n"
                     "    /// CNN split predictor: weights are loaded, folded and packed ONCE here (one context per EncCu / thread)\n"
                     "    if( !m_cnnSplitPredictor )\n"
                     "    {\n"
                     "        const char *dir  = std::getenv( \"MLTCNN_WEIGHTS_DIR\" );   // holds MLTORPQ_splitMode_<S>.mltw\n"
                     "        const char *dev  = std::getenv( \"MLTCNN_DEVICE\" );\n"
                     "        const char *mask = std::getenv( \"MLTCNN_SIZE_MASK\" );     // MLT_SIZE_* bits; default: 128x128 only\n"
                     "        const char *flg  = std::getenv( \"MLTCNN_FLAGS\" );         // MLT_FLAG_* bits; default: the decision guard (split modes are what the encoder consumes)\n"
                     "        m_cnnSplitPredictor = new mlt::SplitPredictor( dir ? dir : \"./torch_model\", dev ? std::atoi( dev ) : 0,\n"
                     "                                                     mask ? (uint32_t) std::strtoul( mask, nullptr, 0 ) : MLT_SIZE_128,\n"
                     "                                                     flg ? (uint32_t) std::strtoul( flg, nullptr, 0 ) : MLT_FLAG_DECISION_GUARD );\n"
                     "    }\n\n}\n", "init")
    t = cut(t, "if (partitioner.chType == 0 && tempCS->slice->getSliceType() != I_SLICE)", "useCNN = true;",
            "        if (m_cnnSplitPredictor && m_cnnSplitPredictor->ok())\n"
            "            useCNN = m_cnnSplitPredictor->gate(partitioner.chType, tempCS->slice->getSliceType() == I_SLICE, cux, cuy, cuw, cuh,\n"
            "                                               tempCS->slice->getPic()->Y().width, tempCS->slice->getPic()->Y().height);\n", "gate")
    t = cut(t, "/// --- Device Setting --- ///", "std::cerr << \"error\\n\";",
            "                /// --- CNN Input - vector --- ///\n"
            "                int poc  = bestCS->slice->getPOC();\n"
            "                int cuQP = currTestMode.qp;\n"
            "                /// --- CNN Input - image: the original and the best prediction so far, as the picture buffers hold them --- ///\n"
            "                const CPelBuf orgY  = bestCS->getOrgBuf().Y();\n"
            "                const CPelBuf predY = bestCS->getPredBuf().Y();\n"
            "                /// gather + absdiff + 1/1023 normalisation + network + argmax on the GPU; -1 on any failure (full RDO)\n"
            "                predictedSplitMode = m_cnnSplitPredictor->predictSplitMode(orgY.buf, orgY.stride, predY.buf, predY.stride, cuw, poc, cuQP);\n",
            "inference block")
    # the cut above ends inside the second catch block: "catch (...) {  std::cerr << "error\n";  }" -- remove its closing brace
    t = replace_once(t,
                     "predictedSplitMode = m_cnnSplitPredictor->predictSplitMode(orgY.buf, orgY.stride, predY.buf, predY.stride, cuw, poc, cuQP);\n                }\n",
                     "predictedSplitMode = m_cnnSplitPredictor->predictSplitMode(orgY.buf, orgY.stride, predY.buf, predY.stride, cuw, poc, cuQP);\n",
                     "catch-brace")
    return t


def patch_enccu_h(t):
    t = replace_once(t, "class EncCu\n", "namespace mlt { class SplitPredictor; }   // host/mlt_split_predictor.hpp (C ABI: include/mltcnn.h)\n\nclass EncCu\n", "fwd decl")
    t = replace_once(t, "  EncModeCtrl          *m_modeCtrl;\n", "  EncModeCtrl          *m_modeCtrl;\n  mlt::SplitPredictor  *m_cnnSplitPredictor = nullptr;   // one per EncCu (thread)\n", "member")
    return t


def patch_top_cmake(t):
    return cut(t, "# LibTorch", "include_directories(${OpenCV_INCLUDE_DIR})",
               "# MI355X-native MLT-CNN split predictor (C ABI, one shared library; no Torch / OpenCV)\n"
               "set(MLTCNN_ROOT \"\" CACHE PATH \"root of the mltcnn repository (include/, host/, fastintercu-vvc_amd/libmltcnn_hip.so)\")\n"
               "include_directories(${MLTCNN_ROOT}/include ${MLTCNN_ROOT}/host)\n"
               "link_directories(${MLTCNN_ROOT}/fastintercu-vvc_amd)\n", "top-level CMake")


def patch_lib_cmake(t):
    return cut(t, "# LibTorch", "target_link_libraries(EncoderLib ${OpenCV_LIBRARIES})",
               "# MI355X-native MLT-CNN split predictor\n"
               "target_link_libraries(EncoderLib mltcnn_hip)\n"
               "set_property(TARGET EncoderLib PROPERTY CXX_STANDARD 14)\n", "EncoderLib CMake")


def patch_encslice_cpp(t):
    # four OpenCV headers are included here but nothing of OpenCV is used in the file (found by LINKING the patched tree, round 3)
    return cut(t, "#include <opencv2/opencv.hpp>", "#include <opencv2/highgui.hpp>", "", "EncSlice.cpp OpenCV includes")


PATCHERS = {FILES[0]: patch_enccu_cpp, FILES[1]: patch_enccu_h, FILES[2]: patch_top_cmake, FILES[3]: patch_lib_cmake, FILES[4]: patch_encslice_cpp}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference/vtm-mlt-cpp")
    ap.add_argument("--out", default=os.path.join(ROOT, "patches", "vtm-mlt-cpp-mltcnn.patch"))
    args = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="vtmpatch_")
    try:
        for f in FILES:
            for side in ("a", "b"):
                os.makedirs(os.path.dirname(os.path.join(tmp, side, f)), exist_ok=True)
            shutil.copy(os.path.join(args.ref, f), os.path.join(tmp, "a", f))
            with open(os.path.join(args.ref, f), newline="") as fh:
                text = fh.read()
            crlf = "\r\n" in text
            if crlf:
                text = text.replace("\r\n", "\n")
            text = PATCHERS[f](text)
            if crlf:
                text = text.replace("\n", "\r\n")
            with open(os.path.join(tmp, "b", f), "w", newline="") as fh:
                fh.write(text)
        chunks = []
        for f in FILES:
            r = subprocess.run(["diff", "-u", "--label", "a/" + f, "--label", "b/" + f, os.path.join("a", f), os.path.join("b", f)],
                               cwd=tmp, capture_output=True)
            if r.returncode not in (0, 1):
                raise SystemExit(r.stderr.decode())
            chunks.append(r.stdout)
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "wb") as fh:
            fh.write(b"".join(chunks))
        print(f"wrote {args.out} ({sum(len(c) for c in chunks)} bytes)")
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
