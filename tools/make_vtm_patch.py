#!/usr/bin/env python3
"""Generate patches/vtm-mlt-cpp-mltcnn.patch: the SURVEY.md 8(f) N1 integration patch against the reference encoder
(/root/reference/vtm-mlt-cpp, the authors' VTM-11.0 tree).  Build container only (needs the reference tree); the patch
file it writes is what ships.

What the patch does (reference lines):
  source/Lib/EncoderLib/EncCu.cpp:60-65     drop <torch/script.h> / OpenCV includes, include host/mlt_split_predictor.hpp
  source/Lib/EncoderLib/EncCu.cpp:160-206   EncCu::destroy  -> delete the predictor (mlt_shutdown)
  source/Lib/EncoderLib/EncCu.cpp:233-259   EncCu::init     -> create the predictor ONCE (weights dir / device / size mask from
                                            MLTCNN_WEIGHTS_DIR / MLTCNN_DEVICE(S) / MLTCNN_SIZE_MASK / MLTCNN_FLAGS; the reference re-loads the .pt per CU, :894-900)
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
                     "        const char *flg  = std::getenv( \"MLTCNN_FLAGS\" );         // MLT_FLAG_* bits; default 0: the library's defaults (decision guard + flat guard on)\n"
                     "        const char *devs = std::getenv( \"MLTCNN_DEVICES\" );       // \"0,1,2,...\": ONE predictor over several GPUs (batched CUs are dealt round-robin)\n"
                     "        int devList[MLT_MAX_DEVICES], numDev = 0;\n"
                     "        for( const char *p = devs; p && *p && numDev < MLT_MAX_DEVICES; )\n"
                     "        {\n"
                     "            char *end = nullptr;\n"
                     "            devList[numDev++] = (int) std::strtol( p, &end, 10 );\n"
                     "            p = ( end && *end == ',' ) ? end + 1 : nullptr;\n"
                     "        }\n"
                     "        m_cnnSplitPredictor = new mlt::SplitPredictor( dir ? dir : \"./torch_model\", dev ? std::atoi( dev ) : 0,\n"
                     "                                                     mask ? (uint32_t) std::strtoul( mask, nullptr, 0 ) : MLT_SIZE_128,\n"
                     "                                                     flg ? (uint32_t) std::strtoul( flg, nullptr, 0 ) : 0u,\n"
                     "                                                     numDev ? devList : nullptr, numDev );\n"
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



# ----------------------------------------------------------------------------------------------------------------------------------
# SURVEY 8(f) N3, encoder half (round 4): an OPT-IN second patch on top of the N1 patch (--n3 -> patches/vtm-mlt-cpp-mltcnn-n3.patch).
# "Probe and replay" inside EncSlice::encodeCtus, for pictures coded with WaveFrontSynchro (entropy_coding_sync) and MLTCNN_BATCH=1:
#   for every WPP anti-diagonal (CTUs with equal x + 2y: mutually independent, INTEGRATION.md 7 rule 1)
#     1. PROBE each of its CTUs: compressCtu runs up to the CNN call site of the 128x128 CU (affine merge + merge / skip checks), where the
#        CU is SUBMITTED to the predictor (mlt_submit) instead of evaluated, and the CTU is abandoned -- nothing of it has reached the
#        picture-level coding structure, the mode-control stack is cleared, the CABAC estimator gets its start-of-CTU contexts back;
#     2. flush: the submitted CUs run as ONE batch on the GPU(s);
#     3. code the CTUs for real, in raster order within the diagonal: the call site now only collects its ticket (mlt_wait).
# The encoder keeps ONE running set of per-row states in raster order (CABAC contexts, HMVP table, previous QP, palette predictor); in
# diagonal order they are saved after every coded CTU and restored before the next CTU of that row.  The bitstream is the serial
# encoder's, bit for bit (tests/test_vtm_encoder.py) -- the decisions are the same, only when the GPU computes them changes.
# Round 5 (found with the reference's own random-access configuration on a 832 x 480 clip, tools/run_ra_eval.py): ONE more piece of encoder
# state is carried from CTU to CTU in CODING order -- InterSearch's two motion-estimation seed lists (m_affMVList / m_uniMvList, reset once
# per slice, EncSlice.cpp:1395-1396; the affine search extrapolates EVERY stored model to the current block wherever it lies,
# InterSearch.cpp:4673-4720).  In raster order a row inherits them from the END of the row above, which a diagonal schedule has not coded
# yet -- no schedule can reproduce that.  The same holds for the GLOBAL uni-MV reuse cache g_reusedUniMVs / g_isReusedUniMVsFilled (Rom.cpp:709-710;
# found with real, forced decisions at QP 27): keyed by the position INSIDE the CTU and cleared once per slice (EncSlice.cpp:1397), so a CU inherits
# motion vectors from the CU at the same in-CTU position of the previously CODED CTU (InterSearch.cpp:2455-2458 -> EncModeCtrl.cpp:1899-1906).
# So, when the variable MLTCNN_BATCH is DEFINED (0 or 1) and entropy-coding sync is on, the lists
# and the reuse cache restart at every CTU row (the idea behind VTM's own EnsureWppBitEqual switch): MLTCNN_BATCH=0 is the serial encoder under
# that rule, MLTCNN_BATCH=1 the diagonal schedule, which saves / restores them per row like the other row states -- bit-identical to each other;
# without the variable the encoder is the N1 encoder, untouched.
N3_FILES = ["source/Lib/EncoderLib/EncCu.cpp", "source/Lib/EncoderLib/EncCu.h", "source/Lib/EncoderLib/EncModeCtrl.h",
            "source/Lib/EncoderLib/EncSlice.cpp", "source/Lib/EncoderLib/InterSearch.h"]


def n3_enccu_cpp(t):
    t = replace_once(t,
                     "    xCompressCU(tempCS, bestCS, partitioner);\n    cs.slice->m_mapPltCost[0].clear();\n",
                     "    xCompressCU(tempCS, bestCS, partitioner);\n"
                     "    if( m_cnnProbe )\n"
                     "    {\n"
                     "        // probe pass (SURVEY 8f N3): the CTU was abandoned at the CNN call site (or before any other mode ran): nothing has been\n"
                     "        // written to the picture-level structure; the estimator gets its start-of-CTU contexts back\n"
                     "        m_CABACEstimator->getCtx() = m_CurrCtx->start;\n"
                     "        m_CurrCtx                  = 0;\n"
                     "        return;\n"
                     "    }\n"
                     "    cs.slice->m_mapPltCost[0].clear();\n", "compressCtu probe exit")
    t = replace_once(t,
                     "        EncTestMode currTestMode = m_modeCtrl->currTestMode();\n        currTestMode.maxCostAllowed = maxCostAllowed;\n",
                     "        EncTestMode currTestMode = m_modeCtrl->currTestMode();\n        currTestMode.maxCostAllowed = maxCostAllowed;\n"
                     "        if( m_cnnProbe && currTestMode.type != ETM_AFFINE && currTestMode.type != ETM_MERGE_SKIP )\n"
                     "        {\n"
                     "            m_modeCtrl->abortCTU();   // probe pass: only the two checks in front of the CNN call run; this CTU never reaches it\n"
                     "            return;\n"
                     "        }\n", "probe guard")
    t = replace_once(t,
                     "                predictedSplitMode = m_cnnSplitPredictor->predictSplitMode(orgY.buf, orgY.stride, predY.buf, predY.stride, cuw, poc, cuQP);\n",
                     "                const PreCalcValues &cnnPcv = *tempCS->pcv;\n"
                     "                const size_t ctuAddr = (size_t) ( cuy / cnnPcv.maxCUHeight ) * cnnPcv.widthInCtus + cux / cnnPcv.maxCUWidth;\n"
                     "                if( m_cnnProbe )\n"
                     "                {\n"
                     "                    // probe pass: stage this CU for the batch of its anti-diagonal and abandon the CTU\n"
                     "                    mlt_ticket tk = 0;\n"
                     "                    if( ctuAddr < m_cnnTicket.size() && m_cnnSplitPredictor->submitSplitMode(orgY.buf, orgY.stride, predY.buf, predY.stride, cuw, poc, cuQP, &tk) )\n"
                     "                        m_cnnTicket[ctuAddr] = (long long) tk + 1;\n"
                     "                    m_modeCtrl->abortCTU();\n"
                     "                    return;\n"
                     "                }\n"
                     "                if( ctuAddr < m_cnnTicket.size() && m_cnnTicket[ctuAddr] )\n"
                     "                {\n"
                     "                    // replay: the decision was computed with the rest of the diagonal's batch\n"
                     "                    predictedSplitMode = m_cnnSplitPredictor->waitSplitMode(cuw, (mlt_ticket) ( m_cnnTicket[ctuAddr] - 1 ));\n"
                     "                    m_cnnTicket[ctuAddr] = 0;\n"
                     "                }\n"
                     "                else\n"
                     "                predictedSplitMode = m_cnnSplitPredictor->predictSplitMode(orgY.buf, orgY.stride, predY.buf, predY.stride, cuw, poc, cuQP);\n",
                     "call site")
    t = replace_once(t, "void EncCu::compressCtu( CodingStructure& cs,",
                     "bool EncCu::cnnBatchingAvailable() const { return m_cnnSplitPredictor && m_cnnSplitPredictor->ok(); }\n"
                     "void EncCu::cnnFlush( int cuw ) { if( m_cnnSplitPredictor ) m_cnnSplitPredictor->flush( cuw ); }\n\n"
                     "void EncCu::compressCtu( CodingStructure& cs,", "helpers")
    return t


def n3_enccu_h(t):
    t = replace_once(t, "  mlt::SplitPredictor  *m_cnnSplitPredictor = nullptr;   // one per EncCu (thread)\n",
                     "  mlt::SplitPredictor  *m_cnnSplitPredictor = nullptr;   // one per EncCu (thread)\n"
                     "  // SURVEY 8f N3 (probe and replay, EncSlice::encodeCtus): probe mode abandons a CTU at the CNN call site after submitting its CU;\n"
                     "  // m_cnnTicket[ctu raster address] = ticket + 1 of a submitted CU (0: none)\n"
                     "  bool                  m_cnnProbe = false;\n"
                     "  std::vector<long long> m_cnnTicket;\n", "members")
    t = replace_once(t, "  void  compressCtu         ( CodingStructure& cs,",
                     "  void  setCnnProbe         ( bool on )                 { m_cnnProbe = on; }\n"
                     "  void  cnnResetTickets     ( size_t numCtus )          { m_cnnTicket.assign( numCtus, 0 ); }\n"
                     "  bool  cnnBatchingAvailable() const;\n"
                     "  void  cnnFlush            ( int cuw = 128 );\n"
                     "  void  compressCtu         ( CodingStructure& cs,", "methods")
    return t


def n3_encmodectrl_h(t):
    return replace_once(t, "    virtual void finishCULevel        ( Partitioner &partitioner )                                                            = 0;\n",
                        "    virtual void finishCULevel        ( Partitioner &partitioner )                                                            = 0;\n"
                        "    // SURVEY 8f N3 probe pass: drop the abandoned CTU's mode stack WITHOUT finishCULevel's book-keeping (which would cache the\n"
                        "    // abandoned CU's best mode for reuse and change the replay)\n"
                        "    void abortCTU() { m_ComprCUCtxList.clear(); }\n", "abortCTU")


def n3_intersearch_h(t):
    return replace_once(t, "  void resetUniMvList() { m_uniMvListIdx = 0; m_uniMvListSize = 0; }\n",
                        "  void resetUniMvList() { m_uniMvListIdx = 0; m_uniMvListSize = 0; }\n"
                        "  // SURVEY 8f N3 (probe and replay over WPP anti-diagonals): the two motion-estimation seed lists above are carried from CTU to CTU in\n"
                        "  // coding order (the affine one is consulted whatever the position of its entries); a diagonal schedule saves / restores them per CTU row\n"
                        "  struct MeSeedLists { std::vector<AffineMVInfo> aff; std::vector<BlkUniMvInfo> uni; int affIdx = 0, affSize = 0, uniIdx = 0, uniSize = 0; };\n"
                        "  void saveMeSeedLists( MeSeedLists &s ) const\n"
                        "  {\n"
                        "    s.aff.assign( m_affMVList, m_affMVList + m_affMVListMaxSize ); s.affIdx = m_affMVListIdx; s.affSize = m_affMVListSize;\n"
                        "    s.uni.assign( m_uniMvList, m_uniMvList + m_uniMvListMaxSize ); s.uniIdx = m_uniMvListIdx; s.uniSize = m_uniMvListSize;\n"
                        "  }\n"
                        "  void restoreMeSeedLists( const MeSeedLists &s )\n"
                        "  {\n"
                        "    if( (int) s.aff.size() != m_affMVListMaxSize || (int) s.uni.size() != m_uniMvListMaxSize ) { resetAffineMVList(); resetUniMvList(); return; }\n"
                        "    std::copy( s.aff.begin(), s.aff.end(), m_affMVList ); m_affMVListIdx = s.affIdx; m_affMVListSize = s.affSize;\n"
                        "    std::copy( s.uni.begin(), s.uni.end(), m_uniMvList ); m_uniMvListIdx = s.uniIdx; m_uniMvListSize = s.uniSize;\n"
                        "  }\n", "ME seed lists")


def n3_encslice_cpp(t):
    t = replace_once(t,
                     "  // for every CTU in the slice\n  for( uint32_t ctuIdx = 0; ctuIdx < pcSlice->getNumCtuInSlice(); ctuIdx++ )\n  {\n"
                     "    const int32_t ctuRsAddr = pcSlice->getCtuAddrInSlice( ctuIdx );\n",
                     "  // SURVEY 8f N3, encoder half (opt-in: MLTCNN_BATCH=1): probe-and-replay over WPP anti-diagonals.  Needs entropy-coding sync (every CTU row\n"
                     "  // starts from the state after the first CTU of the row above, so the CTUs with equal x + 2y are mutually independent), one tile, one\n"
                     "  // slice per picture, no per-CTU lambda adaptation; anything else keeps the raster loop.  sched: (ctuIdx, phase) with phase 0 = code,\n"
                     "  // 1 = probe (submit the 128x128 CU's planes to the CNN, abandon the CTU), 2 = flush the submitted CUs as one batch.\n"
                     "  const uint32_t heightInCtus = pcv.heightInCtus;\n"
                     "  static const bool cnnBatchEnv = std::getenv( \"MLTCNN_BATCH\" ) && std::atoi( std::getenv( \"MLTCNN_BATCH\" ) ) != 0;\n"
                     "  // MLTCNN_BATCH defined (0 or 1) + entropy-coding sync: InterSearch's motion-estimation seed lists (reset once per slice upstream,\n"
                     "  // EncSlice.cpp:1395-1396, and carried across CTU rows in raster order) restart at every CTU row, so that a row's encode depends only on\n"
                     "  // the rows above it.  MLTCNN_BATCH=0 is the serial encoder under this rule -- the leg the diagonal schedule (MLTCNN_BATCH=1) is\n"
                     "  // bit-identical to; without the variable nothing changes.\n"
                     "  static const bool cnnRowRule = std::getenv( \"MLTCNN_BATCH\" ) != nullptr;\n"
                     "  const bool cnnRowReset = cnnRowRule && pEncLib->getEntropyCodingSyncEnabledFlag();\n"
                     "  const bool cnnBatch = cnnBatchEnv && pEncLib->getEntropyCodingSyncEnabledFlag() && !pcSlice->isIntra() && cs.pps->getNumTiles() == 1\n"
                     "                        && pcSlice->getNumCtuInSlice() == widthInCtus * heightInCtus && pcSlice->getCtuAddrInSlice( 0 ) == 0\n"
                     "                        && !pCfg->getUseRateCtrl() && !pCfg->getUsePerceptQPA() && m_pcCuEncoder->cnnBatchingAvailable();\n"
                     "  std::vector<std::pair<uint32_t, int>> sched;\n"
                     "  if( cnnBatch )\n"
                     "  {\n"
                     "    m_pcCuEncoder->cnnResetTickets( widthInCtus * heightInCtus );\n"
                     "    for( uint32_t d = 0; d < widthInCtus + 2 * heightInCtus; d++ )\n"
                     "    {\n"
                     "      std::vector<uint32_t> diag;\n"
                     "      for( uint32_t y = 0; y < heightInCtus && 2 * y <= d; y++ )\n"
                     "        if( d - 2 * y < widthInCtus ) diag.push_back( y * widthInCtus + ( d - 2 * y ) );\n"
                     "      int probed = 0;\n"
                     "      for( uint32_t a : diag )   // only CTUs whose 128x128 CU lies inside the picture reach the CNN (the gate, EncCu.cpp:746-756)\n"
                     "        if( ( a % widthInCtus + 1 ) * pcv.maxCUWidth <= pcv.lumaWidth && ( a / widthInCtus + 1 ) * pcv.maxCUHeight <= pcv.lumaHeight ) { sched.push_back( { a, 1 } ); probed++; }\n"
                     "      if( probed ) sched.push_back( { 0, 2 } );\n"
                     "      for( uint32_t a : diag ) sched.push_back( { a, 0 } );\n"
                     "      if( probed )\n"
                     "        if( const char *lg = std::getenv( \"MLTCNN_BATCH_LOG\" ) )\n"
                     "          if( FILE *f = std::fopen( lg, \"a\" ) ) { std::fprintf( f, \"poc %d diagonal %u ctus %d probed %d\\n\", pcSlice->getPOC(), d, (int) diag.size(), probed ); std::fclose( f ); }\n"
                     "    }\n"
                     "  }\n"
                     "  else\n"
                     "    for( uint32_t i = 0; i < pcSlice->getNumCtuInSlice(); i++ ) sched.push_back( { i, 0 } );\n"
                     "  // per-row encoder state that the raster loop carries implicitly from a CTU to its right-hand neighbour\n"
                     "  std::vector<Ctx>           rowCtx( cnnBatch ? heightInCtus : 0 );\n"
                     "  std::vector<LutMotionCand> rowLut( cnnBatch ? heightInCtus : 0 );\n"
                     "  std::vector<PLTBuf>        rowPLT( cnnBatch ? heightInCtus : 0 );\n"
                     "  std::vector<int>           rowPrevQP( cnnBatch ? 2 * heightInCtus : 0 );\n"
                     "  std::vector<InterSearch::MeSeedLists> rowMe( cnnBatch ? heightInCtus : 0 );\n"
                     "  InterSearch *cnnInterSearch = pEncLib->getInterSearch();\n"
                     "  // ... and the GLOBAL uni-MV reuse cache (Rom.cpp:709-710; written by InterSearch.cpp:2455-2458, read by EncModeCtrl.cpp:1899-1906): keyed by the\n"
                     "  // position INSIDE the CTU and cleared once per slice (EncSlice.cpp:1397) -- a CU inherits the motion vectors the CU at the same in-CTU\n"
                     "  // position of the PREVIOUSLY CODED CTU left behind.  Saved sparsely (the filled entries only) per CTU row.\n"
                     "  struct CnnReusedMvs { std::vector<uint32_t> idx; std::vector<Mv> mvs; };\n"
                     "  std::vector<CnnReusedMvs> rowReused( cnnBatch ? heightInCtus : 0 );\n"
                     "  constexpr size_t cnnReuseSlots = sizeof( g_isReusedUniMVsFilled ) / sizeof( bool ), cnnReuseMvs = 2 * 33;\n"
                     "  auto cnnSaveReused = [&]( CnnReusedMvs &s ) {\n"
                     "    const bool *fl = &g_isReusedUniMVsFilled[0][0][0][0]; const Mv *base = &g_reusedUniMVs[0][0][0][0][0][0];\n"
                     "    s.idx.clear(); s.mvs.clear();\n"
                     "    for( size_t i = 0; i < cnnReuseSlots; i++ ) if( fl[i] ) { s.idx.push_back( (uint32_t) i ); s.mvs.insert( s.mvs.end(), base + i * cnnReuseMvs, base + ( i + 1 ) * cnnReuseMvs ); }\n"
                     "  };\n"
                     "  auto cnnRestoreReused = [&]( const CnnReusedMvs &s ) {\n"
                     "    bool *fl = &g_isReusedUniMVsFilled[0][0][0][0]; Mv *base = &g_reusedUniMVs[0][0][0][0][0][0];\n"
                     "    ::memset( g_isReusedUniMVsFilled, 0, sizeof( g_isReusedUniMVsFilled ) );\n"
                     "    for( size_t k = 0; k < s.idx.size(); k++ ) { fl[s.idx[k]] = true; std::copy( s.mvs.begin() + k * cnnReuseMvs, s.mvs.begin() + ( k + 1 ) * cnnReuseMvs, base + (size_t) s.idx[k] * cnnReuseMvs ); }\n"
                     "  };\n"
                     "\n"
                     "  // for every CTU in the slice\n  for( size_t schedIdx = 0; schedIdx < sched.size(); schedIdx++ )\n  {\n"
                     "    const uint32_t ctuIdx   = sched[schedIdx].first;\n"
                     "    const int      cnnPhase = sched[schedIdx].second;\n"
                     "    if( cnnPhase == 2 ) { m_pcCuEncoder->cnnFlush(); continue; }\n"
                     "    const int32_t ctuRsAddr = pcSlice->getCtuAddrInSlice( ctuIdx );\n", "loop head")
    t = replace_once(t,
                     "    DTRACE_UPDATE( g_trace_ctx, std::make_pair( \"ctu\", ctuRsAddr ) );\n\n    if( pCfg->getSwitchPOC() != pcPic->poc || -1 == pCfg->getDebugCTU() )\n",
                     "    DTRACE_UPDATE( g_trace_ctx, std::make_pair( \"ctu\", ctuRsAddr ) );\n\n"
                     "    if( cnnBatch && ctuXPosInCtus > 0 )\n"
                     "    {\n"
                     "      // diagonal order: what CTU (x - 1, y) left behind (the first CTU of a row is initialised below exactly as in raster order)\n"
                     "      pCABACWriter->getCtx() = rowCtx[ctuYPosInCtus];\n"
                     "      cs.motionLut           = rowLut[ctuYPosInCtus];\n"
                     "      cs.setPrevPLT( rowPLT[ctuYPosInCtus] );\n"
                     "      prevQP[0] = rowPrevQP[2 * ctuYPosInCtus]; prevQP[1] = rowPrevQP[2 * ctuYPosInCtus + 1];\n"
                     "      cnnInterSearch->restoreMeSeedLists( rowMe[ctuYPosInCtus] );\n"
                     "      cnnRestoreReused( rowReused[ctuYPosInCtus] );\n"
                     "    }\n"
                     "    if( cnnRowReset && ctuXPosInCtus == 0 ) { cnnInterSearch->resetAffineMVList(); cnnInterSearch->resetUniMvList(); ::memset( g_isReusedUniMVsFilled, 0, sizeof( g_isReusedUniMVsFilled ) ); }\n\n"
                     "    if( pCfg->getSwitchPOC() != pcPic->poc || -1 == pCfg->getDebugCTU() )\n", "row state restore")
    t = replace_once(t,
                     "  if (pCfg->getSwitchPOC() != pcPic->poc || ctuRsAddr >= pCfg->getDebugCTU())\n    m_pcCuEncoder->compressCtu( cs, ctuArea, ctuRsAddr, prevQP, currQP );\n",
                     "    if( cnnPhase == 1 )\n"
                     "    {\n"
                     "      m_pcCuEncoder->setCnnProbe( true );\n"
                     "      m_pcCuEncoder->compressCtu( cs, ctuArea, ctuRsAddr, prevQP, currQP );   // runs up to the CNN call site, submits, abandons the CTU\n"
                     "      m_pcCuEncoder->setCnnProbe( false );\n"
                     "      continue;\n"
                     "    }\n"
                     "  if (pCfg->getSwitchPOC() != pcPic->poc || ctuRsAddr >= pCfg->getDebugCTU())\n    m_pcCuEncoder->compressCtu( cs, ctuArea, ctuRsAddr, prevQP, currQP );\n",
                     "probe call")
    t = replace_once(t,
                     "    m_uiPicTotalBits += actualBits;\n    m_uiPicDist       = cs.dist;\n    // for last Ctu in the slice\n",
                     "    m_uiPicTotalBits += actualBits;\n    m_uiPicDist       = cs.dist;\n"
                     "    if( cnnBatch )\n"
                     "    {\n"
                     "      rowCtx[ctuYPosInCtus] = pCABACWriter->getCtx();\n"
                     "      rowLut[ctuYPosInCtus] = cs.motionLut;\n"
                     "      cs.storePrevPLT( rowPLT[ctuYPosInCtus] );\n"
                     "      rowPrevQP[2 * ctuYPosInCtus] = prevQP[0]; rowPrevQP[2 * ctuYPosInCtus + 1] = prevQP[1];\n"
                     "      cnnInterSearch->saveMeSeedLists( rowMe[ctuYPosInCtus] );\n"
                     "      cnnSaveReused( rowReused[ctuYPosInCtus] );\n"
                     "    }\n"
                     "    // for last Ctu in the slice\n", "row state save")
    return t


N3_PATCHERS = {N3_FILES[0]: n3_enccu_cpp, N3_FILES[1]: n3_enccu_h, N3_FILES[2]: n3_encmodectrl_h, N3_FILES[3]: n3_encslice_cpp, N3_FILES[4]: n3_intersearch_h}

PATCHERS = {FILES[0]: patch_enccu_cpp, FILES[1]: patch_enccu_h, FILES[2]: patch_top_cmake, FILES[3]: patch_lib_cmake, FILES[4]: patch_encslice_cpp}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference/vtm-mlt-cpp")
    ap.add_argument("--out", default=None)
    ap.add_argument("--n3", action="store_true", help="write the opt-in encoder-side batching patch (applies ON TOP of the N1 patch)")
    args = ap.parse_args()
    if args.out is None:
        args.out = os.path.join(ROOT, "patches", "vtm-mlt-cpp-mltcnn-n3.patch" if args.n3 else "vtm-mlt-cpp-mltcnn.patch")
    files = N3_FILES if args.n3 else FILES
    tmp = tempfile.mkdtemp(prefix="vtmpatch_")
    try:
        for f in files:
            for side in ("a", "b"):
                os.makedirs(os.path.dirname(os.path.join(tmp, side, f)), exist_ok=True)
            with open(os.path.join(args.ref, f), newline="") as fh:
                text = fh.read()
            crlf = "\r\n" in text
            if crlf:
                text = text.replace("\r\n", "\n")
            if args.n3:  # side a = the N1-patched file (or the reference's own where N1 does not touch it), side b = N1 + N3
                base = PATCHERS[f](text) if f in PATCHERS else text
                text = N3_PATCHERS[f](base)
                with open(os.path.join(tmp, "a", f), "w", newline="") as fh:
                    fh.write(base.replace("\n", "\r\n") if crlf else base)
            else:
                shutil.copy(os.path.join(args.ref, f), os.path.join(tmp, "a", f))
                text = PATCHERS[f](text)
            if crlf:
                text = text.replace("\n", "\r\n")
            with open(os.path.join(tmp, "b", f), "w", newline="") as fh:
                fh.write(text)
        chunks = []
        for f in files:
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
