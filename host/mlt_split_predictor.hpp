// mlt_split_predictor.hpp -- C++ host-side mirror of the reference call site, above the C ABI (include/mltcnn.h).
//
// The reference has no operator/plugin interface for this path: the CNN is an inline block of
// EncCu::xCompressCU (vtm-mlt-cpp/source/Lib/EncoderLib/EncCu.cpp:799-930).  This class gives that block a name
// and keeps its argument meaning and error behaviour:
//   gate()              == the useCNN condition                       EncCu.cpp:746-756
//   predictSplitMode()  == gather + absdiff + normalise + forward + argmax   EncCu.cpp:806-921
//   failure             -> returns -1, exactly what the reference leaves in predictedSplitMode when
//                          torch throws (EncCu.cpp:902-905,923-926); EncModeCtrl::setNewModeList(…,-1,…) is then a
//                          no-op (EncModeCtrl.cpp:147-148) and the encoder runs its exhaustive RDO.
// Header-only; link with -lmltcnn_hip.  One instance per EncCu (the encoder is single-threaded, or one EncCu per
// thread under WPP / split parallelism, EncCu.cpp:233).
//
// Decisions: the split mode is what the encoder consumes (EncCu.cpp:921 -> EncModeCtrl.cpp:110-149), so the library's DECISION GUARD is
// on by default (ABI 4; margin 3 x tolerance; MLT_FLAG_NO_DECISION_GUARD turns it off): a CU whose decision-head top-2 margin is too
// small for the fast arithmetic's calibrated error to leave the argmax alone is re-evaluated with the exact arithmetic before the call returns.
//
// MLTCNN_STATS=1 (any build): the destructor prints ONE line to stderr -- calls and wall-clock seconds inside predictSplitMode / submit /
// flush / wait, calls per CU size, seconds inside mlt_init (weights + calibration) -- the "share of an encode spent inside the predictor" figure
// tools/eval_harness.py reads (N4).
//
// Test hooks, compiled in only with -DMLTCNN_TEST_HOOKS (tools/build_vtm.sh does; a production build carries none of them):
//   MLTCNN_FAULT_INJECT=1      the predictor reports ok() without touching a device and every predictSplitMode() fails (-1):
//                              exercises the reference's swallow-and-continue contract from the real call site on a box without a GPU
//   MLTCNN_FORCE_SPLIT=k       (with MLTCNN_FAULT_INJECT=1) every prediction "succeeds" with split mode k instead of failing: real, decision-
//                              dependent encoder paths (EncModeCtrl::setNewModeList, EncModeCtrl.cpp:110-149) on a box without a GPU -- how
//                              the probe-and-replay schedule is debugged against the serial encoder in the build container
//   MLTCNN_CALL_DUMP_FILE=path every predictSplitMode() call is appended to `path` (little-endian records, see dumpCall; one
//                              write(2) per record on an O_APPEND descriptor, so instances on several threads cannot interleave):
//                              tests/test_vtm_encoder.py re-checks each one against the CPU oracle
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#ifdef MLTCNN_TEST_HOOKS
#include <fcntl.h>
#include <unistd.h>
#endif

#include "../include/mltcnn.h"

namespace mlt {

using Pel = int16_t;  // CommonLib/TypeDef.h:277 (RExt__HIGH_BIT_DEPTH_SUPPORT off)

class SplitPredictor {
 public:
  // weightsDir replaces the hard-coded "/home/ubuntu/whyeo/vtm-mlt-final/torch_model" (EncCu.cpp:899); files are
  // MLTORPQ_splitMode_<S>.mltw (tools/convert_weights.py).  sizeMask: MLT_SIZE_* bits (reference: 128 only, :754).
  // flags: MLT_FLAG_* bits; 0 = the library's defaults (decision guard and flat guard on, see above).  devices / nDevices: one predictor serving several GPUs
  // (mlt_config.devices: batches submitted through submitSplitMode() are dealt round-robin); nullptr: `device`.
  explicit SplitPredictor(const std::string &weightsDir, int device = 0, uint32_t sizeMask = MLT_SIZE_128, uint32_t flags = 0,
                          const int *devices = nullptr, int nDevices = 0) {
    mlt_config cfg{};
    cfg.struct_size = sizeof cfg;
    cfg.device = device;
    for (int i = 0; i < nDevices && i < MLT_MAX_DEVICES && devices; ++i) { cfg.devices[i] = devices[i]; cfg.n_devices = i + 1; }
    cfg.weights_dir = weightsDir.c_str();
    cfg.size_mask = sizeMask;
    for (int &h : cfg.head_index) h = -1;  // reference defaults: element [2] for 128, [0] otherwise (EncCu.cpp:913-919)
    cfg.max_batch = 1;
    cfg.flags = flags;
    cfg.guard_margin = 0.f;  // the decision guard's default threshold (3 x tolerance)
    cfg.tolerance = 0.f;     // default |dlogit| contract (1e-3) for the load-time calibration of the fast arithmetic
    m_mask = sizeMask ? sizeMask : MLT_SIZE_128;
    if (const char *s = std::getenv("MLTCNN_STATS")) m_stats = std::atoi(s) != 0;
#ifdef MLTCNN_TEST_HOOKS
    if (const char *d = std::getenv("MLTCNN_CALL_DUMP_FILE")) m_dumpPath = d;
    if (const char *f = std::getenv("MLTCNN_FAULT_INJECT")) m_faultInject = std::atoi(f) != 0;
    if (const char *f = std::getenv("MLTCNN_FORCE_SPLIT")) m_forceSplit = std::atoi(f);
    if (m_faultInject) return;
#endif
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = mlt_init(&cfg, &m_ctx);   // loads, folds, packs and CALIBRATES the weights of every enabled size (once per encoder process)
    m_initSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (rc != MLT_OK) {
      std::fprintf(stderr, "error loading the model\n");  // the reference's message (EncCu.cpp:904)
      std::fprintf(stderr, "  mltcnn: %s\n", mlt_last_error(nullptr));
      m_ctx = nullptr;
    }
  }
  ~SplitPredictor() {
    if (m_stats)
      std::fprintf(stderr, "mltcnn-stats predict_calls=%llu predict_s=%.6f submit_calls=%llu submit_s=%.6f wait_calls=%llu wait_s=%.6f flush_calls=%llu flush_s=%.6f "
                           "calls_128=%llu calls_64=%llu calls_32=%llu calls_16=%llu failed=%llu init_s=%.6f\n",
                   m_n[0], m_t[0], m_n[1], m_t[1], m_n[2], m_t[2], m_n[3], m_t[3], m_bySize[0], m_bySize[1], m_bySize[2], m_bySize[3], m_failed, m_initSeconds);
    mlt_shutdown(m_ctx);
  }
  SplitPredictor(const SplitPredictor &) = delete;
  SplitPredictor &operator=(const SplitPredictor &) = delete;

  bool ok() const { return m_ctx != nullptr || m_faultInject; }

  // EncCu.cpp:746-756.  chType: partitioner.chType (0 = luma / joint tree); isIntraSlice: slice type == I_SLICE;
  // (cux, cuy, cuw, cuh): tempCS->area.Y(); (picW, picH): slice->getPic()->Y().
  bool gate(int chType, bool isIntraSlice, int cux, int cuy, int cuw, int cuh, int picW, int picH) const {
    if (chType != 0 || isIntraSlice || cuw != cuh) return false;
    const uint32_t bit = cuw == 128 ? MLT_SIZE_128 : cuw == 64 ? MLT_SIZE_64 : cuw == 32 ? MLT_SIZE_32 : cuw == 16 ? MLT_SIZE_16 : 0u;
    if (!(bit & m_mask)) return false;
    return cux + cuw <= picW && cuy + cuh <= picH;
  }

  // EncCu.cpp:806-921.  org: bestCS->getOrgBuf().Y() {buf, stride}; pred: bestCS->getPredBuf().Y() {buf, stride};
  // poc: bestCS->slice->getPOC(); cuQP: currTestMode.qp.  Returns predictedSplitMode (0 none, 1 QT, 2 BT_H, 3 BT_V, ...)
  // or -1 on any failure.
  int predictSplitMode(const Pel *org, int orgStride, const Pel *pred, int predStride, int cuw, int poc, int cuQP, float *logitsOpt = nullptr) {
    int32_t split = -1;
    float lg[MLT_MAX_LOGITS] = {0};
    Timer tm(this, 0, cuw);
    if (m_faultInject && m_forceSplit >= 0) return m_forceSplit;
    if (!m_ctx || mlt_predict(m_ctx, org, orgStride, pred, predStride, cuw, poc, cuQP, &split, (logitsOpt || !m_dumpPath.empty()) ? lg : nullptr) != MLT_OK) {
      std::fprintf(stderr, "error\n");  // EncCu.cpp:925
      split = -1;
      ++m_failed;
    }
    if (logitsOpt) for (int i = 0; i < mlt_num_logits(cuw); ++i) logitsOpt[i] = lg[i];
    if (!m_dumpPath.empty()) dumpCall(org, orgStride, pred, predStride, cuw, poc, cuQP, split, lg);
    return split;
  }

  // Encoder-side batching: stage the CU now, decide later.  submitSplitMode() returns a ticket (or
  // false on failure: treat like predictedSplitMode = -1); waitSplitMode() returns the split mode of that ticket and runs
  // every CU staged so far for this size as ONE batch if that has not happened yet.  With k CUs whose setNewModeList
  // call can be postponed together (e.g. the CTUs of a WPP anti-diagonal) the per-CU cost falls from ~150 us to
  // ~150 us / k (+ ~8 us).  At most MLT_DEFER_CAP CUs per batch; a full batch is launched by the next submit.
  bool submitSplitMode(const Pel *org, int orgStride, const Pel *pred, int predStride, int cuw, int poc, int cuQP, mlt_ticket *ticket) {
    Timer tm(this, 1, cuw);
    if (m_faultInject && m_forceSplit >= 0) { *ticket = 0; return true; }
    return m_ctx && mlt_submit(m_ctx, org, orgStride, pred, predStride, cuw, poc, cuQP, ticket) == MLT_OK;
  }
  void flush(int cuw) { Timer tm(this, 3, 0); if (m_ctx) (void)mlt_flush(m_ctx, cuw); }  // start the batch early, e.g. before unrelated host work
  int waitSplitMode(int cuw, mlt_ticket ticket, float *logitsOpt = nullptr) {
    int32_t split = -1;
    Timer tm(this, 2, 0);
    if (m_faultInject && m_forceSplit >= 0) return m_forceSplit;
    if (!m_ctx || mlt_wait(m_ctx, cuw, ticket, &split, logitsOpt) != MLT_OK) return -1;
    return split;
  }

 private:
  // one record per call: int32 {magic 0x4D4C5443, cuw, poc, qp, split, nLogits}, float logits[MLT_MAX_LOGITS], int16 org[cuw*cuw], int16 pred[cuw*cuw]
  // -- built in a buffer and written with ONE write(2) on an O_APPEND descriptor (atomic with respect to other appenders: predictors
  // of several EncCu threads share one dump file); a failed write is reported once.
  void dumpCall(const Pel *org, int orgStride, const Pel *pred, int predStride, int cuw, int poc, int cuQP, int split, const float *lg) const {
#ifdef MLTCNN_TEST_HOOKS
    const int32_t hdr[6] = {0x4D4C5443, cuw, poc, cuQP, split, mlt_num_logits(cuw)};
    std::vector<char> rec(sizeof hdr + sizeof(float) * MLT_MAX_LOGITS + 2 * sizeof(Pel) * (size_t)cuw * cuw);
    char *w = rec.data();
    std::memcpy(w, hdr, sizeof hdr); w += sizeof hdr;
    std::memcpy(w, lg, sizeof(float) * MLT_MAX_LOGITS); w += sizeof(float) * MLT_MAX_LOGITS;
    for (int pl = 0; pl < 2; ++pl)
      for (int y = 0; y < cuw; ++y) {
        const Pel *src = (pl ? pred : org) + (ptrdiff_t)y * (pl ? predStride : orgStride);
        std::memcpy(w, src, sizeof(Pel) * (size_t)cuw); w += sizeof(Pel) * (size_t)cuw;
      }
    const int fd = ::open(m_dumpPath.c_str(), O_WRONLY | O_CREAT | O_APPEND, 0644);
    const bool ok = fd >= 0 && ::write(fd, rec.data(), rec.size()) == (ssize_t)rec.size();
    if (fd >= 0) ::close(fd);
    if (!ok && !m_dumpFailed) { m_dumpFailed = true; std::fprintf(stderr, "mltcnn: cannot append to %s\n", m_dumpPath.c_str()); }
#else
    (void)org; (void)orgStride; (void)pred; (void)predStride; (void)cuw; (void)poc; (void)cuQP; (void)split; (void)lg;
#endif
  }

  // MLTCNN_STATS: wall-clock inside the predictor, per entry point (0 predict, 1 submit, 2 wait, 3 flush)
  struct Timer {
    SplitPredictor *p; int k; std::chrono::steady_clock::time_point t0;
    Timer(SplitPredictor *pp, int kk, int cuw) : p(pp->m_stats ? pp : nullptr), k(kk) {
      if (!p) return;
      t0 = std::chrono::steady_clock::now();
      if (cuw) ++p->m_bySize[cuw == 128 ? 0 : cuw == 64 ? 1 : cuw == 32 ? 2 : 3];
    }
    ~Timer() { if (p) { p->m_t[k] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); ++p->m_n[k]; } }
  };
  bool m_stats = false;
  unsigned long long m_n[4] = {0, 0, 0, 0}, m_bySize[4] = {0, 0, 0, 0}, m_failed = 0;
  double m_t[4] = {0, 0, 0, 0};
  double m_initSeconds = 0.0;   // wall-clock of mlt_init (weights + load-time calibration): what an encoder process pays once

  mlt_ctx *m_ctx = nullptr;
  uint32_t m_mask = MLT_SIZE_128;
  bool m_faultInject = false;   // (only ever set with MLTCNN_TEST_HOOKS)
  int m_forceSplit = -1;        // (only ever set with MLTCNN_TEST_HOOKS)
  mutable bool m_dumpFailed = false;
  std::string m_dumpPath;
};

}  // namespace mlt
