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
#pragma once
#include <cstdint>
#include <cstdio>
#include <string>

#include "../include/mltcnn.h"

namespace mlt {

using Pel = int16_t;  // CommonLib/TypeDef.h:277 (RExt__HIGH_BIT_DEPTH_SUPPORT off)

class SplitPredictor {
 public:
  // weightsDir replaces the hard-coded "/home/ubuntu/whyeo/vtm-mlt-final/torch_model" (EncCu.cpp:899); files are
  // MLTORPQ_splitMode_<S>.mltw (tools/convert_weights.py).  sizeMask: MLT_SIZE_* bits (reference: 128 only, :754).
  explicit SplitPredictor(const std::string &weightsDir, int device = 0, uint32_t sizeMask = MLT_SIZE_128, uint32_t flags = 0) {
    mlt_config cfg{};
    cfg.struct_size = sizeof cfg;
    cfg.device = device;
    cfg.weights_dir = weightsDir.c_str();
    cfg.size_mask = sizeMask;
    for (int &h : cfg.head_index) h = -1;  // reference defaults: element [2] for 128, [0] otherwise (EncCu.cpp:913-919)
    cfg.max_batch = 1;
    cfg.flags = flags;
    cfg.guard_margin = 0.f;  // default threshold when MLT_FLAG_DECISION_GUARD is set
    cfg.tolerance = 0.f;     // default |dlogit| contract (1e-3) for the load-time calibration of the fast arithmetic
    m_mask = sizeMask ? sizeMask : MLT_SIZE_128;
    const int rc = mlt_init(&cfg, &m_ctx);
    if (rc != MLT_OK) {
      std::fprintf(stderr, "error loading the model\n");  // the reference's message (EncCu.cpp:904)
      std::fprintf(stderr, "  mltcnn: %s\n", mlt_last_error(nullptr));
      m_ctx = nullptr;
    }
  }
  ~SplitPredictor() { mlt_shutdown(m_ctx); }
  SplitPredictor(const SplitPredictor &) = delete;
  SplitPredictor &operator=(const SplitPredictor &) = delete;

  bool ok() const { return m_ctx != nullptr; }

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
    if (!m_ctx) return -1;
    if (mlt_predict(m_ctx, org, orgStride, pred, predStride, cuw, poc, cuQP, &split, logitsOpt) != MLT_OK) {
      std::fprintf(stderr, "error\n");  // EncCu.cpp:925
      return -1;
    }
    return split;
  }

  // Encoder-side batching: stage the CU now, decide later.  submitSplitMode() returns a ticket (or
  // false on failure: treat like predictedSplitMode = -1); waitSplitMode() returns the split mode of that ticket and runs
  // every CU staged so far for this size as ONE batch if that has not happened yet.  With k CUs whose setNewModeList
  // call can be postponed together (e.g. the CTUs of a WPP anti-diagonal) the per-CU cost falls from ~150 us to
  // ~150 us / k (+ ~8 us).  At most MLT_DEFER_CAP CUs per batch; a full batch is launched by the next submit.
  bool submitSplitMode(const Pel *org, int orgStride, const Pel *pred, int predStride, int cuw, int poc, int cuQP, mlt_ticket *ticket) {
    return m_ctx && mlt_submit(m_ctx, org, orgStride, pred, predStride, cuw, poc, cuQP, ticket) == MLT_OK;
  }
  void flush(int cuw) { if (m_ctx) (void)mlt_flush(m_ctx, cuw); }  // start the batch early, e.g. before unrelated host work
  int waitSplitMode(int cuw, mlt_ticket ticket, float *logitsOpt = nullptr) {
    int32_t split = -1;
    if (!m_ctx || mlt_wait(m_ctx, cuw, ticket, &split, logitsOpt) != MLT_OK) return -1;
    return split;
  }

 private:
  mlt_ctx *m_ctx = nullptr;
  uint32_t m_mask = MLT_SIZE_128;
};

}  // namespace mlt
