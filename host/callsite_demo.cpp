// callsite_demo.cpp -- stand-alone exercise of host/mlt_split_predictor.hpp the way EncCu::xCompressCU would use it.
// Builds with g++ (no HIP headers needed); with a GPU it runs one synthetic 128x128 CU out of a 1920-wide picture
// buffer.  usage: callsite_demo <weights_dir>      (expects <weights_dir>/MLTORPQ_splitMode_128.mltw)
#include <cstdlib>
#include <vector>

#include "mlt_split_predictor.hpp"

int main(int argc, char **argv) {
  if (argc < 2) {
    std::printf("abi %d, logits(128) %d, logits(32) %d\n", mlt_abi_version(), mlt_num_logits(128), mlt_num_logits(32));
    return 0;  // link / export check only
  }
  mlt::SplitPredictor cnn(argv[1]);
  if (!cnn.ok()) return 2;
  const int picW = 1920, picH = 1080, cux = 256, cuy = 128, cuw = 128;
  std::vector<mlt::Pel> picture((size_t)picW * picH), predBuf((size_t)cuw * cuw);
  uint32_t lcg = 12345u;
  for (auto &v : picture) { lcg = lcg * 1664525u + 1013904223u; v = (mlt::Pel)((lcg >> 22) & 1023); }
  for (int y = 0; y < cuw; ++y)
    for (int x = 0; x < cuw; ++x) {
      lcg = lcg * 1664525u + 1013904223u;
      int p = picture[(size_t)(cuy + y) * picW + cux + x] + (int)((lcg >> 24) % 41) - 20;
      predBuf[(size_t)y * cuw + x] = (mlt::Pel)(p < 0 ? 0 : p > 1023 ? 1023 : p);
    }
  if (!cnn.gate(/*chType*/ 0, /*isIntraSlice*/ false, cux, cuy, cuw, cuw, picW, picH)) return 3;
  float logits[MLT_MAX_LOGITS] = {0};
  const int predictedSplitMode =
      cnn.predictSplitMode(&picture[(size_t)cuy * picW + cux], picW, predBuf.data(), cuw, cuw, /*poc*/ 8, /*qp*/ 32, logits);
  std::printf("predictedSplitMode = %d   lvl3 logits = %.4f %.4f %.4f %.4f\n", predictedSplitMode, logits[5], logits[6], logits[7], logits[8]);
  // m_modeCtrl->setNewModeList(*tempCS, partitioner, predictedSplitMode, currTestMode.qp);   (EncCu.cpp:928, unchanged)
  return predictedSplitMode < 0 ? 4 : 0;
}
