// mlt_kernels.h -- launch interface between the host runtime (mlt_api.cpp) and mlt_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MLT_MAX_HEADS_K 4
#define MLT_MAX_LOGITS_K 16

struct ConvArgs {
  const void *x;      // input  [n][Hin][Hin][CIN]  fp16 NHWC
  void *y;            // output [n][Hout][Hout][COUT] fp16 NHWC
  const void *w;      // packed fp16 weights (mlt_model.cpp: pack_conv)
  const float *bias;  // folded BN bias per output channel
  const void *res;    // optional residual, same shape as y (NULL: none)
  int n;
  int hin_l, hout_l;           // log2 of input / output height (= width)
  int tw_l, th_l, spw_l;       // log2 of tile width, tile height, samples per workgroup
  int ph, pw, rp, half;        // patch rows, cols, row pitch (pixels), parity-split half width
  uint64_t pw_magic, ph_magic; // ceil(2^32 / d)
  int patch_bytes;             // LDS bytes reserved for the patch (multiple of 1024)
  int relu;
};

struct StemArgs {
  const int16_t *org, *pred;       // Pel planes
  long org_row_stride, org_cu_stride, pred_row_stride, pred_cu_stride;  // in elements
  const void *w;                   // packed stem weights (2 KiB)
  void *y;                         // [n][S][S][32] fp16
  int s_l;                         // log2(S)
};

struct HeadArgs {
  const void *feat[MLT_MAX_HEADS_K];  // stage outputs [n][hw][C] fp16
  const float *w[MLT_MAX_HEADS_K];    // [classes][C+2] fp32
  const float *b[MLT_MAX_HEADS_K];
  int c[MLT_MAX_HEADS_K], hw[MLT_MAX_HEADS_K], classes[MLT_MAX_HEADS_K];
  int n_heads, decision_head;
  const int32_t *poc, *qp;
  float *logits;   // [n][sum classes] or NULL
  int32_t *split;  // [n]
};

hipError_t mlt_launch_conv(int cin, int cout, int stride, int taps, const ConvArgs &a, int grid_x, hipStream_t st);
hipError_t mlt_launch_stem(const StemArgs &a, int n, hipStream_t st);
hipError_t mlt_launch_heads(const HeadArgs &a, int n, hipStream_t st);
int mlt_conv_tile_pixels(int cin, int cout, int stride, int taps);  // output pixels per workgroup
int mlt_conv_cout_tile(int cout);                                   // output channels per workgroup
