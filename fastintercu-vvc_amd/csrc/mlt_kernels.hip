// mlt_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the MLT-CNN split predictor.
//
// Arithmetic spec: /root/reference/mlt-cnn-python/codes/models/archs/mlt_ctu_or_pq_arch.py:32-57,273-299
// and mlt_cu_or_pq_arch.py:96-128; preprocessing: vtm-mlt-cpp/source/Lib/EncoderLib/EncCu.cpp:810-877.
//
// Data layout: activations NHWC fp16 in HBM; weights BN-folded, fp16, pre-packed on the host in
// MFMA A-fragment order (see mlt_model.cpp) so a wave fetches one fragment as 64 x 16 contiguous
// bytes (LDS-DMA friendly, conflict-free ds_read_b128).  Accumulation, bias, residual add, GAP and
// the heads are fp32.
//
// MFMA orientation: D[cout][pixel] = sum_k W[cout][k] * X[k][pixel]  (weights = A, activations = B).
// With v_mfma_f32_32x32x16_f16 lane l (p = l&31, h = l>>5) supplies B[k = 8h+j][col p] = 8 consecutive
// input channels of pixel p -> ONE ds_read_b128 from the [pixel][cin] LDS patch, and receives
// D rows (i&3) + 8*(i>>2) + 4h of column p -> 4 consecutive output channels per register quad
// -> packed 8-byte NHWC stores.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include "mlt_kernels.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

#define LDS_PTR(p) ((__attribute__((address_space(3))) void *)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void *)(p))

// one wave-instruction: 64 lanes x 16 B global -> 1 KiB of LDS at a wave-uniform base (lane-linear).
__device__ __forceinline__ void glds16(const void *gsrc_lane, void *lds_wave_base) {
  __builtin_amdgcn_global_load_lds(GLB_PTR(gsrc_lane), LDS_PTR(lds_wave_base), 16, 0, 0);
}

__device__ __forceinline__ uint32_t udiv_magic(uint32_t n, uint64_t magic) { return (uint32_t)(((uint64_t)n * magic) >> 32); }

// ---------------------------------------------------------------------------------------------
// Generic conv (3x3 pad 1 or 1x1 pad 0, stride 1 or 2), NHWC fp16 -> NHWC fp16, fp32 accumulate.
// One workgroup: MT = 32*WPB*WAVES_P output pixels (SPW samples x TH x TW) x CT = 32*WCB*WAVES_C
// output channels.  Per 64-or-32-channel input chunk the (haloed) input patch is staged ONCE in LDS
// and reused by all taps; weights stream through a double-buffered LDS ring by LDS-DMA, one group
// of GT taps per step.
// ---------------------------------------------------------------------------------------------
template <int CIN, int COUT, int STRIDE, int TAPS, int WCB, int WPB, int WAVES_C, int WAVES_P, int GT>
__global__ __launch_bounds__(64 * WAVES_C * WAVES_P) void conv_mfma_kernel(const ConvArgs a) {
  constexpr int KC = (CIN % 64 == 0) ? 64 : 32;
  constexpr int NCHUNK = CIN / KC;
  constexpr int KS = KC / 16;
  constexpr int SLOTS = KC / 8;          // 16-byte slots per pixel
  constexpr int PS = KC * 2 + 16;        // LDS pixel stride (bytes); PS/16 odd -> conflict-free rows
  constexpr int CBT = WCB * WAVES_C;     // 32-channel blocks per workgroup tile
  constexpr int CT = 32 * CBT;
  constexpr int NW = WAVES_C * WAVES_P;
  constexpr int NT = 64 * NW;
  constexpr int NG = TAPS / GT;
  constexpr int WCHUNK = GT * KS * CBT * 1024;  // bytes of one weight step
  constexpr int PAD = TAPS == 9 ? 1 : 0;
  static_assert(TAPS % GT == 0, "tap grouping");
  static_assert(COUT % CT == 0, "cout tiling");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *patch = smem;
  char *wring = smem + a.patch_bytes;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % WAVES_C, wp = wave / WAVES_C;
  const int p = lane & 31, h = lane >> 5;

  // ---- tile decode (all powers of two) ----
  const int tw_l = a.tw_l, th_l = a.th_l, spw_l = a.spw_l;
  const int TW = 1 << tw_l, TH = 1 << th_l;
  const int hout_l = a.hout_l, hin_l = a.hin_l;
  const int Hin = 1 << hin_l;
  const int txs_l = hout_l - tw_l, tys_l = hout_l - th_l;  // tiles per row / column of one sample
  const int mtile = blockIdx.x;
  const int ctile = blockIdx.y;
  const int tx = mtile & ((1 << txs_l) - 1);
  const int ty = (mtile >> txs_l) & ((1 << tys_l) - 1);
  const int n0 = (mtile >> (txs_l + tys_l)) << spw_l;
  const int PH = a.ph, PW = a.pw, RP = a.rp, HALF = a.half;
  const int m_valid = 1 << (tw_l + th_l + spw_l);

  // ---- per-lane pixel mapping for this wave's pixel blocks ----
  int base[WPB];      // LDS byte offset of (pixel, tap (0,0), slot h)
  int opix[WPB];      // output pixel index (flattened n,y,x) or -1
#pragma unroll
  for (int j = 0; j < WPB; ++j) {
    int m = (wp * WPB + j) * 32 + p;
    bool ok = m < m_valid;
    int mm = ok ? m : 0;
    int x = mm & (TW - 1), y = (mm >> tw_l) & (TH - 1), s = mm >> (tw_l + th_l);
    base[j] = ((s * PH + y * STRIDE) * RP + (STRIDE == 2 ? x : x)) * PS + h * 16;
    int oy = (ty << th_l) + y, ox = (tx << tw_l) + x;
    opix[j] = (ok && (n0 + s) < a.n) ? ((((n0 + s) << hout_l) + oy) << hout_l) + ox : -1;
  }

  float16v acc[WCB][WPB];
#pragma unroll
  for (int i = 0; i < WCB; ++i)
#pragma unroll
    for (int j = 0; j < WPB; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const char *wsrc = (const char *)a.w + (size_t)ctile * NCHUNK * TAPS * (KS * CBT * 1024);
  const int patch_items = (1 << spw_l) * PH * PW * SLOTS;
  const int iy0 = ((ty << th_l) * STRIDE) - PAD, ix0 = ((tx << tw_l) * STRIDE) - PAD;

  for (int chunk = 0; chunk < NCHUNK; ++chunk) {
    if (chunk > 0) __syncthreads();  // everyone done reading the previous chunk's patch / weights
    // ---- stage the input patch chunk: global (16 B / lane) -> LDS ----
    for (int it = tid; it < patch_items; it += NT) {
      int slot = it & (SLOTS - 1);
      uint32_t pix = (uint32_t)it / SLOTS;
      uint32_t r = udiv_magic(pix, a.pw_magic);
      int px = pix - r * PW;
      uint32_t s = udiv_magic(r, a.ph_magic);
      int py = r - s * PH;
      int iy = iy0 + py, ix = ix0 + px;
      half8 v;
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.f;
      if (iy >= 0 && iy < Hin && ix >= 0 && ix < Hin && (int)(n0 + s) < a.n) {
        const _Float16 *src = (const _Float16 *)a.x + ((((size_t)(n0 + s) << hin_l) + iy) << hin_l) * CIN + (size_t)ix * CIN +
                              chunk * KC + slot * 8;
        v = *(const half8 *)src;
      }
      int col = STRIDE == 2 ? ((px & 1) * HALF + (px >> 1)) : px;
      *(half8 *)(patch + ((s * PH + py) * RP + col) * PS + slot * 16) = v;
    }
    // ---- first weight step of this chunk ----
    {
      const char *src = wsrc + (size_t)(chunk * TAPS) * (KS * CBT * 1024);
      for (int pi = wave; pi < WCHUNK / 1024; pi += NW) glds16(src + pi * 1024 + lane * 16, wring + pi * 1024);
    }
    __syncthreads();  // drains vmcnt (LDS-DMA landed) and makes the patch visible

#pragma unroll 1
    for (int g = 0; g < NG; ++g) {
      char *wcur = wring + (g & 1) * WCHUNK;
      if (g + 1 < NG) {
        const char *src = wsrc + (size_t)(chunk * TAPS + (g + 1) * GT) * (KS * CBT * 1024);
        char *dst = wring + ((g + 1) & 1) * WCHUNK;
        for (int pi = wave; pi < WCHUNK / 1024; pi += NW) glds16(src + pi * 1024 + lane * 16, dst + pi * 1024);
      }
#pragma unroll
      for (int tt = 0; tt < GT; ++tt) {
        const int t = g * GT + tt;
        int toff;
        if (TAPS == 9) {
          int dy = t / 3, dx = t - dy * 3;
          toff = STRIDE == 2 ? (dy * RP + (dx & 1) * HALF + (dx >> 1)) * PS : (dy * RP + dx) * PS;
        } else {
          toff = 0;
        }
        const char *bp[WPB];
#pragma unroll
        for (int j = 0; j < WPB; ++j) bp[j] = patch + base[j] + toff;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          half8 af[WCB], bf[WPB];
#pragma unroll
          for (int i = 0; i < WCB; ++i)
            af[i] = *(const half8 *)(wcur + ((tt * KS + ks) * CBT + wc * WCB + i) * 1024 + lane * 16);
#pragma unroll
          for (int j = 0; j < WPB; ++j) bf[j] = *(const half8 *)(bp[j] + ks * 32);
#pragma unroll
          for (int i = 0; i < WCB; ++i)
#pragma unroll
            for (int j = 0; j < WPB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
      }
      if (g + 1 < NG) __syncthreads();  // next weights landed (vmcnt(0)) + everyone finished this buffer
    }
  }

  // ---- epilogue: + bias (+ residual) (ReLU) -> fp16 NHWC, 8 B per register quad ----
#pragma unroll
  for (int i = 0; i < WCB; ++i) {
    const int cbase = ctile * CT + (wc * WCB + i) * 32 + 4 * h;
#pragma unroll
    for (int j = 0; j < WPB; ++j) {
      if (opix[j] < 0) continue;
      const size_t o = (size_t)opix[j] * COUT + cbase;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4v b = *(const float4v *)(a.bias + cbase + 8 * q);
        float v0 = acc[i][j][4 * q + 0] + b[0], v1 = acc[i][j][4 * q + 1] + b[1];
        float v2 = acc[i][j][4 * q + 2] + b[2], v3 = acc[i][j][4 * q + 3] + b[3];
        if (a.res) {
          const half4 r = *(const half4 *)((const _Float16 *)a.res + o + 8 * q);
          v0 += (float)r[0]; v1 += (float)r[1]; v2 += (float)r[2]; v3 += (float)r[3];
        }
        if (a.relu) {
          v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f);
        }
        half4 out;
        out[0] = (_Float16)v0; out[1] = (_Float16)v1; out[2] = (_Float16)v2; out[3] = (_Float16)v3;
        *(half4 *)((_Float16 *)a.y + o + 8 * q) = out;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Stem: raw Pel (int16) org/pred -> [org, |org-pred|] (exact integers in fp16; the 1/1023 scale of
// EncCu.cpp:836,838 is folded into the weights) -> conv3x3 2->32 (no BN, no ReLU: arch:277-278).
// K = 9 taps x 2 channels = 18, zero-padded to 32 (two MFMA k-steps).
// Tile: 8 x 32 output pixels of one CU per workgroup (4 waves x 2 pixel blocks).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stem_kernel(const StemArgs a) {
  __shared__ uint32_t patch[340];  // (org, resi) fp16 pair per pixel; max (8+2)*(32+2)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int p = lane & 31, h = lane >> 5;
  const int S = 1 << a.s_l;
  const int tw_eff = S < 32 ? S : 32;                       // S = 16: 16 x 16 tile, else 8 x 32
  const int th_eff = (256 / tw_eff) > S ? S : (256 / tw_eff);
  const int txs = S / tw_eff, tys = S / th_eff;
  const int mt = blockIdx.x;
  const int tx = mt % txs, ty = (mt / txs) % tys, n = mt / (txs * tys);
  const int pw = tw_eff + 2, ph = th_eff + 2;
  const int16_t *org = a.org + (size_t)n * a.org_cu_stride;
  const int16_t *prd = a.pred + (size_t)n * a.pred_cu_stride;
  for (int it = tid; it < ph * pw; it += 256) {
    int py = it / pw, px = it - py * pw;
    int iy = ty * th_eff + py - 1, ix = tx * tw_eff + px - 1;
    uint32_t v = 0;
    if (iy >= 0 && iy < S && ix >= 0 && ix < S) {
      uint16_t o = (uint16_t)org[(size_t)iy * a.org_row_stride + ix];   // EncCu.cpp:816
      uint16_t q = (uint16_t)prd[(size_t)iy * a.pred_row_stride + ix];  // EncCu.cpp:827
      uint16_t r = o > q ? o - q : q - o;                               // cv::absdiff, EncCu.cpp:833
      // clip to [0,1] after *1/1023 (EncCu.cpp:848-867) == clip the integer to [0,1023]
      o = o > 1023 ? 1023 : o;
      r = r > 1023 ? 1023 : r;
      half2v hv;
      hv[0] = (_Float16)(float)o;
      hv[1] = (_Float16)(float)r;
      v = *(uint32_t *)&hv;
    }
    patch[py * pw + px] = v;
  }
  // A fragments: 2 k-steps x 1 channel block, straight from global (2 KiB, L2-resident)
  const half8 a0 = *(const half8 *)((const char *)a.w + lane * 16);
  const half8 a1 = *(const half8 *)((const char *)a.w + 1024 + lane * 16);
  __syncthreads();
  const int npix = th_eff * tw_eff;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    int m = (wave * 2 + j) * 32 + p;
    bool ok = m < npix;
    int mm = ok ? m : 0;
    int x = mm % tw_eff, y = mm / tw_eff;
    // k-step 0: lane half h supplies taps 4h..4h+3 (k = 2*tap + channel); k-step 1: tap 8 (h = 0 only)
    uint32_t d[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      int t = 4 * h + e;
      int dy = t / 3, dx = t - dy * 3;
      d[e] = patch[(y + dy) * pw + x + dx];
    }
    uint32_t d8 = h == 0 ? patch[(y + 2) * pw + x + 2] : 0u;
    half8 b0, b1;
    uint32_t *b0w = (uint32_t *)&b0, *b1w = (uint32_t *)&b1;
    b0w[0] = d[0]; b0w[1] = d[1]; b0w[2] = d[2]; b0w[3] = d[3];
    b1w[0] = d8; b1w[1] = 0; b1w[2] = 0; b1w[3] = 0;
    float16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, acc, 0, 0, 0);
    if (ok) {
      int oy = ty * th_eff + y, ox = tx * tw_eff + x;
      _Float16 *dst = (_Float16 *)a.y + ((((size_t)n << a.s_l) + oy) << a.s_l) * 32 + (size_t)ox * 32 + 4 * h;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        half4 out;
        out[0] = (_Float16)acc[4 * q + 0]; out[1] = (_Float16)acc[4 * q + 1];
        out[2] = (_Float16)acc[4 * q + 2]; out[3] = (_Float16)acc[4 * q + 3];
        *(half4 *)(dst + 8 * q) = out;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// GAP + heads + argmax (arch:282-297, EncCu.cpp:913-921).  One workgroup per CU; fp32 throughout.
// logits_k = W_k . [mean_hw(feat) (C floats), poc, qp] + b_k ; split = first maximal index.
// Every class of a head runs the identical operation sequence, so identical rows tie exactly.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gap_heads_kernel(const HeadArgs a) {
  __shared__ float feat[256 + 2];
  __shared__ float lg[MLT_MAX_LOGITS_K];
  __shared__ float part[256];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float fpoc = (float)a.poc[n], fqp = (float)a.qp[n];  // EncCu.cpp:881-882 (int -> float, exact)
  int lo = 0;
  for (int hd = 0; hd < a.n_heads; ++hd) {
    const int C = a.c[hd], hw = a.hw[hd], K = a.classes[hd];
    const _Float16 *f = (const _Float16 *)a.feat[hd] + (size_t)n * hw * C;
    {
      const int ng = 256 / C, g = tid / C, c = tid - g * C;  // pixel groups x channels
      if (g < ng) {
        float s = 0.f;
        for (int i = g; i < hw; i += ng) s += (float)f[(size_t)i * C + c];
        part[g * C + c] = s;
      }
      __syncthreads();
      if (tid < C) {
        float s = 0.f;
        for (int gg = 0; gg < ng; ++gg) s += part[gg * C + tid];
        feat[tid] = s / (float)hw;
      }
    }
    if (tid == 0) { feat[C] = fpoc; feat[C + 1] = fqp; }
    __syncthreads();
    for (int k = wave; k < K; k += 4) {
      const float *w = a.w[hd] + (size_t)k * (C + 2);
      float s = 0.f;
      for (int i = lane; i < C + 2; i += 64) s += w[i] * feat[i];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
      if (lane == 0) lg[lo + k] = s + a.b[hd][k];
    }
    lo += K;
    __syncthreads();
  }
  if (tid < lo && a.logits) a.logits[(size_t)n * lo + tid] = lg[tid];
  if (tid == 0) {
    int off = 0;
    for (int hd = 0; hd < a.decision_head; ++hd) off += a.classes[hd];
    int best = 0;
    for (int k = 1; k < a.classes[a.decision_head]; ++k)
      if (lg[off + k] > lg[off + best]) best = k;
    a.split[n] = best;
  }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
template <int CIN, int COUT, int STRIDE, int TAPS, int WCB, int WPB, int WAVES_C, int WAVES_P, int GT>
static hipError_t launch_conv_t(const ConvArgs &a, int grid_x, hipStream_t st) {
  auto kern = conv_mfma_kernel<CIN, COUT, STRIDE, TAPS, WCB, WPB, WAVES_C, WAVES_P, GT>;
  constexpr int KC = (CIN % 64 == 0) ? 64 : 32;
  constexpr int CBT = WCB * WAVES_C;
  const int lds = a.patch_bytes + 2 * GT * (KC / 16) * CBT * 1024;
  static int configured = 0;
  if (configured < lds) {
    hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    configured = 160 * 1024;
  }
  dim3 grid(grid_x, COUT / (32 * CBT));
  hipLaunchKernelGGL(kern, grid, dim3(64 * WAVES_C * WAVES_P), lds, st, a);
  return hipGetLastError();
}

// wave tiling per output width (must match conv_plan() in mlt_model.cpp)
#define CONV_CASE(CIN, COUT, STRIDE, TAPS, WCB, WPB, WC, WP, GT)                                  \
  if (cin == CIN && cout == COUT && stride == STRIDE && taps == TAPS)                             \
    return launch_conv_t<CIN, COUT, STRIDE, TAPS, WCB, WPB, WC, WP, GT>(a, grid_x, st);

hipError_t mlt_launch_conv(int cin, int cout, int stride, int taps, const ConvArgs &a, int grid_x, hipStream_t st) {
  //        CIN  COUT S  T  WCB WPB WC WP GT
  CONV_CASE(32, 32, 1, 9, 1, 2, 1, 4, 9)
  CONV_CASE(32, 32, 2, 9, 1, 1, 1, 4, 9)
  CONV_CASE(32, 32, 2, 1, 1, 1, 1, 4, 1)
  CONV_CASE(32, 64, 2, 9, 2, 1, 1, 4, 3)
  CONV_CASE(32, 64, 2, 1, 2, 1, 1, 4, 1)
  CONV_CASE(64, 64, 1, 9, 2, 2, 1, 4, 1)
  CONV_CASE(64, 128, 2, 9, 2, 2, 2, 2, 1)
  CONV_CASE(64, 128, 2, 1, 2, 2, 2, 2, 1)
  CONV_CASE(128, 128, 1, 9, 2, 2, 2, 2, 1)
  CONV_CASE(128, 256, 2, 9, 2, 2, 2, 2, 1)
  CONV_CASE(128, 256, 2, 1, 2, 2, 2, 2, 1)
  CONV_CASE(256, 256, 1, 9, 2, 2, 2, 2, 1)
  // CU model (planes 32/64/96/128/256)
  CONV_CASE(64, 96, 2, 9, 3, 1, 1, 4, 1)
  CONV_CASE(64, 96, 2, 1, 3, 1, 1, 4, 1)
  CONV_CASE(96, 96, 1, 9, 3, 1, 1, 4, 3)
  CONV_CASE(96, 128, 2, 9, 2, 2, 2, 2, 3)
  CONV_CASE(96, 128, 2, 1, 2, 2, 2, 2, 1)
  return hipErrorInvalidValue;
}

int mlt_conv_tile_pixels(int cin, int cout, int stride, int taps) {
  (void)cin; (void)taps;
  if (cout == 32 || cout == 64) return stride == 2 ? 128 : 256;
  return 128;
}

int mlt_conv_cout_tile(int cout) { return cout >= 128 ? 128 : cout; }

hipError_t mlt_launch_stem(const StemArgs &a, int n, hipStream_t st) {
  const int S = 1 << a.s_l;
  const int tw = S < 32 ? S : 32;
  int th = 256 / tw; if (th > S) th = S;
  const int tiles = (S / tw) * (S / th);
  hipLaunchKernelGGL(stem_kernel, dim3(n * tiles), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t mlt_launch_heads(const HeadArgs &a, int n, hipStream_t st) {
  hipLaunchKernelGGL(gap_heads_kernel, dim3(n), dim3(256), 0, st, a);
  return hipGetLastError();
}
