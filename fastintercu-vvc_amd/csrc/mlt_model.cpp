// mlt_model.cpp -- host side of weight handling: MLTW blob -> BN-folded, fp16, MFMA-fragment-packed
// tensors.  Replaces what torch::jit::load did per call in the reference (EncCu.cpp:894-900); runs once.
//
// BN folding (eval mode, eps 1e-5; arch:39,42,49 / SURVEY.md A.2):  w' = w * gamma/sqrt(var+eps),
// b' = beta - mean * gamma/sqrt(var+eps), in double, before the cast to fp16.
//
// fp16 rounding: "tap-diffused" -- within each (cout, cin) pair the 9 taps are rounded in order and each
// target is corrected by the accumulated rounding error of the previous taps, so sum_taps(w_fp16) tracks
// sum_taps(w) to half an ulp.  Activations are spatially smooth, so the coherent part of the weight
// rounding error (the part global average pooling cannot average away) cancels.  Every value is still
// within one fp16 ulp of the folded fp32 weight.
//
// Exact mode: each folded weight is stored as an fp16 pair hi = rn(w), lo = rn(w - hi) (two planes); the
// kernels accumulate Wh*Xh + Wh*Xl + Wl*Xh, so no diffusion is needed there.
#include "mlt_model.h"

#include "mlt_kernels.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace mlt {

uint16_t f32_to_f16(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  x &= 0x7FFFFFFFu;
  if (x >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | (x > 0x7F800000u ? 0x200u : 0));
  if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);  // rounds to inf
  if (x < 0x38800000u) {                                    // subnormal half or zero
    if (x < 0x33000000u) return (uint16_t)sign;
    const int e = (int)(x >> 23);
    uint32_t m = (x & 0x7FFFFFu) | 0x800000u;
    const int shift = 126 - e;  // 14..24
    uint32_t r = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (r & 1))) ++r;
    return (uint16_t)(sign | r);
  }
  uint32_t r = ((x - 0x38000000u) >> 13);
  const uint32_t rem = x & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) ++r;
  return (uint16_t)(sign | r);
}

float f16_to_f32(uint16_t hbits) {
  const uint32_t sign = (uint32_t)(hbits & 0x8000u) << 16;
  uint32_t e = (hbits >> 10) & 0x1F, m = hbits & 0x3FF, x;
  if (e == 0) {
    if (m == 0) x = sign;
    else {
      int sh = 0;
      while (!(m & 0x400)) { m <<= 1; ++sh; }
      x = sign | ((uint32_t)(113 - sh) << 23) | ((m & 0x3FF) << 13);
    }
  } else if (e == 31) x = sign | 0x7F800000u | (m << 13);
  else x = sign | ((e + 112) << 23) | (m << 13);
  float f;
  std::memcpy(&f, &x, 4);
  return f;
}

// OCP FP8 e4m3fn (bias 7, no infinities, max 448, subnormal step 2^-9): round to nearest even, saturating
static uint8_t f32_to_e4m3(float f) {
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint8_t sign = (uint8_t)((x >> 24) & 0x80u);
  float a = std::fabs(f);
  if (!(a == a)) return (uint8_t)(sign | 0x7Fu);
  if (a >= 448.f) return (uint8_t)(sign | 0x7Eu);  // saturate to the largest finite value
  if (a < 0.0009765625f) return sign;               // below half the smallest subnormal (2^-10): zero
  int e;
  const float m = std::frexp(a, &e);                // a = m * 2^e, m in [0.5, 1)
  int exp = e - 1;                                  // a = (2m) * 2^exp, 2m in [1, 2)
  if (exp < -6) {                                   // subnormal: multiples of 2^-9
    const float q = std::nearbyint(a * 512.f);      // (default rounding mode: to nearest even)
    const int qi = (int)q;
    return (uint8_t)(sign | (qi >= 8 ? 0x08u : (uint8_t)qi));
  }
  float frac = (2.f * m - 1.f) * 8.f;               // 3 mantissa bits
  int qi = (int)std::nearbyint(frac);
  if (qi == 8) { qi = 0; ++exp; }
  if (exp > 8 || (exp == 8 && qi > 6)) return (uint8_t)(sign | 0x7Eu);
  return (uint8_t)(sign | ((exp + 7) << 3) | qi);
}

#pragma pack(push, 1)
struct BlobHead { char magic[4]; uint32_t version, arch, n; };
struct BlobEntry { char name[64]; uint32_t ndim, dims[4]; uint64_t off, numel; };
#pragma pack(pop)

struct BlobView {
  const BlobEntry *e = nullptr;
  uint32_t n = 0;
  const float *data = nullptr;
  size_t nfloats = 0;
  const float *find(const char *name, uint64_t numel, std::string &err) const {
    for (uint32_t i = 0; i < n; ++i)
      if (std::strncmp(e[i].name, name, 64) == 0) {
        if (e[i].numel != numel || e[i].off + e[i].numel > nfloats) { err = std::string("tensor ") + name + ": bad size"; return nullptr; }
        return data + e[i].off;
      }
    err = std::string("tensor ") + name + " missing from blob";
    return nullptr;
  }
};

static void fold_scale(const BlobView &b, const std::string &prefix, int c, std::vector<double> &scale, std::vector<float> &bias, std::string &err) {
  const float *g = b.find((prefix + ".weight").c_str(), c, err);
  const float *be = b.find((prefix + ".bias").c_str(), c, err);
  const float *mu = b.find((prefix + ".running_mean").c_str(), c, err);
  const float *var = b.find((prefix + ".running_var").c_str(), c, err);
  scale.assign(c, 1.0);
  bias.assign(c, 0.f);
  if (!g || !be || !mu || !var) return;
  for (int i = 0; i < c; ++i) {
    const double s = (double)g[i] / std::sqrt((double)var[i] + 1e-5);
    scale[i] = s;
    bias[i] = (float)((double)be[i] - (double)mu[i] * s);
  }
}

// w: torch layout [cout][cin][taps]; scale: per-cout multiplier; w_sc (optional): [cout][cin] 1x1 shortcut.
// Output: fragment-packed fp16 (see header).
static void pack_conv(PackedConv &pc, const float *w, const std::vector<double> &scale, const float *w_sc,
                      const std::vector<double> &scale_sc, int rounding = 0) {
  const int cin = pc.cin, cout = pc.cout, taps = pc.taps, tt = taps + (w_sc ? 1 : 0);
  const int KC = pc.kc, NCHUNK = cin / KC, KS = KC / 16, CT = pc.ct, CBT = CT / 32;
  pc.plane_halves = (size_t)cout * cin * tt;
  const bool two = pc.exact || pc.w2;  // (hi, lo) planes, plain rounding
  // the layers the 128- and 256-channel chain_kernel<..., W2, LO8> stream: FP8 lo plane (see mlt_model.h).  The 64-channel chain keeps an fp16 lo
  // plane: a wave owns 128 pixels there (8 accumulators), and the e4m3 copy of four pixel blocks' fragments does not fit its registers (284 B of scratch)
  pc.lo8 = pc.w2 && KC == 64 && !w_sc && cin >= 128;
  pc.w.assign(pc.lo8 ? pc.plane_halves + pc.plane_halves / 2 : pc.plane_halves * (two ? 2 : 1), 0);
  // power-of-two storage scale: largest 2^s <= 2^6 keeping every stored weight below 2^12
  double wmax = 1e-30;
  for (int co = 0; co < cout; ++co) {
    for (size_t k = 0; k < (size_t)cin * taps; ++k) wmax = std::max(wmax, std::fabs((double)w[(size_t)co * cin * taps + k] * scale[co]));
    if (w_sc) for (int ci = 0; ci < cin; ++ci) wmax = std::max(wmax, std::fabs((double)w_sc[(size_t)co * cin + ci] * scale_sc[co]));
  }
  int sexp = two ? 6 : 0;
  while (sexp > 0 && wmax * std::ldexp(1.0, sexp) >= 4096.0) --sexp;
  const double wmul = std::ldexp(1.0, sexp);
  pc.acc_scale = (float)std::ldexp(1.0, -sexp);
  auto put = [&](uint16_t &hi_slot, double exact_unscaled, double &err) {
    const double exact = exact_unscaled * wmul;
    if (two) {
      const uint16_t q = f32_to_f16((float)exact);
      hi_slot = q;
      (&hi_slot)[pc.plane_halves] = f32_to_f16((float)(exact - (double)f16_to_f32(q)));
    } else {
      static const bool plain = [] { const char *t = std::getenv("MLT_TUNING"), *e = std::getenv("MLT_PLAIN_ROUNDING"); return t && t[0] == '1' && e; }();  // experiment: no tap diffusion
      const uint16_t q = f32_to_f16((float)(plain ? exact : exact - err));
      err += (double)f16_to_f32(q) - exact;
      hi_slot = q;
    }
  };
  std::vector<double> lo8v;  // lo8: the exact lo residuals, encoded after their common scale is known
  if (pc.lo8) lo8v.assign((size_t)cout * cin * tt, 0.0);
  auto put8 = [&](uint16_t &hi_slot, double exact_unscaled, size_t flat) {
    const double exact = exact_unscaled * wmul;
    const uint16_t q = f32_to_f16((float)exact);
    hi_slot = q;
    lo8v[flat] = exact - (double)f16_to_f32(q);
  };
  // rounding realisations of the single-pass model (mlt_model.h): tap order of the error diffusion, and whether the error carries across cin
  static const int kOrder[MLT_N_ROUNDINGS][9] = {{0, 1, 2, 3, 4, 5, 6, 7, 8}, {0, 1, 2, 3, 4, 5, 6, 7, 8}, {8, 7, 6, 5, 4, 3, 2, 1, 0},
                                                 {0, 3, 6, 7, 4, 1, 2, 5, 8}, {4, 5, 2, 1, 0, 3, 6, 7, 8}, {0, 1, 2, 5, 4, 3, 6, 7, 8}};
  static const bool kCarry[MLT_N_ROUNDINGS] = {false, true, false, false, false, true};
  const int rv = (!two && taps == 9 && rounding > 0 && rounding < MLT_N_ROUNDINGS) ? rounding : 0;
  for (int co = 0; co < cout; ++co) {
    const int ctile = co / CT, cbt = (co % CT) / 32, r = co % 32;
    double err = 0.0;  // running (sum of rounded) - (sum of exact) over the taps of one (co, ci) -- or of one co (carried roundings)
    for (int ci = 0; ci < cin; ++ci) {
      const int chunk = ci / KC, kk = ci % KC, ks = kk / 16, hh = (kk % 16) / 8, j = kk % 8;
      auto at = [&](int t) -> uint16_t & {
        return pc.w[((((size_t)(ctile * NCHUNK + chunk) * tt + t) * KS + ks) * CBT + cbt) * 512 + (size_t)(hh * 32 + r) * 8 + j];
      };
      if (!kCarry[rv]) err = 0.0;
      if (pc.lo8) {
        for (int t = 0; t < taps; ++t) put8(at(t), (double)w[((size_t)co * cin + ci) * taps + t] * scale[co], ((size_t)co * cin + ci) * tt + t);
        continue;
      }
      for (int k = 0; k < taps; ++k) {
        const int t = taps == 9 ? kOrder[rv][k] : k;
        put(at(t), (double)w[((size_t)co * cin + ci) * taps + t] * scale[co], err);
      }
      double err_sc = 0.0;
      if (w_sc) put(at(taps), (double)w_sc[(size_t)co * cin + ci] * scale_sc[co], err_sc);
    }
  }
  if (pc.xl) {
    // exact-lite: the lo plane becomes the FP8 A operand of the cross-term MFMA (mlt_model.h).  Power-of-two scales put the largest |Wl| / |Wh|
    // into [112, 224] -- inside e4m3's range (max 448) with headroom; the MFMA's E8M0 block scales undo them.
    const size_t np = pc.plane_halves;
    std::vector<uint16_t> lo16(pc.w.begin() + np, pc.w.end());
    double lomax = 1e-30, himax = 1e-30;
    for (size_t k = 0; k < np; ++k) { lomax = std::max(lomax, std::fabs((double)f16_to_f32(lo16[k]))); himax = std::max(himax, std::fabs((double)f16_to_f32(pc.w[k]))); }
    pc.xl_ewl = std::min(60, (int)std::floor(std::log2(224.0 / lomax)));
    pc.xl_ewh = std::min(60, (int)std::floor(std::log2(224.0 / himax)));
    const double ml = std::ldexp(1.0, pc.xl_ewl), mh = std::ldexp(1.0, pc.xl_ewh);
    uint8_t *lo = (uint8_t *)(pc.w.data() + np);
    std::memset(lo, 0, np * 2);
    for (int co = 0; co < cout; ++co) {
      const int ctile = co / CT, cbt = (co % CT) / 32, r = co % 32;
      for (int ci = 0; ci < cin; ++ci) {
        const int chunk = ci / KC, kk = ci % KC, ks = kk / 16, hh = (kk % 16) / 8, j = kk % 8;
        for (int t = 0; t < tt; ++t) {
          const size_t hidx = ((((size_t)(ctile * NCHUNK + chunk) * tt + t) * KS + ks) * CBT + cbt) * 512 + (size_t)(hh * 32 + r) * 8 + j;  // this weight in the fp16 planes
          // FP8 operand: 16-byte half g of lane (kk >> 4) * 32 + r sits in k-step slot g of the lo plane; half 0 = Wl (K block 0: bytes 0-15 of BOTH
          // lane halves, scaled by lanes 0-31), half 1 = Wh (K block 1) -- scripts/probes/f8_mfma_scale_probe2.hip
          const size_t blk0 = ((((size_t)(ctile * NCHUNK + chunk) * tt + t) * KS + 0) * CBT + cbt) * 1024, blk1 = blk0 + (size_t)CBT * 1024;
          const size_t lane16 = (size_t)((kk >> 4) * 32 + r) * 16 + (kk & 15);
          lo[blk0 + lane16] = f32_to_e4m3((float)((double)f16_to_f32(lo16[hidx]) * ml));
          lo[blk1 + lane16] = f32_to_e4m3((float)((double)f16_to_f32(pc.w[hidx]) * mh));
        }
      }
    }
  }
  if (pc.lo8) {
    double lomax = 1e-30;
    for (double v : lo8v) lomax = std::max(lomax, std::fabs(v));
    pc.lo8_exp = (int)std::floor(std::log2(224.0 / lomax));  // largest residual -> [112, 224]: inside e4m3's range (max 448) with headroom
    if (pc.lo8_exp > 60) pc.lo8_exp = 60;
    const double mul8 = std::ldexp(1.0, pc.lo8_exp);
    uint8_t *lo = (uint8_t *)(pc.w.data() + pc.plane_halves);
    for (int co = 0; co < cout; ++co) {
      const int ctile = co / CT, cbt = (co % CT) / 32, r = co % 32;
      for (int ci = 0; ci < cin; ++ci) {
        const int chunk = ci / KC, kk = ci % KC, hh = kk / 32, jb = kk % 32;  // channel kk of the chunk <-> byte jb of lane half hh
        for (int t = 0; t < tt; ++t)
          // fragment = [16-byte half of the lane's 32 bytes][lane][16 bytes]: two lane-linear (conflict-free) ds_read_b128 per lane
          lo[((((size_t)(ctile * NCHUNK + chunk) * tt + t) * CBT + cbt) * 2048) + (size_t)(jb >> 4) * 1024 + (size_t)(hh * 32 + r) * 16 + (jb & 15)] =
              f32_to_e4m3((float)(lo8v[((size_t)co * cin + ci) * tt + t] * mul8));
      }
    }
  }
}

// First layer, composed (kernel: stem5_kernel).  The stem conv has no BN / ReLU (arch:277-278), so with
//   W1 = layer0.0.conv1 folded with bn1, Wsc = layer0.0.shortcut folded with its BN, Ws = stem conv1:
//   W5[co][c][u][v]  = sum_cm sum_{a+b=(u,v)} W1[co][cm][a] * Ws[cm][c][b]      (5x5, stride 2, pad 2)
//   Wsc3[co][c][b]   = sum_cm Wsc[co][cm] * Ws[cm][c][b]                        (3x3, stride 2, pad 1)
// conv1 pads the STEM activation with zeros, the composed conv pads the INPUT; they differ where conv1 reads stem
// row -1 / column -1, i.e. for output row 0 / column 0.  The difference is cancelled by extra K slots:
//   Top[co][c][v]  = -sum_cm sum_{ax+bx=v} W1[co][cm][0][ax] * Ws[cm][c][2][bx]   x in(0, 2x-2+v)    (output row 0)
//   Left[co][c][u] = -sum_cm sum_{ay+by=u} W1[co][cm][ay][0] * Ws[cm][c][by][2]   x in(2y-2+u, 0)    (output column 0)
//   Corner[co][c]  = +sum_cm W1[co][cm][0][0] * Ws[cm][c][2][2]                   x in(0, 0)         (output (0,0))
// K slot order (one slot = the (org, resi) pair of one tap): 25 taps, 5 top, 5 left, 1 corner, 4 zero = 40 slots =
// 5 MFMA k-steps; then the 9 shortcut taps in 2 k-steps.  Inputs are exact integers in fp16, so the 1/1023 scale
// (EncCu.cpp:836,838) rides in the weights together with 2^12 (acc_scale = 2^-12) to keep them in fp16's normal range.
static void pack_stem5(PackedConv &pc, const float *ws, const float *w1, const std::vector<double> &s1, const float *wsc,
                       const std::vector<double> &ssc) {
  auto Ws = [&](int cm, int c, int by, int bx) { return (double)ws[((cm * 2 + c) * 3 + by) * 3 + bx]; };
  auto W1 = [&](int co, int cm, int ay, int ax) { return (double)w1[((co * 32 + cm) * 3 + ay) * 3 + ax] * s1[co]; };
  auto Wsc = [&](int co, int cm) { return (double)wsc[co * 32 + cm] * ssc[co]; };
  const double mul = (double)(float)(1.0 / 1023) * 4096.0;
  pc.acc_scale = 1.0f / 4096.0f;
  pc.plane_halves = 7 * 64 * 8;
  const bool two = pc.exact || pc.w2;
  pc.w.assign(pc.plane_halves * (two ? 2 : 1), 0);
  auto store = [&](int kstep0, int slot, int co, int c, double exact, double *err) {
    const int k = 2 * slot + c, ks = kstep0 + k / 16, hh = (k % 16) / 8, j = k % 8;
    const size_t idx = ((size_t)ks * 64 + hh * 32 + co) * 8 + j;
    if (two) {
      const uint16_t q = f32_to_f16((float)exact);
      pc.w[idx] = q;
      pc.w[pc.plane_halves + idx] = f32_to_f16((float)(exact - (double)f16_to_f32(q)));
    } else {
      static const bool plain = [] { const char *t = std::getenv("MLT_TUNING"), *e = std::getenv("MLT_PLAIN_ROUNDING"); return t && t[0] == '1' && e; }();
      const double tgt = (err && !plain) ? exact - *err : exact;
      const uint16_t q = f32_to_f16((float)tgt);
      if (err) *err += (double)f16_to_f32(q) - exact;
      pc.w[idx] = q;
    }
  };
  for (int co = 0; co < 32; ++co)
    for (int c = 0; c < 2; ++c) {
      double w5[5][5] = {}, top[5] = {}, left[5] = {}, corner = 0.0, sc3[3][3] = {};
      for (int cm = 0; cm < 32; ++cm) {
        for (int ay = 0; ay < 3; ++ay)
          for (int ax = 0; ax < 3; ++ax)
            for (int by = 0; by < 3; ++by)
              for (int bx = 0; bx < 3; ++bx) w5[ay + by][ax + bx] += W1(co, cm, ay, ax) * Ws(cm, c, by, bx);
        for (int ax = 0; ax < 3; ++ax)
          for (int bx = 0; bx < 3; ++bx) top[ax + bx] -= W1(co, cm, 0, ax) * Ws(cm, c, 2, bx);
        for (int ay = 0; ay < 3; ++ay)
          for (int by = 0; by < 3; ++by) left[ay + by] -= W1(co, cm, ay, 0) * Ws(cm, c, by, 2);
        corner += W1(co, cm, 0, 0) * Ws(cm, c, 2, 2);
        for (int by = 0; by < 3; ++by)
          for (int bx = 0; bx < 3; ++bx) sc3[by][bx] += Wsc(co, cm) * Ws(cm, c, by, bx);
      }
      double err = 0.0;  // tap-diffused rounding over the 25 taps (fast mode), plain rounding for the rest
      for (int t = 0; t < 25; ++t) store(0, t, co, c, w5[t / 5][t % 5] * mul, &err);
      for (int v = 0; v < 5; ++v) store(0, 25 + v, co, c, top[v] * mul, nullptr);
      for (int u = 0; u < 5; ++u) store(0, 30 + u, co, c, left[u] * mul, nullptr);
      store(0, 35, co, c, corner * mul, nullptr);
      double err_sc = 0.0;
      for (int t = 0; t < 9; ++t) store(5, t, co, c, sc3[t / 3][t % 3] * mul, &err_sc);
    }
}

// The same composed first layer, packed for stem_block_kernel (fast arithmetic; round 3).  K is ordered so that a lane's B fragment is
// 16 CONTIGUOUS bytes of the row-major raw (org, |org-pred|) patch -- one ds_read2_b64 instead of four gathered ds_read_b32:
//   k-steps 0..4 (main, one per window row dy): slot s = dx = 0..7, pixel (2y-2+dy, 2x-2+dx); weights W5[dy][dx] for dx < 5, ZERO for dx 5..7
//   k-step  5    (shortcut): slots 0..3 = raw row 2y-1, columns 2x-2..2x+1 (weight 0, Wsc3[0][0..2]); slots 4..7 = row 2y the same way
//   k-step  6    (shortcut): slots 0..3 = raw row 2y+1; slots 4..7 zero weights
//   k-step  7    (border, output column 0 / pixel (0,0)): slots 0..4 = Left[0..4], slot 5 = Corner, rest zero
//   k-step  8    (border, output row 0): slots 0..4 = Top[0..4], rest zero
// (slot s of a k-step = K entries 2s (org) and 2s+1 (resi); lane half hh supplies slots 4hh .. 4hh+3).  The 25 main taps and the 9
// shortcut taps are rounded in the order of pack_stem5 (tap-diffused), so every stored fp16 value equals the one stem5_kernel uses.
static void pack_stem_b(PackedConv &pc, const float *ws, const float *w1, const std::vector<double> &s1, const float *wsc,
                        const std::vector<double> &ssc) {
  auto Ws = [&](int cm, int c, int by, int bx) { return (double)ws[((cm * 2 + c) * 3 + by) * 3 + bx]; };
  auto W1 = [&](int co, int cm, int ay, int ax) { return (double)w1[((co * 32 + cm) * 3 + ay) * 3 + ax] * s1[co]; };
  auto Wsc = [&](int co, int cm) { return (double)wsc[co * 32 + cm] * ssc[co]; };
  const double mul = (double)(float)(1.0 / 1023) * 4096.0;
  pc.acc_scale = 1.0f / 4096.0f;
  pc.plane_halves = 9 * 64 * 8;
  pc.w.assign(pc.plane_halves * (pc.w2 ? 2 : 1), 0);
  auto store = [&](int ks, int slot, int co, int c, double exact, double *err) {
    const int hh = slot / 4, j = (slot % 4) * 2 + c;
    const size_t idx = ((size_t)ks * 64 + hh * 32 + co) * 8 + j;
    if (pc.w2) {  // (hi, lo) planes, plain rounding: the values pack_stem5 stores for the same mode
      const uint16_t q = f32_to_f16((float)exact);
      pc.w[idx] = q;
      pc.w[pc.plane_halves + idx] = f32_to_f16((float)(exact - (double)f16_to_f32(q)));
      return;
    }
    static const bool plain = [] { const char *t = std::getenv("MLT_TUNING"), *e = std::getenv("MLT_PLAIN_ROUNDING"); return t && t[0] == '1' && e; }();
    const double tgt = (err && !plain) ? exact - *err : exact;
    const uint16_t q = f32_to_f16((float)tgt);
    if (err) *err += (double)f16_to_f32(q) - exact;
    pc.w[idx] = q;
  };
  for (int co = 0; co < 32; ++co)
    for (int c = 0; c < 2; ++c) {
      double w5[5][5] = {}, top[5] = {}, left[5] = {}, corner = 0.0, sc3[3][3] = {};
      for (int cm = 0; cm < 32; ++cm) {
        for (int ay = 0; ay < 3; ++ay)
          for (int ax = 0; ax < 3; ++ax)
            for (int by = 0; by < 3; ++by)
              for (int bx = 0; bx < 3; ++bx) w5[ay + by][ax + bx] += W1(co, cm, ay, ax) * Ws(cm, c, by, bx);
        for (int ax = 0; ax < 3; ++ax)
          for (int bx = 0; bx < 3; ++bx) top[ax + bx] -= W1(co, cm, 0, ax) * Ws(cm, c, 2, bx);
        for (int ay = 0; ay < 3; ++ay)
          for (int by = 0; by < 3; ++by) left[ay + by] -= W1(co, cm, ay, 0) * Ws(cm, c, by, 2);
        corner += W1(co, cm, 0, 0) * Ws(cm, c, 2, 2);
        for (int by = 0; by < 3; ++by)
          for (int bx = 0; bx < 3; ++bx) sc3[by][bx] += Wsc(co, cm) * Ws(cm, c, by, bx);
      }
      double err = 0.0;
      for (int t = 0; t < 25; ++t) store(t / 5, t % 5, co, c, w5[t / 5][t % 5] * mul, &err);
      for (int v = 0; v < 5; ++v) store(8, v, co, c, top[v] * mul, nullptr);
      for (int u = 0; u < 5; ++u) store(7, u, co, c, left[u] * mul, nullptr);
      store(7, 5, co, c, corner * mul, nullptr);
      double err_sc = 0.0;
      for (int t = 0; t < 9; ++t) {
        const int by = t / 3, bx = t % 3;
        store(by < 2 ? 5 : 6, (by == 1 ? 4 : 0) + bx + 1, co, c, sc3[by][bx] * mul, &err_sc);
      }
    }
}

bool build_model(const void *blob, size_t bytes, int mode, int size, Model &m, std::string &err, int rounding) {
  const bool xlite = mode == MLT_MODEL_XLITE;   // exact-lite: the exact packing (two planes, exact tiling) with FP8 lo planes in the 3x3 convs (mlt_model.h)
  const bool exact = mode == MLT_MODEL_EXACT || xlite, w2 = mode == MLT_MODEL_W2;
  if (bytes < sizeof(BlobHead)) { err = "blob too small"; return false; }
  const BlobHead *h = (const BlobHead *)blob;
  if (std::memcmp(h->magic, "MLTW", 4) != 0 || h->version != 1 || h->arch > 1) { err = "not an MLTW v1 blob"; return false; }
  const size_t hdr = sizeof(BlobHead) + (size_t)h->n * sizeof(BlobEntry);
  if (bytes < hdr) { err = "truncated blob header"; return false; }
  std::vector<float> data((bytes - hdr) / 4);  // aligned copy
  std::memcpy(data.data(), (const char *)blob + hdr, data.size() * 4);
  BlobView b;
  b.e = (const BlobEntry *)((const char *)blob + sizeof(BlobHead));
  b.n = h->n;
  b.data = data.data();
  b.nfloats = data.size();

  m = Model();
  m.arch = (int)h->arch;
  m.exact = exact;
  m.xl = xlite;
  m.w2 = w2;
  m.rounding = (exact || w2) ? 0 : rounding;
  static const int planes_ctu[4] = {32, 64, 128, 256}, planes_cu[5] = {32, 64, 96, 128, 256};  // arch:243-256 / cu arch:63-79
  static const int cls_ctu[3] = {2, 3, 4}, cls_cu[4] = {2, 3, 4, 6};
  m.n_stages = m.arch == 0 ? 4 : 5;
  m.n_heads = m.n_stages - 1;

  err.clear();
  {  // first layer: stem composed with layer0.0.conv1 (+bn1) and layer0.0.shortcut (+BN)
    const float *ws = b.find("conv1.weight", 32 * 2 * 9, err);
    const float *w1 = b.find("layer0.0.conv1.weight", 32 * 32 * 9, err);
    const float *wsc = b.find("layer0.0.shortcut.0.weight", 32 * 32, err);
    if (!ws || !w1 || !wsc) return false;
    std::vector<double> s1, ssc;
    m.stem.cin = 2; m.stem.cout = 32; m.stem.taps = 25; m.stem.stride = 2; m.stem.kc = 32; m.stem.ct = 32; m.stem.has_sc = true;
    m.stem.exact = exact; m.stem.w2 = w2;
    fold_scale(b, "layer0.0.bn1", 32, s1, m.stem.bias, err);
    fold_scale(b, "layer0.0.shortcut.1", 32, ssc, m.stem.bias_sc, err);
    if (!err.empty()) return false;
    pack_stem5(m.stem, ws, w1, s1, wsc, ssc);
    if (!exact) {  // second packing of the same weights for stem_block_kernel (contiguous B fragments)
      m.stem_b = PackedConv();
      m.stem_b.cin = 2; m.stem_b.cout = 32; m.stem_b.taps = 25; m.stem_b.stride = 2; m.stem_b.kc = 32; m.stem_b.ct = 32; m.stem_b.has_sc = true;
      m.stem_b.w2 = w2;
      pack_stem_b(m.stem_b, ws, w1, s1, wsc, ssc);
      m.stem_b.bias = m.stem.bias; m.stem_b.bias_sc = m.stem.bias_sc;
    }
  }
  int cin = 32;
  char nm[96];
  for (int s = 0; s < m.n_stages; ++s) {
    const int c = m.arch == 0 ? planes_ctu[s] : planes_cu[s];
    m.planes[s] = c;
    for (int bi = 0; bi < 2; ++bi) {
      Block &B = m.blocks[s][bi];
      const int bin = bi == 0 ? cin : c, st = bi == 0 ? 2 : 1;  // _make_layer strides [2,1] (arch:265-271)
      auto make = [&](PackedConv &pc, const char *wname, const char *bnname, int ci, int stride, bool with_sc) -> bool {
        pc.cin = ci; pc.cout = c; pc.taps = 9; pc.stride = stride; pc.has_sc = with_sc; pc.exact = exact; pc.w2 = w2;
        ConvCfg cfg;
        if (!mlt_conv_cfg(ci, c, stride, exact ? 1 : 0, &cfg)) { err = "no kernel configuration for this layer shape"; return false; }
        pc.kc = cfg.kc; pc.ct = cfg.ct; pc.mt = cfg.mt; pc.gt = cfg.gt; pc.gt_w2 = cfg.gt_w2; pc.dma = cfg.dma; pc.mt_dma = cfg.mt_dma; pc.lat = cfg.lat;
        std::snprintf(nm, sizeof nm, "layer%d.%d.%s", s, bi, wname);
        const float *w = b.find(nm, (uint64_t)c * ci * 9, err);
        if (!w) return false;
        // A 3x3 pad-1 stride-1 conv on a 1x1 map touches only its centre tap (the other eight multiply the zero padding),
        // so for the stages that run on 1x1 maps at this CU size only that tap is packed and the kernel runs as a 1x1 conv:
        // bit-identical result (the dropped products are exact zeros), 9x less weight traffic and MFMA work.
        std::vector<float> centre;
        const int hmap = (size >> (s + 1)) > 0 ? (size >> (s + 1)) : 1;  // output height of stage s
        const int hin = (size >> s) > 0 ? (size >> s) : 1;  // input height of stage s
        // (the same holds for the stride-2 conv + shortcut of a stage whose INPUT is 1x1: every tap but the centre is padding)
        if (((stride == 1 && !with_sc && hmap == 1) || (stride == 2 && with_sc && hin == 1)) && mlt_conv_has_centre_variant(ci, c)) {
          centre.resize((size_t)c * ci);
          for (size_t k = 0; k < centre.size(); ++k) centre[k] = w[k * 9 + 4];
          w = centre.data();
          pc.taps = 1;
          pc.gt = 1;
          pc.gt_w2 = 1;
        }
        std::vector<double> scale, scale_sc;
        std::snprintf(nm, sizeof nm, "layer%d.%d.%s", s, bi, bnname);
        fold_scale(b, nm, c, scale, pc.bias, err);
        if (!err.empty()) return false;
        const float *wsc = nullptr;
        if (with_sc) {  // shortcut = Sequential(conv1x1(stride), BN) (arch:46-50)
          std::snprintf(nm, sizeof nm, "layer%d.%d.shortcut.0.weight", s, bi);
          wsc = b.find(nm, (uint64_t)c * ci, err);
          if (!wsc) return false;
          std::snprintf(nm, sizeof nm, "layer%d.%d.shortcut.1", s, bi);
          fold_scale(b, nm, c, scale_sc, pc.bias_sc, err);
          if (!err.empty()) return false;
        }
        pc.xl = xlite && pc.taps == 9 && pc.kc == 32;   // (the centre-tap forms of the small models' 1x1 maps and the composed first layer stay exact)
        pack_conv(pc, w, scale, wsc, scale_sc, m.rounding);
        return true;
      };
      const bool has_sc = (st != 1 || bin != c);  // arch:44-45
      if (!(s == 0 && bi == 0) &&  // layer0.0.conv1 + shortcut live in the composed first layer (m.stem)
          !make(B.conv1, "conv1.weight", "bn1", bin, st, has_sc)) return false;
      // second packing of the stride-2 conv for the whole-stage kernel: 16-channel chunks, all 10 "taps" of a chunk = one 40 KiB step
      if (!exact && !w2 && bi == 0 && s >= 1 && has_sc && B.conv1.taps == 9 && mlt_stage_supported(c, (size >> (s + 1)) > 0 ? (size >> (s + 1)) : 1)) {
        PackedConv &pc = B.conv1_s2c;
        pc = PackedConv();
        pc.cin = bin; pc.cout = c; pc.taps = 9; pc.stride = 2; pc.has_sc = true; pc.exact = false;
        pc.kc = 16; pc.ct = 128; pc.gt = 10;
        std::snprintf(nm, sizeof nm, "layer%d.%d.conv1.weight", s, bi);
        const float *w = b.find(nm, (uint64_t)c * bin * 9, err);
        std::snprintf(nm, sizeof nm, "layer%d.%d.shortcut.0.weight", s, bi);
        const float *wsc = b.find(nm, (uint64_t)c * bin, err);
        if (!w || !wsc) return false;
        std::vector<double> scale, scale_sc;
        std::snprintf(nm, sizeof nm, "layer%d.%d.bn1", s, bi);
        fold_scale(b, nm, c, scale, pc.bias, err);
        std::snprintf(nm, sizeof nm, "layer%d.%d.shortcut.1", s, bi);
        fold_scale(b, nm, c, scale_sc, pc.bias_sc, err);
        if (!err.empty()) return false;
        pack_conv(pc, w, scale, wsc, scale_sc, m.rounding);
      }
      if (!make(B.conv2, "conv2.weight", "bn2", c, 1, false)) return false;
    }
    cin = c;
    if (s >= 1) {
      Head &H = m.heads[s - 1];
      H.classes = m.arch == 0 ? cls_ctu[s - 1] : cls_cu[s - 1];
      H.c = c;
      std::snprintf(nm, sizeof nm, "branch%d.weight", s);
      const float *w = b.find(nm, (uint64_t)H.classes * (c + 2), err);
      std::snprintf(nm, sizeof nm, "branch%d.bias", s);
      const float *bb = b.find(nm, H.classes, err);
      if (!w || !bb) return false;
      H.w.assign(w, w + (size_t)H.classes * (c + 2));
      H.b.assign(bb, bb + H.classes);
      m.n_logits += H.classes;
    }
  }
  return true;
}

}  // namespace mlt
