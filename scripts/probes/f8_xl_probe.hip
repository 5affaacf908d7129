// Round 5: the exact-lite cross-term MFMA in isolation.  32 x 32 outputs, one (tap, 32-channel chunk): D = sum_c Wl[r][c] Xh[p][c] + Wh[r][c] Xl[p][c]
// through v_mfma_scale_f32_32x32x64_f8f6f4 with A = [e4m3(Wl 2^ewl) | e4m3(Wh 2^ewh)], B = [e4m3(Xh) | e4m3(Xl 2^12)] and per-lane block scales,
// against the fp64 sum of the SAME rounded operands (tests the instruction and the scale plumbing) and of the unrounded ones (tests the precision).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
typedef int int8v __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
__global__ void probe(const uint8_t *A, const uint8_t *B, float *D, int sa0, int sa1, int sb1) {
  const int lane = threadIdx.x, h = lane >> 5;
  int8v a = *(const int8v *)(A + lane * 32), b = *(const int8v *)(B + lane * 32);
  float16v c;
  for (int i = 0; i < 16; ++i) c[i] = 0.f;
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, h ? sa1 : sa0, 0, h ? sb1 : 0x7F7F7F7F);
  for (int i = 0; i < 16; ++i) D[lane * 16 + i] = c[i];
}
static uint8_t enc(float f) {  // e4m3fn, round to nearest even, saturating (mlt_model.cpp: f32_to_e4m3)
  uint8_t sign = f < 0 ? 0x80 : 0; float a = std::fabs(f);
  if (a >= 448.f) return sign | 0x7E;
  if (a < 0.0009765625f) return sign;
  int e; float m = std::frexp(a, &e); int ex = e - 1;
  if (ex < -6) { int q = (int)std::nearbyint(a * 512.f); return sign | (q >= 8 ? 0x08 : q); }
  int q = (int)std::nearbyint((2.f * m - 1.f) * 8.f);
  if (q == 8) { q = 0; ++ex; }
  if (ex > 8 || (ex == 8 && q > 6)) return sign | 0x7E;
  return sign | ((ex + 7) << 3) | q;
}
static double dec(uint8_t b) { int s = b >> 7, e = (b >> 3) & 15, m = b & 7; double v = e == 0 ? m / 512.0 : (1.0 + m / 8.0) * std::ldexp(1.0, e - 7); return s ? -v : v; }
int main() {
  srand(5);
  auto rnd = []() { double u = 0; for (int i = 0; i < 12; ++i) u += rand() / (double)RAND_MAX; return u - 6.0; };
  const int C = 32;
  std::vector<double> W(32 * C), X(32 * C), Wh(32 * C), Wl(32 * C), Xh(32 * C), Xl(32 * C);
  double wlmax = 0, whmax = 0;
  for (int i = 0; i < 32 * C; ++i) {
    W[i] = rnd() * 0.06 * 64; X[i] = std::fabs(rnd() * 0.6 + 0.5);
    Wh[i] = (double)(_Float16)W[i]; Wl[i] = (double)(_Float16)(W[i] - Wh[i]); Xh[i] = (double)(_Float16)X[i]; Xl[i] = (double)(_Float16)(X[i] - Xh[i]);
    wlmax = std::max(wlmax, std::fabs(Wl[i])); whmax = std::max(whmax, std::fabs(Wh[i]));
  }
  const int ewl = (int)std::floor(std::log2(224.0 / wlmax)), ewh = (int)std::floor(std::log2(224.0 / whmax));
  std::vector<uint8_t> A(64 * 32), B(64 * 32);
  for (int r = 0; r < 32; ++r)
    for (int c = 0; c < C; ++c) {
      // (scripts/probes/f8_mfma_scale_probe2.hip: the scale of lane r covers bytes 0-15 of lanes r AND r + 32 -- K block 0 -- the scale of lane r + 32 bytes 16-31 of both)
      const int hh = c >> 4, b = c & 15;
      A[(hh * 32 + r) * 32 + b] = enc((float)(Wl[r * C + c] * std::ldexp(1.0, ewl)));
      A[(hh * 32 + r) * 32 + 16 + b] = enc((float)(Wh[r * C + c] * std::ldexp(1.0, ewh)));
      B[(hh * 32 + r) * 32 + b] = enc((float)Xh[r * C + c]);
      B[(hh * 32 + r) * 32 + 16 + b] = enc((float)(Xl[r * C + c] * 4096.0));
    }
  for (int mode = 0; mode < 3; ++mode) {   // 0: both K blocks, 1: block 0 only (Wl Xh), 2: block 1 only (Wh Xl)
  std::vector<uint8_t> A0 = A, B0 = B;
  if (mode == 1) for (int l = 0; l < 64; ++l) for (int c = 16; c < 32; ++c) { A[l * 32 + c] = 0; }
  if (mode == 2) for (int l = 0; l < 64; ++l) for (int c = 0; c < 16; ++c) { A[l * 32 + c] = 0; }
  uint8_t *dA, *dB; float *dD;
  hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 4096);
  hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
  const int sa0 = 0x01010101 * ((127 - ewl) & 0xFF), sa1 = 0x01010101 * ((127 - ewh) & 0xFF), sb1 = 0x01010101 * (127 - 12);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, sa0, sa1, sb1);
  std::vector<float> D(1024);
  hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
  double e_same = 0, e_true = 0, mag = 0;
  for (int l = 0; l < 64; ++l)
    for (int i = 0; i < 16; ++i) {
      const int col = l & 31, row = (i & 3) + 8 * (i >> 2) + 4 * (l >> 5);
      double same = 0, tru = 0;
      for (int c = 0; c < C; ++c) {
        const int hh = c >> 4, b = c & 15;
        if (mode != 2) { same += dec(A[(hh * 32 + row) * 32 + b]) * std::ldexp(1.0, -ewl) * dec(B[(hh * 32 + col) * 32 + b]); tru += Wl[row * C + c] * Xh[col * C + c]; }
        if (mode != 1) { same += dec(A[(hh * 32 + row) * 32 + 16 + b]) * std::ldexp(1.0, -ewh) * dec(B[(hh * 32 + col) * 32 + 16 + b]) / 4096.0; tru += Wh[row * C + c] * Xl[col * C + c]; }
      }
      e_same += (D[l * 16 + i] - same) * (D[l * 16 + i] - same); e_true += (D[l * 16 + i] - tru) * (D[l * 16 + i] - tru); mag += tru * tru;
    }
  printf("ewl %d ewh %d  rms cross term %.3e  rms(D - sum of the same rounded operands) %.3e  rms(D - true cross terms) %.3e  = %.1f %% of the cross terms\n",
         ewl, ewh, std::sqrt(mag / 1024), std::sqrt(e_same / 1024), std::sqrt(e_true / 1024), 100 * std::sqrt(e_true / mag));
  printf("mode %d: D[0..3] %g %g %g %g\n", mode, D[0], D[1], D[2], D[3]);
  A = A0; B = B0;
  }
  return 0;
}
