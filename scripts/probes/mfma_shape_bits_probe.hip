// Round 6: does v_mfma_f32_16x16x32_f16 give the BITS of two chained v_mfma_f32_32x32x16_f16 (k 0-15, then k 16-31) on the same matrices?
// If it did, the MFMA loops could move to the faster shape (profiles/r06a_mfma_shape_probe.txt: +7-11 % under the power cap) kernel by kernel while
// every entry point stays bit-identical to the others; if not, every form of a conv (streaming, tiled, per-conv latency variant) has to move together.
// Random fp16 operands (post-ReLU-like B, +-0.05 weights), accumulators started from a random fp32 C.  Prints the number of differing elements.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));
__global__ void k(const _Float16 *A, const _Float16 *B, const float *C, float *D32, float *D16, int chain) {
  // A [32][K], B [K][32] row-major with K = 32 * chain; one wave
  const int l = threadIdx.x, K = 32 * chain;
  float16v acc;
  for (int i = 0; i < 16; ++i) acc[i] = C[((i & 3) + 8 * (i >> 2) + 4 * (l >> 5)) * 32 + (l & 31)];
  for (int s = 0; s < 2 * chain; ++s) {
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = A[(l & 31) * K + 16 * s + 8 * (l >> 5) + j]; b[j] = B[(16 * s + 8 * (l >> 5) + j) * 32 + (l & 31)]; }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) D32[((i & 3) + 8 * (i >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = acc[i];
  for (int ti = 0; ti < 2; ++ti)
    for (int tj = 0; tj < 2; ++tj) {
      float4v c;
      for (int i = 0; i < 4; ++i) c[i] = C[(16 * ti + 4 * (l >> 4) + i) * 32 + 16 * tj + (l & 15)];
      for (int s = 0; s < chain; ++s) {
        half8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = A[(16 * ti + (l & 15)) * K + 32 * s + 8 * (l >> 4) + j]; b[j] = B[(32 * s + 8 * (l >> 4) + j) * 32 + 16 * tj + (l & 15)]; }
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
      }
      for (int i = 0; i < 4; ++i) D16[(16 * ti + 4 * (l >> 4) + i) * 32 + 16 * tj + (l & 15)] = c[i];
    }
}
int main() {
  uint64_t z = 12345;
  auto rnd = [&]() { z = z * 6364136223846793005ull + 1442695040888963407ull; return (double)(z >> 11) / 9007199254740992.0; };
  for (int chain : {1, 9}) {
    const int K = 32 * chain, trials = 2000;
    long diff = 0, total = 0; double maxrel = 0;
    _Float16 *dA, *dB; float *dC, *d32, *d16;
    hipMalloc(&dA, 32 * K * 2); hipMalloc(&dB, K * 32 * 2); hipMalloc(&dC, 4096); hipMalloc(&d32, 4096); hipMalloc(&d16, 4096);
    std::vector<_Float16> A(32 * K), B(K * 32); std::vector<float> C(1024), r32(1024), r16(1024);
    for (int t = 0; t < trials; ++t) {
      for (auto &v : A) v = (_Float16)((rnd() - 0.5) * 0.1);
      for (auto &v : B) v = (_Float16)(rnd() < 0.5 ? 0.0 : rnd() * 2.0);
      for (auto &v : C) v = (float)((rnd() - 0.5) * 4.0);
      hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC, d32, d16, chain);
      hipMemcpy(r32.data(), d32, 4096, hipMemcpyDeviceToHost); hipMemcpy(r16.data(), d16, 4096, hipMemcpyDeviceToHost);
      for (int i = 0; i < 1024; ++i) { ++total; if (memcmp(&r32[i], &r16[i], 4)) { ++diff; double r = fabs((double)r32[i] - r16[i]) / (fabs((double)r32[i]) + 1e-30); if (r > maxrel) maxrel = r; } }
    }
    printf("K = %3d (%d x 16x16x32 against %d x 32x32x16 chained): %ld of %ld output elements differ in their bits (largest relative difference %.2e)\n", K, chain, 2 * chain, diff, total, maxrel);
  }
  return 0;
}
