// Round 5: how many MFMA chains does a SIMD need?  Waves that each run DEPENDENT v_mfma_f32_32x32x16_f16 chains (one accumulator: the shape of a
// conv with 32 output channels) against waves with two independent accumulators, at 1 / 2 / 4 waves per SIMD; optionally with one ds_read_b128 per
// MFMA (the activation fragment) and with a VALU epilogue + barrier every 18 MFMAs (layer0_stream_kernel's step).  Prints TFLOP/s of the whole chip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
template <int CHAINS, bool LDS, bool EPI>
__global__ void probe(float *out, int iters) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63;
  half8 a[18];
  for (int i = 0; i < 18; ++i) for (int j = 0; j < 8; ++j) a[i][j] = (_Float16)(0.001f * (i + j + lane));
  for (int o = threadIdx.x * 16; o < 64 * 1024; o += blockDim.x * 16) *(float4 *)(smem + o) = float4{1e-3f, 2e-3f, 3e-3f, 4e-3f};
  __syncthreads();
  float16v acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
  half8 b = a[3];
  const char *row = smem + (threadIdx.x >> 6) * 2048 + (lane & 31) * 80 + (lane >> 5) * 16;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 18; ++i) {
      if (LDS) b = *(const half8 *)(row + (i % 6) * 160 + (i / 6) * 5280);
      acc[i % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b, acc[i % CHAINS], 0, 0, 0);
    }
    if (EPI) {
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < CHAINS; ++c)
#pragma unroll
        for (int i = 0; i < 16; ++i) { s += fmaxf(acc[c][i] + 0.5f, 0.f); acc[c][i] = 0.f; }
      *(float *)(smem + 32768 + threadIdx.x * 4) = s;
      __syncthreads();
    }
  }
  float s = 0.f;
  for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int CHAINS, bool LDS, bool EPI> static void run(const char *name, int threads, float *d) {
  auto k = probe<CHAINS, LDS, EPI>;
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k, dim3(256), dim3(threads), 64 * 1024, 0, d, 50);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(256), dim3(threads), 64 * 1024, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = 256.0 * (threads / 64) * iters * 18 * 32768.0;
  printf("%-44s waves/SIMD %d: %7.3f ms  %7.1f TFLOP/s  (%.1f cycles per MFMA and SIMD at 2.1 GHz)\n", name, threads / 256, ms, flop / ms * 1e-9,
         ms * 1e-3 * 2.1e9 / (iters * 18.0 * (threads / 256)));
}
int main() {
  float *d;
  hipMalloc(&d, 256 * 1024 * 4);
  for (int t : {256, 512, 1024}) {
    run<1, false, false>("1 chain, registers only", t, d);
    run<2, false, false>("2 chains, registers only", t, d);
    run<1, true, false>("1 chain, one ds_read_b128 per MFMA", t, d);
    run<2, true, false>("2 chains, one ds_read_b128 per MFMA", t, d);
    run<1, true, true>("1 chain, reads, epilogue + barrier per 18", t, d);
    if (t <= 512) run<2, true, true>("2 chains, reads, epilogue + barrier per 18", t, d);
  }
  return 0;
}
