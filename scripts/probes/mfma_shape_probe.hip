// Round 6: is v_mfma_f32_16x16x32_f16 worth a port of the MFMA loops?  Every MFMA in the tree is v_mfma_f32_32x32x16_f16.
// MI355X_MICROARCH.md ("DVFS give-back", item 7) measured the 16x16x32 shape at 1.12-1.15 x the FLOP/s of the 32x32x16 shape in power-limited
// loops at equal cycles per FLOP.  This probe prices both shapes in the two loop forms the product has, same FLOPs and same LDS reads per FLOP:
//   R  "streaming" (layer0_stream_kernel / layer1_stream_kernel): 18 A fragments (a 32 x 32-channel 3x3 conv) resident in registers, one wave
//      = 32 output channels x 32 pixels, ONE activation fragment read from LDS (ds_read_b128, pixel pitch 80 B) per 32 kFLOP:
//        32x32x16: 18 items of (1 read, 1 MFMA), one accumulator chain;  16x16x32: 9 items of (2 reads, 4 MFMAs), four chains
//   L  "chain" (chain_kernel): weights AND activations from LDS, one wave = 64 output channels x 64 pixels, per 64 input channels of a tap
//        32x32x16: 4 k-steps of (2 A + 2 B reads, 4 MFMAs);  16x16x32: 2 k-steps of (4 A + 4 B reads, 16 MFMAs)
// on random post-ReLU activations (half of them zero) and random weights, fragment reads from inline asm one item ahead with counted lgkmcnt (the
// product's idiom), optionally one barrier per iteration (R: the streaming kernels' step).  Protocol of the guide: >= 2 s of back-to-back launches,
// then wall time (HIP events) AND in-kernel cycles (s_memtime) + clock (s_memtime / s_memrealtime) of the timed launches.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_shape_probe.bin mfma_shape_probe.hip && ./mfma_shape_probe.bin
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef float float4v __attribute__((ext_vector_type(4)));

template <int... I, class F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F &&f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) { static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F &&>(f)); }
template <int OFF> __device__ __forceinline__ void lds_read128(half8 &dst, uint32_t addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory"); }
template <int N> __device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void lds_touch(half8 &v) { asm volatile("" : "+v"(v)); }

__device__ __forceinline__ uint32_t hash32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }
// post-ReLU-like activation: zero with probability 1/2, else uniform in (0, 2)
__device__ __forceinline__ _Float16 act_of(uint32_t i) { const uint32_t h = hash32(i * 2654435761u + 12345u); return (h & 1) ? (_Float16)0 : (_Float16)((float)((h >> 8) & 0xFFFF) * (2.0f / 65536.0f)); }
__device__ __forceinline__ _Float16 wgt_of(uint32_t i) { const uint32_t h = hash32(i * 40503u + 977u); return (_Float16)(((float)((h >> 8) & 0xFFFF) - 32768.0f) * (0.05f / 32768.0f)); }

struct Stamps { unsigned long long cyc, real; };

// ---- R: weights in registers ------------------------------------------------------------------------------------------------------------
template <int SHAPE, bool BAR>
__global__ __launch_bounds__(1024) void probe_r(float *out, Stamps *st, int iters) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (uint32_t o = threadIdx.x; o < 160 * 1024 / 2; o += blockDim.x) ((_Float16 *)smem)[o] = act_of(o + blockIdx.x * 7919u);
  half8 a[18];
#pragma unroll
  for (int i = 0; i < 18; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) a[i][j] = wgt_of((uint32_t)(i * 512 + lane * 8 + j + wave * 9216));
  __syncthreads();
  // three rows of 66 pixels x 80 B per wave group (the streaming kernels' ring rows); a wave's 32 pixels start at its half of the row
  const uint32_t row = (uint32_t)(uintptr_t)smem + (uint32_t)((wave >> 1) % 6) * 3 * 5280u + (uint32_t)(wave & 1) * 32u * 80u;
  unsigned long long c0 = 0, r0 = 0;
  if (lane == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  float s = 0.f;
  if constexpr (SHAPE == 32) {
    // lane (p = l & 31, h = l >> 5): B[k = 8 h + j][pixel p] -> 16 bytes at pixel p, channel block 2 ks + h of tap t
    const uint32_t base = row + (uint32_t)(lane & 31) * 80u + (uint32_t)(lane >> 5) * 16u;
    float16v acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    half8 f[3];
    for (int it = 0; it < iters; ++it) {
      // item i = (tap t = i / 2, k-step ks = i & 1): row t / 3, pixel shift t % 3, channels 16 ks ..
      lds_read128<0>(f[0], base);
      static_for<18>([&](auto ic) {
        constexpr int i = decltype(ic)::value;
        if constexpr (i + 1 < 18) { constexpr int n = i + 1, t = n / 2, ks = n & 1; lds_read128<(t / 3) * 5280 + (t % 3) * 80 + ks * 32>(f[n % 3], base); lds_wait<1>(); }
        else lds_wait<0>();
        lds_touch(f[i % 3]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], f[i % 3], acc, 0, 0, 0);
      });
      if (BAR) __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
  } else {
    // lane (p = l & 15, g = l >> 4): B[k = 8 g + j][pixel p] -> 16 bytes at pixel p, channel block g of tap t; two pixel halves per tap
    const uint32_t base = row + (uint32_t)(lane & 15) * 80u + (uint32_t)(lane >> 4) * 16u;
    float4v acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[c][i] = 0.f;
    half8 f[2][2];
    for (int it = 0; it < iters; ++it) {
      lds_read128<0>(f[0][0], base);
      lds_read128<16 * 80>(f[0][1], base);
      static_for<9>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if constexpr (t + 1 < 9) {
          constexpr int n = t + 1;
          lds_read128<(n / 3) * 5280 + (n % 3) * 80>(f[n & 1][0], base);
          lds_read128<(n / 3) * 5280 + (n % 3) * 80 + 16 * 80>(f[n & 1][1], base);
          lds_wait<2>();
        } else lds_wait<0>();
        lds_touch(f[t & 1][0]); lds_touch(f[t & 1][1]);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2 * t], f[t & 1][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2 * t + 1], f[t & 1][0], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2 * t], f[t & 1][1], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[2 * t + 1], f[t & 1][1], acc[3], 0, 0, 0);
      });
      if (BAR) __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 4; ++i) s += acc[c][i];
  }
  if (lane == 0) { st[blockIdx.x * (blockDim.x >> 6) + wave] = Stamps{__builtin_amdgcn_s_memtime() - c0, __builtin_amdgcn_s_memrealtime() - r0}; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// ---- L: weights and activations from LDS (wave tile 64 couts x 64 pixels, 64 input channels of 3 taps per iteration) ------------------------
template <int SHAPE>
__global__ __launch_bounds__(512) void probe_l(float *out, Stamps *st, int iters) {
  extern __shared__ char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // [0, 48 KiB): weight ring step = 3 taps x 64 cin x 128 couts, A-fragment order (1 KiB fragments, lane-linear); [48 KiB, ..): 270 pixels x 144 B
  for (uint32_t o = threadIdx.x; o < 48 * 1024 / 2; o += blockDim.x) ((_Float16 *)smem)[o] = wgt_of(o + blockIdx.x * 104729u);
  for (uint32_t o = threadIdx.x; o < 40 * 1024 / 2; o += blockDim.x) ((_Float16 *)(smem + 48 * 1024))[o] = act_of(o + blockIdx.x * 7919u);
  __syncthreads();
  const uint32_t wbase = (uint32_t)(uintptr_t)smem + (uint32_t)(wave & 1) * 24u * 1024u + (uint32_t)lane * 16u;   // this wave's 64 couts: 24 fragments of 1 KiB
  const uint32_t xrow = (uint32_t)(uintptr_t)smem + 48u * 1024u + (uint32_t)(wave >> 1) * 64u * 144u;             // this wave's 64 pixels
  unsigned long long c0 = 0, r0 = 0;
  if (lane == 0) { c0 = __builtin_amdgcn_s_memtime(); r0 = __builtin_amdgcn_s_memrealtime(); }
  float s = 0.f;
  if constexpr (SHAPE == 32) {
    const uint32_t xb = xrow + (uint32_t)(lane & 31) * 144u + (uint32_t)(lane >> 5) * 16u;
    float16v acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    half8 fa[2][2], fb[2][2];
    for (int it = 0; it < iters; ++it) {
      // 12 k-steps: (tap t = k / 4, ks = k & 3); A fragments 2 k, 2 k + 1 of the wave's 24; B: pixel blocks 0 / 32, tap shift 144 t, channels 16 ks ..
      lds_read128<0>(fa[0][0], wbase); lds_read128<1024>(fa[0][1], wbase);
      lds_read128<0>(fb[0][0], xb); lds_read128<32 * 144>(fb[0][1], xb);
      static_for<12>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k + 1 < 12) {
          constexpr int n = k + 1, t = n / 4, ks = n & 3;
          lds_read128<2 * n * 1024>(fa[n & 1][0], wbase); lds_read128<(2 * n + 1) * 1024>(fa[n & 1][1], wbase);
          lds_read128<t * 144 + ks * 32>(fb[n & 1][0], xb); lds_read128<t * 144 + ks * 32 + 32 * 144>(fb[n & 1][1], xb);
          lds_wait<4>();
        } else lds_wait<0>();
        lds_touch(fa[k & 1][0]); lds_touch(fa[k & 1][1]); lds_touch(fb[k & 1][0]); lds_touch(fb[k & 1][1]);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[k & 1][0], fb[k & 1][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[k & 1][1], fb[k & 1][0], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[k & 1][0], fb[k & 1][1], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[k & 1][1], fb[k & 1][1], acc[3], 0, 0, 0);
      });
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int i = 0; i < 16; ++i) s += acc[c][i];
  } else {
    const uint32_t xb = xrow + (uint32_t)(lane & 15) * 144u + (uint32_t)(lane >> 4) * 16u;
    float4v acc[4][4];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[c][p][i] = 0.f;
    half8 fa[2][4], fb[2][4];
    for (int it = 0; it < iters; ++it) {
      // 6 k-steps of 32 channels: (tap t = k / 2, ks = k & 1); A fragments 4 k .. 4 k + 3; B: pixel blocks 0 / 16 / 32 / 48, channels 32 ks ..
      static_for<4>([&](auto jc) { constexpr int j = decltype(jc)::value; lds_read128<j * 1024>(fa[0][j], wbase); lds_read128<j * 16 * 144>(fb[0][j], xb); });
      static_for<6>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k + 1 < 6) {
          constexpr int n = k + 1, t = n / 2, ks = n & 1;
          static_for<4>([&](auto jc) { constexpr int j = decltype(jc)::value; lds_read128<(4 * n + j) * 1024>(fa[n & 1][j], wbase); lds_read128<t * 144 + ks * 64 + j * 16 * 144>(fb[n & 1][j], xb); });
          lds_wait<8>();
        } else lds_wait<0>();
        static_for<4>([&](auto jc) { constexpr int j = decltype(jc)::value; lds_touch(fa[k & 1][j]); lds_touch(fb[k & 1][j]); });
        static_for<4>([&](auto pc) {
          constexpr int p = decltype(pc)::value;
          static_for<4>([&](auto cc) { constexpr int c = decltype(cc)::value; acc[c][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[k & 1][c], fb[k & 1][p], acc[c][p], 0, 0, 0); });
        });
      });
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[c][p][i];
  }
  if (lane == 0) { st[blockIdx.x * (blockDim.x >> 6) + wave] = Stamps{__builtin_amdgcn_s_memtime() - c0, __builtin_amdgcn_s_memrealtime() - r0}; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K> static void run(const char *name, K kern, int threads, int lds, double flop_per_wave_iter, float *d, Stamps *dst, double soak_s) {
  hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const int iters = 4000, nblk = 256, waves = threads / 64;
  hipLaunchKernelGGL(kern, dim3(nblk), dim3(threads), lds, 0, d, dst, 50);
  hipDeviceSynchronize();
  // >= soak_s of back-to-back launches on random data (the clock the chip holds under THIS loop), then 8 timed launches
  const auto t0 = std::chrono::steady_clock::now();
  int warm = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < soak_s) {
    for (int k = 0; k < 4; ++k) hipLaunchKernelGGL(kern, dim3(nblk), dim3(threads), lds, 0, d, dst, iters);
    hipDeviceSynchronize();
    warm += 4;
  }
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int reps = 8;
  hipEventRecord(e0);
  for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(kern, dim3(nblk), dim3(threads), lds, 0, d, dst, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  std::vector<Stamps> h((size_t)nblk * waves);
  hipMemcpy(h.data(), dst, h.size() * sizeof(Stamps), hipMemcpyDeviceToHost);
  std::vector<double> cyc, clk;
  for (const Stamps &s : h) { cyc.push_back((double)s.cyc); clk.push_back(s.real ? (double)s.cyc / (double)s.real * 0.1 : 0.0); }   // s_memrealtime: 100 MHz
  std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
  const double flop = (double)nblk * waves * iters * flop_per_wave_iter;
  const double mcyc = cyc[cyc.size() / 2];
  printf("%-46s waves/SIMD %d: %8.3f ms %8.1f TFLOP/s | median wave %10.0f cycles = %6.2f cycles per 32 kFLOP and wave, clock %5.3f GHz (soak %d launches)\n", name, waves / 4, ms,
         flop / ms * 1e-9, mcyc, mcyc / (iters * flop_per_wave_iter / 32768.0), clk[clk.size() / 2], warm);
  fflush(stdout);
}

int main(int argc, char **argv) {
  const double soak = argc > 1 ? atof(argv[1]) : 2.0;
  float *d; Stamps *st;
  hipMalloc(&d, 256 * 1024 * 4);
  hipMalloc(&st, 256 * 16 * sizeof(Stamps));
  const double fr = 18.0 * 32768.0, fl = 48.0 * 32768.0;
  for (int t : {256, 512, 1024}) {
    run("R 32x32x16 (weights in registers, 1 read / MFMA)", probe_r<32, false>, t, 160 * 1024, fr, d, st, soak);
    run("R 16x16x32 (weights in registers, 1 read / 2 MFMA)", probe_r<16, false>, t, 160 * 1024, fr, d, st, soak);
  }
  run("R 32x32x16 + barrier per 18 MFMAs", probe_r<32, true>, 1024, 160 * 1024, fr, d, st, soak);
  run("R 16x16x32 + barrier per 36 MFMAs", probe_r<16, true>, 1024, 160 * 1024, fr, d, st, soak);
  for (int t : {256, 512}) {
    run("L 32x32x16 (weights + activations from LDS)", probe_l<32>, t, 160 * 1024, fl, d, st, soak);
    run("L 16x16x32 (weights + activations from LDS)", probe_l<16>, t, 160 * 1024, fl, d, st, soak);
  }
  // and back again (drift of the box over the run)
  run("R 32x32x16 (repeat)", probe_r<32, false>, 1024, 160 * 1024, fr, d, st, soak);
  run("R 16x16x32 (repeat)", probe_r<16, false>, 1024, 160 * 1024, fr, d, st, soak);
  run("L 32x32x16 (repeat)", probe_l<32>, 512, 160 * 1024, fl, d, st, soak);
  run("L 16x16x32 (repeat)", probe_l<16>, 512, 160 * 1024, fl, d, st, soak);
  return 0;
}
