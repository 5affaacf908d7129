// mlt_kernels.hip -- hand-written gfx950 (CDNA4) kernels of the MLT-CNN split predictor.
//
// Arithmetic spec: /root/reference/mlt-cnn-python/codes/models/archs/mlt_ctu_or_pq_arch.py:32-57,273-299
// and mlt_cu_or_pq_arch.py:96-128; preprocessing: vtm-mlt-cpp/source/Lib/EncoderLib/EncCu.cpp:810-877.
//
// Data layout: activations NHWC fp16 in HBM; weights BN-folded, fp16, pre-packed on the host in
// MFMA A-fragment order (mlt_model.cpp) so a wave fetches one fragment as 64 x 16 contiguous bytes
// (LDS-DMA friendly, conflict-free ds_read_b128).  Accumulation, bias, residual add, global average
// pooling and the heads are fp32.
//
// MFMA orientation: D[cout][pixel] = sum_k W[cout][k] * X[k][pixel]  (weights = A, activations = B).
// With v_mfma_f32_32x32x16_f16 lane l (p = l&31, h = l>>5) supplies B[k = 8h+j][col p] = 8 consecutive
// input channels of pixel p -> ONE ds_read_b128 from the [pixel][cin] LDS patch, and receives
// D rows (i&3) + 8*(i>>2) + 4h of column p -> 4 consecutive output channels per register quad
// -> packed 8-byte NHWC stores.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

#include <atomic>
#include <utility>

#include "mlt_kernels.h"

// build-time tuning knobs (scripts/sweep_cfg.py)
#ifndef CFG_ASM_PIPE  // 1: fragment ds_reads issued from inline asm, 2 items ahead, with counted lgkmcnt waits (fast arithmetic)
#define CFG_ASM_PIPE 1
#endif
#ifndef CFG_S2_BIAS_EARLY  // ... of the stride-2 ring kernels (two bias sets = 64 VGPRs)
#define CFG_S2_BIAS_EARLY 1
#endif
#ifndef CFG_DMA_SPREAD  // DMA mode: issue one patch piece every N fragment items of the MFMA loop (0: all before the loop)
#define CFG_DMA_SPREAD 5
#endif
#ifndef CFG_RING_FD16  // fragment prefetch distance of the 16-wave ring kernels (1 or 2 items): same speed, 16 VGPRs fewer at 1
#define CFG_RING_FD16 1
#endif
#ifndef CFG_RING_RES_EARLY
#define CFG_RING_RES_EARLY 1
#endif
#ifndef CFG_BIAS_EARLY  // 1: bias loads before the MFMA phase (latency hidden, +16..32 VGPRs); 0: at the epilogue
#define CFG_BIAS_EARLY 1
#endif

// -DMLT_PHASE_TIMING: wave 0 of every conv_mfma_kernel workgroup accumulates s_memtime deltas per phase into g_phase
// (debug builds only, scripts/phase_timing.py; perturbs the pipelining slightly)
#ifdef MLT_PHASE_TIMING
__device__ unsigned long long g_phase[16][8];
__device__ unsigned long long g_phase_chain[4][16];  // chain_kernel: [C == 256][S2] x 16 phases (see scripts/phase_timing.py)
extern "C" __attribute__((visibility("default"))) int mlt_debug_phase_read(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(g_phase)) != hipSuccess) return 1;
  if (reset) {
    static unsigned long long z[16][8];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
__device__ unsigned long long g_phase_l0[4][8];      // layer0_stream_kernel: stage x {top, reads + MFMA, epilogue, step end}
extern "C" __attribute__((visibility("default"))) int mlt_debug_phase_read_l0(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_l0), sizeof(g_phase_l0)) != hipSuccess) return 1;
  if (reset) {
    static unsigned long long z[4][8];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_l0), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
extern "C" __attribute__((visibility("default"))) int mlt_debug_phase_read_chain(unsigned long long *out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase_chain), sizeof(g_phase_chain)) != hipSuccess) return 1;
  if (reset) {
    static unsigned long long z[4][16];
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_chain), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
// (one asm statement per stamp: s_memtime counts on lgkmcnt and returns out of order with LDS reads, so it must not be in flight
// inside the sections with counted lgkmcnt waits)
#define PHC_STAMP(t_) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory")
#define PHC_DECL unsigned int phc_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long phc_t; PHC_STAMP(phc_t)
#define PHC_MARK(i) do { unsigned long long n_; PHC_STAMP(n_); phc_acc[i] += (unsigned int)(n_ - phc_t); phc_t = n_; } while (0)
#define PHC_FLUSH(id) do { if (threadIdx.x == 0) { for (int i_ = 0; i_ < 16; ++i_) atomicAdd(&g_phase_chain[id][i_], (unsigned long long)phc_acc[i_]); } } while (0)
#define PH_DECL unsigned long long ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_t = __builtin_readcyclecounter()
#define PH_MARK(i) do { unsigned long long n_ = __builtin_readcyclecounter(); ph_acc[i] += n_ - ph_t; ph_t = n_; } while (0)
#define PH_FLUSH(id) do { if (threadIdx.x == 0) { for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_phase[id][i_], ph_acc[i_]); } } while (0)
#else
#define PH_DECL
#define PH_MARK(i)
#define PH_FLUSH(id)
#define PHC_DECL
#define PHC_MARK(i)
#define PHC_FLUSH(id)
#endif

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

// compile-time loop: f(std::integral_constant<int, 0>) ... f(std::integral_constant<int, N-1>), so that the index can
// feed asm immediates and if constexpr
template <int... I, class F> __device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F &&f) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F> __device__ __forceinline__ void static_for(F &&f) {
  static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F &&>(f));
}

// Fragment reads from inline asm.  While an LDS-DMA (global_load_lds) is pending the compiler makes every LDS wait an
// s_waitcnt lgkmcnt(0) and sinks the reads of the next item below the MFMAs of the current one, i.e. read -> full
// drain -> MFMA with the LDS latency exposed every item (seen in the ISA of every weight-ring kernel).  Reads issued
// here are invisible to that logic: they stay where they are written, and lds_wait<N>() + lds_touch() state exactly
// how many younger reads may still be in flight when a fragment is consumed (LDS operations of a wave return in order).
template <int OFF> __device__ __forceinline__ void lds_read128(half8 &dst, uint32_t addr) {
  static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field is 16 bits");
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int N> __device__ __forceinline__ void lds_wait() {
  static_assert(N >= 0 && N <= 15, "lgkmcnt is 4 bits");
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
__device__ __forceinline__ void lds_touch(half8 &v) { asm volatile("" : "+v"(v)); }  // orders the consumer after lds_wait

// Phase stagger for persistent kernels: identical workgroups started together stay in lock-step chip-wide, so every CU
// reaches its memory-heavy phase (epilogue residual reads + stores) at the same moment and the launch takes
// T_compute + T_memory instead of max(...).  Delaying workgroup b by (b mod phases) * units * ~4 us spreads the phases.
__device__ __forceinline__ void stagger_start(int phases, int units) {
  const int k = ((int)blockIdx.x % phases) * units;
  for (int i = 0; i < k; ++i) __builtin_amdgcn_s_sleep(127);
}

// n / d for d >= 2 with magic = ceil(2^32 / d); exact while n * d < 2^32 (patch indices are < 2^16)
__device__ __forceinline__ uint32_t udiv_magic(uint32_t n, uint32_t magic) { return magic ? __umulhi(n, magic) : n; }  // magic 0: d == 1

// (org, |org - pred|) as an exact-integer fp16 pair: EncCu.cpp:816,827 (u16 cast), :833 (absdiff),
// :848-867 (clip to [0,1] after the 1/1023 scale == clip the integer to [0,1023]; the scale itself
// lives in the stem weights).
__device__ __forceinline__ uint32_t prep_pair(int16_t so, int16_t sp) {
  uint16_t o = (uint16_t)so, q = (uint16_t)sp;
  uint16_t r = o > q ? o - q : q - o;
  o = o > 1023 ? 1023 : o;
  r = r > 1023 ? 1023 : r;
  half2v hv;
  hv[0] = (_Float16)(float)o;
  hv[1] = (_Float16)(float)r;
  return *(uint32_t *)&hv;
}


// Flat-content guard statistic of one aligned quad (mlt_kernels.h: MLT_FLAT_RANGE): w[j] = prep_pair() of its four pixels, i.e. exact
// integers <= 1023 as fp16 pairs (org, |org - pred|) -> packed min / max / add / subtract are exact (|values| <= 2046).  A plane of the
// quad is "coherent" when its four values span <= MLT_FLAT_RANGE (constant, dither, low contrast) OR are linear to within one step
// (both second differences <= 1 in magnitude: ramps of any slope -- every pixel of a perfect gradient sees the same local pattern,
// so its rounding errors are as coherent as a constant area's); the quad counts when BOTH planes are.
// Returns bit 0: near-flat, bit 1: exactly flat (each plane constant or exactly linear; implies near-flat).
__device__ __forceinline__ int quad_flat_bits(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
  const half2v a = *(half2v *)&w0, b = *(half2v *)&w1, c = *(half2v *)&w2, d = *(half2v *)&w3;
  const half2v mx = __builtin_elementwise_max(__builtin_elementwise_max(a, b), __builtin_elementwise_max(c, d));
  const half2v mn = __builtin_elementwise_min(__builtin_elementwise_min(a, b), __builtin_elementwise_min(c, d));
  const half2v r = mx - mn;
  const half2v d1 = (a + c) - (b + b), d2 = (b + d) - (c + c);
  const half2v l = __builtin_elementwise_max(__builtin_elementwise_max(d1, -d1), __builtin_elementwise_max(d2, -d2));
  const _Float16 R = (_Float16)MLT_FLAT_RANGE, one = (_Float16)1, zero = (_Float16)0;
  const bool near = (r[0] <= R || l[0] <= one) && (r[1] <= R || l[1] <= one);
  const bool exact = (r[0] == zero || l[0] == zero) && (r[1] == zero || l[1] == zero);
  return (near ? 1 : 0) | (exact ? 2 : 0);
}

// ---- FP8 lo products (round 4; hi+lo-weights forms on 64-channel chunks) ----
// The lo product of a (tap, 64-channel chunk) is ONE v_mfma_scale_f32_32x32x64_f8f6f4 (e4m3 x e4m3, twice the fp16 rate): its A operand is
// the FP8 lo plane (mlt_model.cpp: byte j of lane (r, h) = lo[cout r][cin 32 h + j]), its B operand channels 32 h .. 32 h + 31 of the lane's
// pixel from an e4m3 image of the activation (two ds_read_b128).  Lane (x, h) of A meets lane (y, h) of B byte for byte (probed with exact
// integer data: scripts/probes/f8_mfma_layout_probe.hip), so any channel order works as long as both operands use the same one.
typedef int int8v __attribute__((ext_vector_type(8)));
typedef short short2v __attribute__((ext_vector_type(2)));
// 8 fp16 activations (>= 0: every chain input is a ReLU output) -> 8 e4m3 bytes (2 dwords), clamped to e4m3's largest finite value
__device__ __forceinline__ void cvt_frag_fp8(const half8 &v, uint32_t &d0, uint32_t &d1) {
  const half2v top = {(_Float16)448, (_Float16)448};
  const half2v *h = (const half2v *)&v;
  short2v r = {0, 0};
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, __builtin_elementwise_min(h[0], top), 1.0f, false);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, __builtin_elementwise_min(h[1], top), 1.0f, true);
  d0 = *(uint32_t *)&r;
  short2v q = {0, 0};
  q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q, __builtin_elementwise_min(h[2], top), 1.0f, false);
  q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q, __builtin_elementwise_min(h[3], top), 1.0f, true);
  d1 = *(uint32_t *)&q;
}
__device__ __forceinline__ float16v mfma_lo8(const half8 &a_lo, const half8 &a_hi, const half8 &b_lo, const half8 &b_hi, const float16v &c, int scale_a) {
  int8v a, b;
  const uint32_t *pl = (const uint32_t *)&a_lo, *ph = (const uint32_t *)&a_hi, *ql = (const uint32_t *)&b_lo, *qh = (const uint32_t *)&b_hi;
#pragma unroll
  for (int e = 0; e < 4; ++e) { a[e] = (int)pl[e]; a[4 + e] = (int)ph[e]; b[e] = (int)ql[e]; b[4 + e] = (int)qh[e]; }
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, 0x7F7F7F7F);  // fp8 x fp8; A scaled by 2^-lo8_exp, B by 1
}

// ---- exact-lite (NSPLIT == 6, round 5): both cross terms of the (hi, lo) product in ONE scaled FP8 MFMA ----
// lo parts of 8 fp16 activations (signed, |lo| <= 2^-11 |x|) * 2^12 -> 8 e4m3 bytes; clamped so that neither the fp16 product nor the e4m3
// conversion overflows (activations beyond ~220 lose part of their lo term; the calibration measures the outcome like any other tier's)
#define MLT_XL_LO_EXP 12
__device__ __forceinline__ void cvt_frag_fp8_lo(const half8 &v, uint32_t &d0, uint32_t &d1) {
  const half2v top = {(_Float16)0.109375, (_Float16)0.109375}, mul = {(_Float16)4096, (_Float16)4096};  // 448 / 4096
  const half2v *h = (const half2v *)&v;
  short2v r = {0, 0}, q = {0, 0};
  auto prep = [&](half2v x) { return __builtin_elementwise_max(__builtin_elementwise_min(x, top), -top) * mul; };
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, prep(h[0]), 1.0f, false);
  r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(r, prep(h[1]), 1.0f, true);
  q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q, prep(h[2]), 1.0f, false);
  q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q, prep(h[3]), 1.0f, true);
  d0 = *(uint32_t *)&r;
  d1 = *(uint32_t *)&q;
}
// A = [e4m3 Wl | e4m3 Wh] (lanes 0-31 | 32-63), B = [e4m3 Xh ; e4m3 Xl]; scale_a / scale_b: this LANE's E8M0 byte (replicated) for its 32 K elements
__device__ __forceinline__ float16v mfma_xl8(const half8 &a_lo, const half8 &a_hi, const half8 &b_lo, const half8 &b_hi, const float16v &c, int scale_a, int scale_b) {
  int8v a, b;
  const uint32_t *pl = (const uint32_t *)&a_lo, *ph = (const uint32_t *)&a_hi, *ql = (const uint32_t *)&b_lo, *qh = (const uint32_t *)&b_hi;
#pragma unroll
  for (int e = 0; e < 4; ++e) { a[e] = (int)pl[e]; a[4 + e] = (int)ph[e]; b[e] = (int)ql[e]; b[4 + e] = (int)qh[e]; }
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
}

// ---- 16-byte epilogue I/O -------------------------------------------------------------------------------------
// After a 32x32 MFMA lane l = (p, h) holds, per register quad q, output channels 8q+4h .. 8q+4h+3 of pixel p, so
// lanes p and p+32 own the two 8-byte halves of one 16-byte span.  v_permlane32_swap exchanges the upper half-wave
// of its first operand with the lower half-wave of its second: applied to quads (q, q+1) it leaves lanes 0-31 with
// channels 8q..8q+7 and lanes 32-63 with channels 8(q+1)..8(q+1)+7 -> ONE dwordx4 access per quad pair instead of
// two dwordx2 (half the memory instructions / requests, same bytes; cdna_hip_programming.md T21).
typedef uint32_t uint4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4v pair16(half4 qa, half4 qb) {
  uint32_t ax = ((uint32_t *)&qa)[0], ay = ((uint32_t *)&qa)[1], bx = ((uint32_t *)&qb)[0], by = ((uint32_t *)&qb)[1];
  auto r0 = __builtin_amdgcn_permlane32_swap(ax, bx, false, false);
  auto r1 = __builtin_amdgcn_permlane32_swap(ay, by, false, false);
  uint4v v;
  v[0] = r0[0]; v[1] = r1[0]; v[2] = r0[1]; v[3] = r1[1];
  return v;
}
__device__ __forceinline__ void unpair16(uint4v v, half4 &qa, half4 &qb) {  // inverse of pair16 (the swap is an involution)
  auto r0 = __builtin_amdgcn_permlane32_swap(v[0], v[2], false, false);
  auto r1 = __builtin_amdgcn_permlane32_swap(v[1], v[3], false, false);
  ((uint32_t *)&qa)[0] = r0[0]; ((uint32_t *)&qa)[1] = r1[0];
  ((uint32_t *)&qb)[0] = r0[1]; ((uint32_t *)&qb)[1] = r1[1];
}

// Four accumulators -> four activated fp16 values, in packed arithmetic: v_pk_fma_f32 (acc * scale + bias), v_pk_add_f32 (+ residual),
// v_cvt_pk_f16_f32, v_pk_max_f16 -- 6 VALU ops per 4 values instead of 14 (the epilogues are VALU-bound phases in which no MFMA runs).
// The ReLU is applied AFTER the rounding: max(0, .) commutes with a monotonic rounding, so the values are those of fmaxf before it.
// x4 (optional) receives the fp32 values before the ReLU (GAP sums).
typedef float float2v __attribute__((ext_vector_type(2)));
template <int Q, class ACC = float16v> __device__ __forceinline__ half4 act_quad(const ACC &A, float scale, const float4v &b, bool has_res, const half4 &r, bool relu, float *x4 = nullptr) {
  half4 out;
#pragma unroll
  for (int ep = 0; ep < 2; ++ep) {
    float2v a2 = {A[4 * Q + 2 * ep], A[4 * Q + 2 * ep + 1]};
    const float2v b2 = {b[2 * ep], b[2 * ep + 1]};
    float2v x2 = a2 * scale + b2;
    if (has_res) {
      const float2v r2 = {(float)r[2 * ep], (float)r[2 * ep + 1]};
      x2 += r2;
    }
    if (x4) { x4[2 * ep] = x2[0]; x4[2 * ep + 1] = x2[1]; }
    half2v h2 = __builtin_convertvector(x2, half2v);
    if (relu) h2 = __builtin_elementwise_max(h2, (half2v){(_Float16)0, (_Float16)0});
    out[2 * ep] = h2[0];
    out[2 * ep + 1] = h2[1];
  }
  return out;
}

// ---- the kernels (one translation unit; split by family for readability) ----
#include "mlt_conv_kernels.inc"   // conv_epilogue, conv_mfma_kernel, conv_ring_dma_kernel
#include "mlt_chain_kernel.inc"   // chain_kernel (+ KARG)
#include "mlt_front_kernels.inc"  // block32_kernel, stem5_kernel, stem_block_kernel
#include "mlt_layer0_kernel.inc"  // layer0_stream_kernel
#include "mlt_layer1_kernel.inc"  // layer1_stream_kernel
#include "mlt_tail_kernels.inc"   // heads_kernel, flat_stat / guard kernels

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per device: remember which device ordinals a kernel has been
// configured on (a process may hold contexts on several GPUs; concurrent first launches at worst set it twice).
struct DeviceOnce {
  std::atomic<uint64_t> mask[4] = {};
  bool need(int *dev) const {
    if (hipGetDevice(dev) != hipSuccess) *dev = 0;
    return !(mask[(*dev >> 6) & 3].load(std::memory_order_acquire) & (1ull << (*dev & 63)));
  }
  void done(int dev) { mask[(dev >> 6) & 3].fetch_or(1ull << (dev & 63), std::memory_order_release); }
};
template <class K> static hipError_t ensure_big_lds(K kern, DeviceOnce &once) {
  int dev = 0;
  if (!once.need(&dev)) return hipSuccess;
  hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  if (e == hipSuccess) once.done(dev);
  return e;
}
template <int CIN, int COUT, int STRIDE, int TAPS, bool SC, int KC, int NSPLIT, int WCB, int WPB, int WAVES_C, int WAVES_P, int GT, int RB, int UN, int MINW, bool DMA, int CBP = 0, bool KMAJ = false>
static hipError_t launch_conv_t(const ConvArgs &a, int grid_x, int extra_lds, hipStream_t st) {
  auto kern = conv_mfma_kernel<CIN, COUT, STRIDE, TAPS, SC, KC, NSPLIT, WCB, WPB, WAVES_C, WAVES_P, GT, RB, UN, MINW, DMA, CBP, KMAJ>;
  constexpr int CBT = WCB * WAVES_C;
  constexpr int TT = TAPS + (SC ? 1 : 0);
  constexpr int NBUF = (TT / GT) > 1 ? RB : 1;
  const int patch_lds = (DMA ? 2 : ((NSPLIT == 2 || NSPLIT == 6) ? 2 : 1)) * a.patch_bytes;
  // (NSPLIT == 5: + the FP8 lo plane of every ring step and the e4m3 copy of the patch, 80 bytes per pixel of the fp16 patch's KC * 2 + 16)
  const int patch8 = NSPLIT == 5 ? ((a.patch_bytes / (KC * 2 + 16) + 1) * (KC + 16) + 1023) / 1024 * 1024 : 0;
  const int lds = patch_lds + NBUF * ((NSPLIT == 2 || NSPLIT == 3 || NSPLIT == 6 ? 2 : 1) * GT * (KC / 16) * CBT * 1024 + (NSPLIT == 5 ? GT * CBT * 2048 : 0)) + patch8 + extra_lds;
  static DeviceOnce once;
  if (hipError_t e = ensure_big_lds(kern, once); e != hipSuccess) return e;
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  dim3 grid(grid_x, COUT / (32 * CBT));
  hipLaunchKernelGGL(kern, grid, dim3(64 * WAVES_C * WAVES_P), lds, st, a);
  return hipGetLastError();
}

template <int CIN, int COUT, int STRIDE, bool SC, int KC, int WCB, int WPB, int WAVES_C, int WAVES_P, int GT, int RB, int UNP, int MINW, int NWL, int FD>
static hipError_t launch_ring_dma_t(const ConvArgs &a, int grid_x, hipStream_t st) {
  auto kern = conv_ring_dma_kernel<CIN, COUT, STRIDE, SC, KC, WCB, WPB, WAVES_C, WAVES_P, GT, RB, UNP, MINW, NWL, FD>;
  constexpr int CBT = WCB * WAVES_C, NW = WAVES_C * WAVES_P;
  constexpr int NWP = NWL ? NWL - NWL / 2 : NW - NW / 2;
  const int lds = 2 * a.patch_bytes + RB * GT * (KC / 16) * CBT * 1024 + (NWL ? 32 * CBT * 4 * (SC ? 2 : 1) : 0);
  static DeviceOnce once;
  if (hipError_t e = ensure_big_lds(kern, once); e != hipSuccess) return e;
  if (lds > 160 * 1024 || (a.patch_bytes >> 10) > UNP * NWP) return hipErrorInvalidValue;
  dim3 grid(grid_x, COUT / (32 * CBT));
  hipLaunchKernelGGL(kern, grid, dim3(64 * (NW + NWL)), lds, st, a);
  return hipGetLastError();
}

// Per layer shape: cin chunk (KC), wave tiling and taps per weight step.  mlt_conv_cfg() is the single
// table both the launcher and the host (packing, patch sizing) read.
// build-time tuning knobs (scripts/sweep_cfg.py): taps per weight step / ring depth of selected kernels
#ifndef CFG_3264_GT
#define CFG_3264_GT 10
#endif
#ifndef CFG_3264_RB
#define CFG_3264_RB 1
#endif
#ifndef CFG_S1_RB
#define CFG_S1_RB 2
#endif
#ifndef CFG_64_WPB   // 64@32 stride-1: pixel blocks per wave / waves along pixels
#define CFG_64_WPB 1
#endif
#ifndef CFG_64_WP
#define CFG_64_WP 8
#endif
#ifndef CFG_64_WCB
#define CFG_64_WCB 2
#endif
#ifndef CFG_64_WC
#define CFG_64_WC 1
#endif
#ifndef CFG_S2BIG_WCB
#define CFG_S2BIG_WCB 2
#endif
#ifndef CFG_S2BIG_WC
#define CFG_S2BIG_WC 2
#endif
#ifndef CFG_BIG_WPB  // 128@16, 256@8 stride-1
#define CFG_BIG_WPB 1
#endif
#ifndef CFG_BIG_WP   // 8: one 16-wave workgroup per CU on 256-pixel tiles (with 3 taps per weight step: 128@16 0.405 -> 0.389 ms,
#define CFG_BIG_WP 8 // 256@8 0.340 -> 0.317 ms vs two 8-wave workgroups on 128-pixel tiles with 1 tap per step)
#endif
#ifndef CFG_GTE_S1   // exact arithmetic: taps per weight step, stride-1 (9 taps: 1, 3) / stride-2 + shortcut (10: 1, 2, 5)
#define CFG_GTE_S1 1
#endif
#ifndef CFG_GTE_S2   // round 4 sweep (profiles/r04h_sweep_exact*.txt): 2 taps per step -> 128->256 s2 1.57 -> 1.25 ms, 32->64 s2 1.46 -> 1.33 ms (exact 128 model
#define CFG_GTE_S2 2 // +2 %), 64x64 model 958 k -> 1.00 M CU/s, 16x16 +3 %; 5 taps do not fit the LDS beside two activation planes
#endif
#ifndef CFG_UNE_32   // exact arithmetic: patch items per lane prefetched in registers (two planes each); a patch with more items per lane takes a
#define CFG_UNE_32 5  // synchronous tail at the top of the tile (the 128 model's tiles have 4.4 - 4.8 items per lane: phase stamps, commit 36 - 45 %);
                      // round 4 sweep (profiles/r04m_sweep_exact_prefetch.txt): 32 -> 32 1.57 -> 1.49 ms, 32 -> 64 s2 1.33 -> 1.27; the 64 / 128-channel
                      // stride-2 layers LOSE with 5 (1.02 -> 1.09, 0.84 -> 0.87) and keep 3
#endif
#ifndef CFG_UNE_64
#define CFG_UNE_64 3
#endif
#ifndef CFG_UNE_S2
#define CFG_UNE_S2 3
#endif
#ifndef CFG_BIG_WP_EXACT  // ... exact arithmetic (small-CU models, maps of 1..16 pixels): 128-pixel tiles, 8 waves
#define CFG_BIG_WP_EXACT 4
#endif
#ifndef CFG_BIG_WCB
#define CFG_BIG_WCB 2
#endif
#ifndef CFG_128_WPB    // 128@16 stride-1 pixel blocks per wave (256@8 uses CFG_BIG_WPB)
#define CFG_128_WPB CFG_BIG_WPB
#endif
#ifndef CFG_S2BIG_WPB  // 64->128, 128->256 stride-2 (+shortcut)
#define CFG_S2BIG_WPB 1
#endif
#ifndef CFG_S2BIG_WP
#define CFG_S2BIG_WP 4
#endif
#ifndef CFG_3264_WP    // 32->64 stride-2 (+shortcut): waves along pixels (tile = 32 * WP pixels) and patch items per lane in registers
#define CFG_3264_WP 4
#endif
#ifndef CFG_3264_UN
#define CFG_3264_UN 5
#endif
#ifndef CFG_3264_WCB   // 32->64 stride-2 (+shortcut)
#define CFG_3264_WCB 1
#endif
#ifndef CFG_3264_WC
#define CFG_3264_WC 2
#endif
#ifndef CFG_32_WPB     // 32@64 stride-1
#define CFG_32_WPB 2
#endif
#ifndef CFG_32_WP
#define CFG_32_WP 8
#endif
#ifndef CFG_64_GT      // taps per weight step, 64@32 stride-1 (9 = all weights resident in LDS)
#define CFG_64_GT 9
#endif
#ifndef CFG_BIG_GT     // 128@16 / 256@8 stride-1: taps per weight step (48 KiB steps, 6 barriers per 64-channel chunk instead of 18)
#define CFG_BIG_GT 3
#endif
#ifndef CFG_S1_MINW   // min waves / SIMD of the 64..256-channel stride-1 kernels (VGPR cap)
#define CFG_S1_MINW 1
#endif
#ifndef CFG_S2_MINW
#define CFG_S2_MINW 1
#endif
#ifndef CFG_32_MINW
#define CFG_32_MINW 1
#endif
#ifndef CFG_BIG_WC
#define CFG_BIG_WC 2
#endif
#ifndef CFG_64_DMA     // 64@32 stride-1: LDS-DMA double-buffered patch staging (fast arithmetic, maps >= 16 x 16)
#define CFG_64_DMA 1
#endif
#ifndef CFG_64_DMA_UN  // LDS-DMA instructions per wave and tile: 10 x 34 pixels x 8 slots / (8 waves x 64 lanes) = 5.3
#define CFG_64_DMA_UN 6
#endif
#ifndef CFG_BIG_DMA     // 128@16, 256@8 stride-1: ring kernel with LDS-DMA patch staging (conv_ring_dma_kernel), maps >= 8 x 8
#define CFG_BIG_DMA 0
#endif
#ifndef CFG_BIG_DMA_WP  // ... waves along pixels (x CFG_BIG_WC = waves per workgroup): 4 -> 128-pixel tiles, 8 -> 256
#define CFG_BIG_DMA_WP 4
#endif
#ifndef CFG_BIG_DMA_WPB
#define CFG_BIG_DMA_WPB 1
#endif
#ifndef CFG_BIG_DMA_UNP   // patch pieces per patch wave and stage (upper bound, checked at launch)
#define CFG_BIG_DMA_UNP 7
#endif
#ifndef CFG_BIG_DMA_NWL   // extra loader waves (0: the compute waves issue the LDS-DMA themselves)
#define CFG_BIG_DMA_NWL 0
#endif
#ifndef CFG_BIG_DMA_FD    // fragment prefetch distance (items)
#define CFG_BIG_DMA_FD 2
#endif
#ifndef CFG_BIG_DMA_MINW  // min waves per SIMD (VGPR cap)
#define CFG_BIG_DMA_MINW 4
#endif
#ifndef CFG_BIG_DMA_RB  // ring depth, 128@16
#define CFG_BIG_DMA_RB 2
#endif
#ifndef CFG_256_DMA_RB  // ring depth, 256@8 (its 4-sample patch buffers leave room for 3 slots at most)
#define CFG_256_DMA_RB (CFG_BIG_DMA_RB > 3 ? 3 : CFG_BIG_DMA_RB)
#endif
#ifndef CFG_S2A_DMA      // 64->128 stride-2 (+shortcut) on conv_ring_dma_kernel: 0.51 -> 0.47 ms (one workgroup per CU, ring depth 3)
#define CFG_S2A_DMA 1
#endif
#ifndef CFG_S2B_DMA      // 128->256 stride-2 (+shortcut): with 5 taps per weight step 0.41 -> 0.345 ms (register-staged kernel at 5 taps: 0.365)
#define CFG_S2B_DMA 1
#endif
#ifndef CFG_S2_DMA_RB
#define CFG_S2_DMA_RB 2
#endif
#ifndef CFG_S2A_GT       // taps per weight step of the 64->128 ring-DMA kernel (10 weight taps: 1, 2, 5); 5 with ring depth 2: 0.46 -> 0.41 ms
#define CFG_S2A_GT 5
#endif
#ifndef CFG_S2B_GT       // ... of the 128->256 kernel (either variant)
#define CFG_S2B_GT 5
#endif
#ifndef CFG_S2_DMA_NWL
#define CFG_S2_DMA_NWL 0
#endif
#ifndef CFG_S2_DMA_MINW
#define CFG_S2_DMA_MINW 2
#endif
// dma (fast arithmetic only): 0 none, 1 resident-weights DMA mode of conv_mfma_kernel, 2 conv_ring_dma_kernel with
// its own pixel tiling {wpb, wp}_dma (the cout tiling, KC and GT -- i.e. the weight packing -- are shared)
#ifndef CFG_LAT        // small-batch ("latency") variants of the >= 128-channel layers: 32-cout tiles, 4 waves, 128 pixels
#define CFG_LAT 1
#endif
#ifndef CFG_LAT_GT     // taps per weight step, stride-1 latency variants (9 taps: 1, 3 or 9); batch-1 sweep: 9 is fastest
#define CFG_LAT_GT 9
#endif
#ifndef CFG_LAT_GT2    // stride-2 (+shortcut, 10 weight taps: 1, 2, 5 or 10)
#define CFG_LAT_GT2 10
#endif
struct CfgRow { int cin, cout, stride, kc[2], wcb, wpb, wc, wp, gt[2], dma, wpb_dma, wp_dma, lat; };  // lat: has a latency variant (fast arithmetic)
static const CfgRow kCfg[] = {
    //cin cout s   KC{fast,exact} WCB WPB WC WP  GT{fast,exact}
    {32, 32, 1, {32, 32}, 1, CFG_32_WPB, 1, CFG_32_WP, {9, 3}},
    {32, 64, 2, {32, 32}, CFG_3264_WCB, 1, CFG_3264_WC, CFG_3264_WP, {CFG_3264_GT, CFG_GTE_S2}},
    {64, 64, 1, {64, 32}, CFG_64_WCB, CFG_64_WPB, CFG_64_WC, CFG_64_WP, {CFG_64_GT, CFG_GTE_S1}, CFG_64_DMA, 0, 0, CFG_LAT},
    {64, 128, 2, {32, 32}, CFG_S2BIG_WCB, CFG_S2BIG_WPB, CFG_S2BIG_WC, CFG_S2BIG_WP, {2, CFG_GTE_S2}, CFG_S2A_DMA ? 2 : 0, CFG_S2BIG_WPB, CFG_S2BIG_WP, CFG_LAT},
    {128, 128, 1, {64, 32}, CFG_BIG_WCB, CFG_128_WPB, CFG_BIG_WC, CFG_BIG_WP, {CFG_BIG_GT, CFG_GTE_S1}, CFG_BIG_DMA ? 2 : 0, CFG_BIG_DMA_WPB, CFG_BIG_DMA_WP, CFG_LAT},
    {128, 256, 2, {32, 32}, CFG_S2BIG_WCB, CFG_S2BIG_WPB, CFG_S2BIG_WC, CFG_S2BIG_WP, {CFG_S2B_GT, CFG_GTE_S2}, CFG_S2B_DMA ? 2 : 0, CFG_S2BIG_WPB, CFG_S2BIG_WP, CFG_LAT},
    {256, 256, 1, {64, 32}, CFG_BIG_WCB, CFG_BIG_WPB, CFG_BIG_WC, CFG_BIG_WP, {CFG_BIG_GT, CFG_GTE_S1}, CFG_BIG_DMA ? 2 : 0, CFG_BIG_DMA_WPB, CFG_BIG_DMA_WP, CFG_LAT},
    // CU model (planes 32/64/96/128/256)
    {64, 96, 2, {32, 32}, 3, 1, 1, 4, {1, CFG_GTE_S2}},
    {96, 96, 1, {32, 32}, 3, 1, 1, 4, {3, CFG_GTE_S1}},
    {96, 128, 2, {32, 32}, 2, 2, 2, 2, {2, CFG_GTE_S2}},
};

// hi+lo-WEIGHTS tier (NSPLIT = 3) on the exact packing: layers whose cin chunk is 32 in both packings run the FAST tiling's taps per step --
// all taps resident + persistent workgroups for the 32-channel layers, 2 taps per ring step for the two big stride-2 layers (two weight
// planes: 5 taps would not fit the ring) -- instead of the exact tiling's one / three taps, which is sized for the small models' maps
// (round 3: 32->32 0.83 -> ... ms per launch in this tier).  Must mirror the nsplit == 3 branches of mlt_launch_conv.
static int w2_gt(int cin, int cout, int stride, int gt_exact) {
  if (cin == 32 && cout == 32 && stride == 1) return 9;
  if (cin == 32 && cout == 64 && stride == 2) return CFG_3264_GT;
  if (cin == 64 && cout == 128 && stride == 2) return 2;
  if (cin == 128 && cout == 256 && stride == 2) return 2;
  return gt_exact;
}

bool mlt_conv_cfg(int cin, int cout, int stride, int exact, ConvCfg *out) {
  for (const CfgRow &r : kCfg)
    if (r.cin == cin && r.cout == cout && r.stride == stride) {
      out->kc = r.kc[exact ? 1 : 0];
      out->ct = 32 * r.wcb * r.wc;
      out->dma = exact ? 0 : r.dma;
      out->mt = 32 * r.wpb * ((exact && r.cin >= 128 && r.stride == 1) ? CFG_BIG_WP_EXACT : r.wp);
      out->mt_dma = r.dma == 2 ? 32 * r.wpb_dma * r.wp_dma : out->mt;
      // the variants assume weights packed for 128-cout tiles (64 for the 64-channel layer)
      out->lat = (r.lat && r.wcb * r.wc == (r.cout == 64 ? 2 : 4)) ? 1 : 0;  // (round 4: the exact arithmetic has its latency variants too)
      out->gt = r.gt[exact ? 1 : 0];
      out->gt_w2 = w2_gt(r.cin, r.cout, r.stride, r.gt[1]);
      return true;
    }
  return false;
}

// RBF / RBE: weight-ring depth (fast / exact).  Must mirror kCfg.
// UNF / UNE: patch items per lane prefetched in registers (fast / exact), sized for the 128x128 model's tiles
// WPF / WPE: waves along pixels, fast / exact (the exact kernels serve the small-CU models, whose maps are tiny)
#define CONV_CASE2(CIN, COUT, STRIDE, SCF, KCF, KCE, WCB, WPB, WC, WPF, WPE, GTF, GTE, RBF, RBE, UNF, UNE, MWF)                              \
  if (cin == CIN && cout == COUT && stride == STRIDE) {                                                                             \
    if (!exact) return launch_conv_t<CIN, COUT, STRIDE, 9, SCF, KCF, 1, WCB, WPB, WC, WPF, GTF, RBF, UNF, MWF, false>(a, grid_x, extra_lds, st); \
    if (nsplit == 3) return launch_conv_t<CIN, COUT, STRIDE, 9, SCF, KCE, 3, WCB, WPB, WC, WPE, GTE, RBE, UNE, 1, false>(a, grid_x, extra_lds, st); \
    if (nsplit == 6) return launch_conv_t<CIN, COUT, STRIDE, 9, SCF, KCE, 6, WCB, WPB, WC, WPE, GTE, RBE, UNE, 1, false>(a, grid_x, extra_lds, st); \
    return launch_conv_t<CIN, COUT, STRIDE, 9, SCF, KCE, 2, WCB, WPB, WC, WPE, GTE, RBE, UNE, 1, false>(a, grid_x, extra_lds, st);          \
  }
#define CONV_CASE(CIN, COUT, STRIDE, SCF, KCF, KCE, WCB, WPB, WC, WP, GTF, GTE, RBF, RBE, UNF, UNE, MWF) \
  CONV_CASE2(CIN, COUT, STRIDE, SCF, KCF, KCE, WCB, WPB, WC, WP, WP, GTF, GTE, RBF, RBE, UNF, UNE, MWF)

// stride-2 convs always carry their block's projection shortcut (layer0.0's lives in stem5_kernel).
bool mlt_conv_has_centre_variant(int cin, int cout) { return (cin == cout && (cin == 128 || cin == 256)) || (cin == 128 && cout == 256); }

// nsplit: 1 fast, 2 exact (weights and activations hi+lo), 3 weights hi+lo only (the tiling of the exact kernels, 2 MFMAs, single activation planes)
hipError_t mlt_launch_conv(int cin, int cout, int stride, int nsplit, int variant, const ConvArgs &a, int grid_x, int extra_lds, hipStream_t st) {
  if (nsplit == 4) {
    // hi+lo weights on the FAST tiling (MLT_MODEL_W2).  The stride-2 and 32-channel layers have ONE tiling for both packings (same kc / ct), so
    // their bytes are those of the exact packing and they run the nsplit == 3 kernels below; the stride-1 layers with >= 64 channels (kc = 64)
    // run the 32-cout x 128-pixel variants on any launch size -- large launches of those layers go through chain_kernel<..., W2>, and this is
    // its bit-identical per-conv form: (chunk, tap, k-step, hi, lo) per accumulator.
    if (stride == 1 && cin == cout && cin >= 64) {
      if (variant != MLT_CONV_LATENCY) return hipErrorInvalidValue;
      // (NSPLIT = 5: hi fp16 plane + FP8 lo plane, the packing of these layers in the MLT_MODEL_W2 model)
      if (cin == 64) return launch_conv_t<64, 64, 1, 9, false, 64, 3, 1, 1, 1, 4, 9, 1, 6, 1, false, 2>(a, grid_x, extra_lds, st);   // (fp16 lo plane: see mlt_model.cpp)
      if (cin == 128) return launch_conv_t<128, 128, 1, 9, false, 64, 5, 1, 1, 1, 4, 3, 2, 6, 1, false, 4>(a, grid_x, extra_lds, st);
      if (cin == 256) return launch_conv_t<256, 256, 1, 9, false, 64, 5, 1, 1, 1, 4, 3, 2, 6, 1, false, 4>(a, grid_x, extra_lds, st);
      return hipErrorInvalidValue;
    }
    if (variant != MLT_CONV_DEFAULT) return hipErrorInvalidValue;
    nsplit = 3;
  }
  const bool exact = nsplit >= 2;
  const bool dma = variant == MLT_CONV_DMA;
  if (variant == MLT_CONV_CENTRE && (nsplit == 3 || nsplit == 6)) return hipErrorInvalidValue;  // (1x1 maps: small-CU models only, which do not use this tier)
  if (variant == MLT_CONV_CENTRE) {  // stride-1 layers of the small-CU models on 1x1 maps: centre tap only (TAPS = 1), 128 samples per tile
    if (cin == 128 && cout == 128 && stride == 1)
      return exact ? launch_conv_t<128, 128, 1, 1, false, 32, 2, CFG_BIG_WCB, 1, CFG_BIG_WC, CFG_BIG_WP_EXACT, 1, 1, 2, 1, false>(a, grid_x, extra_lds, st)
                   : launch_conv_t<128, 128, 1, 1, false, 64, 1, CFG_BIG_WCB, 1, CFG_BIG_WC, CFG_BIG_WP, 1, 1, 2, 1, false>(a, grid_x, extra_lds, st);
    if (cin == 128 && cout == 256 && stride == 2)  // 1x1 input: centre tap + projection shortcut, both 1x1 on the same pixel
      return exact ? launch_conv_t<128, 256, 2, 1, true, 32, 2, CFG_S2BIG_WCB, CFG_S2BIG_WPB, CFG_S2BIG_WC, CFG_S2BIG_WP, 1, 2, 3, 1, false>(a, grid_x, extra_lds, st)
                   : launch_conv_t<128, 256, 2, 1, true, 32, 1, CFG_S2BIG_WCB, CFG_S2BIG_WPB, CFG_S2BIG_WC, CFG_S2BIG_WP, 1, 2, 5, 1, false>(a, grid_x, extra_lds, st);
    if (cin == 256 && cout == 256 && stride == 1)
      return exact ? launch_conv_t<256, 256, 1, 1, false, 32, 2, CFG_BIG_WCB, 1, CFG_BIG_WC, CFG_BIG_WP_EXACT, 1, 1, 2, 1, false>(a, grid_x, extra_lds, st)
                   : launch_conv_t<256, 256, 1, 1, false, 64, 1, CFG_BIG_WCB, 1, CFG_BIG_WC, CFG_BIG_WP, 1, 1, 2, 1, false>(a, grid_x, extra_lds, st);
    return hipErrorInvalidValue;
  }
#if CFG_LAT
  if (variant == MLT_CONV_LATENCY && nsplit == 6) {  // exact-lite, small launches: the geometry of the exact arithmetic's latency variants (same per-accumulator order as the large tiles)
    if (cin == 64 && cout == 64 && stride == 1) return launch_conv_t<64, 64, 1, 9, false, 32, 6, 1, 1, 1, 4, CFG_GTE_S1, 2, 4, 1, false, 2>(a, grid_x, extra_lds, st);
    if (cin == 128 && cout == 128 && stride == 1) return launch_conv_t<128, 128, 1, 9, false, 32, 6, 1, 1, 1, 4, CFG_GTE_S1, 2, 4, 1, false, 4>(a, grid_x, extra_lds, st);
    if (cin == 256 && cout == 256 && stride == 1) return launch_conv_t<256, 256, 1, 9, false, 32, 6, 1, 1, 1, 4, CFG_GTE_S1, 2, 4, 1, false, 4>(a, grid_x, extra_lds, st);
    if (cin == 64 && cout == 128 && stride == 2) return launch_conv_t<64, 128, 2, 9, true, 32, 6, 1, 1, 1, 4, CFG_GTE_S2, 2, 5, 1, false, 4>(a, grid_x, extra_lds, st);
    if (cin == 128 && cout == 256 && stride == 2) return launch_conv_t<128, 256, 2, 9, true, 32, 6, 1, 1, 1, 4, CFG_GTE_S2, 2, 5, 1, false, 4>(a, grid_x, extra_lds, st);
    return hipErrorInvalidValue;
  }
  if (variant == MLT_CONV_LATENCY && nsplit == 2) {
    // Exact arithmetic, small launches (round 4: the guards' re-runs of a few flagged CUs, one-CU calls of the exact / exact-stage tiers, small
    // batches of the small models): the same 32-cout x 128-pixel tiles on the exact packing (4x the workgroups, a quarter of the weight
    // bytes each); per accumulator the order of the large tiles (chunk, tap, k-step: Wh Xh, Wh Xl, Wl Xh), so the results are the same bits.
    if (cin == 64 && cout == 64 && stride == 1) return launch_conv_t<64, 64, 1, 9, false, 32, 2, 1, 1, 1, 4, CFG_GTE_S1, 2, 4, 1, false, 2>(a, grid_x, extra_lds, st);
    if (cin == 128 && cout == 128 && stride == 1) return launch_conv_t<128, 128, 1, 9, false, 32, 2, 1, 1, 1, 4, CFG_GTE_S1, 2, 4, 1, false, 4>(a, grid_x, extra_lds, st);
    if (cin == 256 && cout == 256 && stride == 1) return launch_conv_t<256, 256, 1, 9, false, 32, 2, 1, 1, 1, 4, CFG_GTE_S1, 2, 4, 1, false, 4>(a, grid_x, extra_lds, st);
    if (cin == 64 && cout == 128 && stride == 2) return launch_conv_t<64, 128, 2, 9, true, 32, 2, 1, 1, 1, 4, CFG_GTE_S2, 2, 5, 1, false, 4>(a, grid_x, extra_lds, st);
    if (cin == 128 && cout == 256 && stride == 2) return launch_conv_t<128, 256, 2, 9, true, 32, 2, 1, 1, 1, 4, CFG_GTE_S2, 2, 5, 1, false, 4>(a, grid_x, extra_lds, st);
    return hipErrorInvalidValue;
  }
  if (variant == MLT_CONV_LATENCY && !exact) {  // 32 couts x 128 pixels per 4-wave workgroup, weights packed for CBP = 4
    if (cin == 64 && cout == 64 && stride == 1) return launch_conv_t<64, 64, 1, 9, false, 64, 1, 1, 1, 1, 4, 9, 1, 6, 1, false, 2>(a, grid_x, extra_lds, st);
    if (cin == 128 && cout == 128 && stride == 1) return launch_conv_t<128, 128, 1, 9, false, 64, 1, 1, 1, 1, 4, CFG_LAT_GT, 2, 6, 1, false, 4>(a, grid_x, extra_lds, st);
    if (cin == 256 && cout == 256 && stride == 1) return launch_conv_t<256, 256, 1, 9, false, 64, 1, 1, 1, 1, 4, CFG_LAT_GT, 2, 6, 1, false, 4>(a, grid_x, extra_lds, st);
    // (k-step-major items: the accumulation order of the whole-stage kernel, see KMAJ)
    static_assert(CFG_LAT_GT2 == 10, "the stride-2 latency variants take all 10 weight taps of a chunk per step");
    if (cin == 64 && cout == 128 && stride == 2) return launch_conv_t<64, 128, 2, 9, true, 32, 1, 1, 1, 1, 4, CFG_LAT_GT2, 2, 10, 1, false, 4, true>(a, grid_x, extra_lds, st);
    if (cin == 128 && cout == 256 && stride == 2) return launch_conv_t<128, 256, 2, 9, true, 32, 1, 1, 1, 1, 4, CFG_LAT_GT2, 2, 10, 1, false, 4, true>(a, grid_x, extra_lds, st);
    return hipErrorInvalidValue;
  }
#endif
#if CFG_BIG_DMA
  if (dma && !exact && cin == 128 && cout == 128 && stride == 1)
    return launch_ring_dma_t<128, 128, 1, false, 64, CFG_BIG_WCB, CFG_BIG_DMA_WPB, CFG_BIG_WC, CFG_BIG_DMA_WP, CFG_BIG_GT, CFG_BIG_DMA_RB, CFG_BIG_DMA_UNP, CFG_BIG_DMA_MINW, CFG_BIG_DMA_NWL, CFG_BIG_DMA_FD>(a, grid_x, st);
  if (dma && !exact && cin == 256 && cout == 256 && stride == 1)
    return launch_ring_dma_t<256, 256, 1, false, 64, CFG_BIG_WCB, CFG_BIG_DMA_WPB, CFG_BIG_WC, CFG_BIG_DMA_WP, CFG_BIG_GT, CFG_256_DMA_RB, CFG_BIG_DMA_UNP, CFG_BIG_DMA_MINW, CFG_BIG_DMA_NWL, CFG_BIG_DMA_FD>(a, grid_x, st);
#endif
#if CFG_S2A_DMA
  if (dma && !exact && cin == 64 && cout == 128 && stride == 2)
    return launch_ring_dma_t<64, 128, 2, true, 32, CFG_S2BIG_WCB, CFG_S2BIG_WPB, CFG_S2BIG_WC, CFG_S2BIG_WP, CFG_S2A_GT, CFG_S2_DMA_RB, 20, CFG_S2_DMA_MINW, CFG_S2_DMA_NWL, 2>(a, grid_x, st);
#endif
#if CFG_S2B_DMA
  if (dma && !exact && cin == 128 && cout == 256 && stride == 2)
    return launch_ring_dma_t<128, 256, 2, true, 32, CFG_S2BIG_WCB, CFG_S2BIG_WPB, CFG_S2BIG_WC, CFG_S2BIG_WP, CFG_S2B_GT, CFG_S2_DMA_RB, 20, CFG_S2_DMA_MINW, CFG_S2_DMA_NWL, 2>(a, grid_x, st);
#endif
#if CFG_64_DMA
  if (dma && !exact && cin == 64 && cout == 64 && stride == 1)
    return launch_conv_t<64, 64, 1, 9, false, 64, 1, CFG_64_WCB, CFG_64_WPB, CFG_64_WC, CFG_64_WP, 9, 1, CFG_64_DMA_UN, CFG_S1_MINW, true>(a, grid_x, extra_lds, st);
#endif
  if (nsplit == 3 && variant == MLT_CONV_DEFAULT) {  // hi+lo-weights tier with its own tiling (w2_gt above)
    static_assert(CFG_3264_GT == 10, "32->64 stride-2: all ten weight taps resident");
    if (cin == 32 && cout == 32 && stride == 1) return launch_conv_t<32, 32, 1, 9, false, 32, 3, 1, CFG_32_WPB, 1, CFG_32_WP, 9, 1, 5, 1, false>(a, grid_x, extra_lds, st);
    if (cin == 32 && cout == 64 && stride == 2) return launch_conv_t<32, 64, 2, 9, true, 32, 3, CFG_3264_WCB, 1, CFG_3264_WC, CFG_3264_WP, CFG_3264_GT, 1, CFG_3264_UN, 1, false>(a, grid_x, extra_lds, st);
    if (cin == 64 && cout == 128 && stride == 2) return launch_conv_t<64, 128, 2, 9, true, 32, 3, CFG_S2BIG_WCB, CFG_S2BIG_WPB, CFG_S2BIG_WC, CFG_S2BIG_WP, 2, 2, 5, 1, false>(a, grid_x, extra_lds, st);
    if (cin == 128 && cout == 256 && stride == 2) return launch_conv_t<128, 256, 2, 9, true, 32, 3, CFG_S2BIG_WCB, CFG_S2BIG_WPB, CFG_S2BIG_WC, CFG_S2BIG_WP, 2, 2, 5, 1, false>(a, grid_x, extra_lds, st);
  }
  CONV_CASE(32, 32, 1, false, 32, 32, 1, CFG_32_WPB, 1, CFG_32_WP, 9, 3, 1, 2, 5, CFG_UNE_32, CFG_32_MINW)
  CONV_CASE(32, 64, 2, true, 32, 32, CFG_3264_WCB, 1, CFG_3264_WC, CFG_3264_WP, CFG_3264_GT, CFG_GTE_S2, CFG_3264_RB, 2, CFG_3264_UN, CFG_UNE_32, CFG_32_MINW)
  CONV_CASE(64, 64, 1, false, 64, 32, CFG_64_WCB, CFG_64_WPB, CFG_64_WC, CFG_64_WP, CFG_64_GT, CFG_GTE_S1, CFG_S1_RB, 2, 6, CFG_UNE_64, CFG_S1_MINW)
  CONV_CASE(64, 128, 2, true, 32, 32, CFG_S2BIG_WCB, CFG_S2BIG_WPB, CFG_S2BIG_WC, CFG_S2BIG_WP, 2, CFG_GTE_S2, 2, 2, 5, CFG_UNE_S2, CFG_S2_MINW)
  CONV_CASE2(128, 128, 1, false, 64, 32, CFG_BIG_WCB, CFG_128_WPB, CFG_BIG_WC, CFG_BIG_WP, CFG_BIG_WP_EXACT, CFG_BIG_GT, CFG_GTE_S1, CFG_S1_RB, 2, 3, 2, CFG_S1_MINW)
  CONV_CASE(128, 256, 2, true, 32, 32, CFG_S2BIG_WCB, CFG_S2BIG_WPB, CFG_S2BIG_WC, CFG_S2BIG_WP, CFG_S2B_GT, CFG_GTE_S2, 2, 2, 5, CFG_UNE_S2, CFG_S2_MINW)
  CONV_CASE2(256, 256, 1, false, 64, 32, CFG_BIG_WCB, CFG_BIG_WPB, CFG_BIG_WC, CFG_BIG_WP, CFG_BIG_WP_EXACT, CFG_BIG_GT, CFG_GTE_S1, CFG_S1_RB, 2, 4, 2, CFG_S1_MINW)
  CONV_CASE(64, 96, 2, true, 32, 32, 3, 1, 1, 4, 1, CFG_GTE_S2, 2, 2, 5, 3, 1)
  CONV_CASE(96, 96, 1, false, 32, 32, 3, 1, 1, 4, 3, CFG_GTE_S1, 2, 2, 3, 2, 1)
  CONV_CASE(96, 128, 2, true, 32, 32, 2, 2, 2, 2, 2, CFG_GTE_S2, 2, 2, 5, 3, 1)
  return hipErrorInvalidValue;
}

// fused chain kernels: (channels, map height) -> instantiation.  128@16: one sample per 16-wave workgroup, weights packed for
// 128-cout tiles / 64-channel chunks / 3 taps per step (the stand-alone layer's packing).
// ---- LDS out-of-range probe: chain_kernel<..., OOBZ = true> lets a tap outside the map read beyond the LDS allocation and relies on
// the hardware returning zeros for such a DS read.  One workgroup with the chain kernels' full 160 KiB allocation reads where they do;
// *ok = 1 iff every lane saw zeros.  Run once per context at mlt_init: a device that answers differently gets the masked kernels. ----
__global__ __launch_bounds__(64) void lds_oob_probe_kernel(int *ok) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x;
  *(uint4v *)(smem + 160 * 1024 - 1024 + lane * 16) = uint4v{0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu};  // (the allocation is really there)
  __syncthreads();
  half8 v;
  lds_read128<96>(v, 0x100000u + (uint32_t)lane * 16);  // same base as the chain kernels, an offset field in use
  lds_wait<0>();
  lds_touch(v);
  const uint4v u = *(const uint4v *)&v;
  const bool zero = (u[0] | u[1] | u[2] | u[3]) == 0;
  const unsigned long long all = __ballot(zero);
  if (lane == 0) *ok = all == ~0ull ? 1 : 0;
}
hipError_t mlt_probe_lds_oob(int *d_ok, hipStream_t st) {
  static DeviceOnce once;
  hipError_t e = ensure_big_lds(lds_oob_probe_kernel, once);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(lds_oob_probe_kernel, dim3(1), dim3(64), 160 * 1024, st, d_ok);
  return hipGetLastError();
}

bool mlt_chain_supported(int c, int h) { return (c == 64 && h == 32) || (c == 128 && h == 16) || (c == 256 && h == 8); }
bool mlt_stage_supported(int c, int h) { return (c == 128 && h == 16) || (c == 256 && h == 8); }  // whole-stage (S2) variant

#ifndef CFG_CHAIN64_GT   // taps per weight step of the 64-channel chain: 1 (4-deep ring of 8 KiB steps, 9 barriers per conv) or 2 (2 x 16 KiB, 5 barriers)
#define CFG_CHAIN64_GT 2
#endif
#ifndef CFG_CHAIN_FD     // fragment prefetch distance of chain_kernel (items)
#define CFG_CHAIN_FD 2
#endif
#ifndef CFG_CHAIN_SPLIT  // 1: ring DMA on the first half of the waves, activation DMA on the second; 0: every wave issues both
#define CFG_CHAIN_SPLIT 1
#endif
#ifndef CFG_CHAINW2_RB   // ring depth of the hi+lo-weights stage chains (32 KiB steps beside a 64 KiB sample: 2 or 3)
#define CFG_CHAINW2_RB 3
#endif
template <class K> static hipError_t launch_chain_t(K kern, DeviceOnce &once, const ChainArgs &a, int grid_x, int threads, int lds, hipStream_t st) {
  if (hipError_t e = ensure_big_lds(kern, once); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid_x), dim3(threads), lds, st, a);
  return hipGetLastError();
}
hipError_t mlt_launch_chain(int c, int h, bool with_s2, bool oob_zero, bool w2, const ChainArgs &a, int grid_x, hipStream_t st) {
  static_assert(CFG_BIG_GT == 3 && CFG_BIG_WCB == 2 && CFG_BIG_WC == 2, "chain_kernel reads the packing of the stand-alone 128->128 / 256->256 layers");
  constexpr int lds = 64 * 1024 + 2 * (CFG_BIG_GT * 4 * 4 * 1024);  // 64 KiB activation + two 48 KiB weight steps = all of the LDS
  static DeviceOnce once[13];
  if (w2) {  // hi+lo-weights forms (MLT_MODEL_W2 packing: the fast tiling, two planes): one tap per ring step, conv padding from beyond the LDS only
    if (!oob_zero || with_s2 || a.nconv != 3) return hipErrorInvalidValue;
    // ring step = hi fp16 plane + FP8 lo plane (LO8): 8 + 4 KiB per tap (64 channels), 16 + 8 KiB (128 / 256)
    if (c == 64 && h == 32)   // 128 KiB sample + 2 steps of two fp16 planes (8 + 8 KiB)
      return launch_chain_t(chain_kernel<64, 5, 0, 2, 4, 1, 8, 1, 2, 1, 2, 3, true, false, false, true, true, false>, once[10], a, grid_x, 512, 128 * 1024 + 2 * 16 * 1024, st);
    if (c == 128 && h == 16)  // 64 KiB sample + 2 ring steps + the 32 KiB e4m3 image of the sample
      return launch_chain_t(chain_kernel<128, 4, 0, 2, 2, 2, 4, 1, 2, 1, 2, 3, true, false, true, true, true, true>, once[11], a, grid_x, 512, (64 + 2 * 24 + 32) * 1024, st);
    if (c == 256 && h == 8)   // (64 KiB sample + 2 ring steps + the 32 KiB e4m3 image of the sample)
      return launch_chain_t(chain_kernel<256, 3, 1, 2, 1, 2, 4, 1, 2, 1, 2, 3, true, false, true, true, true, true>, once[12], a, grid_x, 512, (64 + 2 * 24 + 32) * 1024, st);
    return hipErrorInvalidValue;
  }
  // oob_zero: conv padding from DS reads beyond the LDS allocation (the probed default) or from zero masks; the kernels off the default path
  // (chains without the stride-2 front conv, MLT_NO_CHAIN_S2) exist in the masked form only
  if (c == 64 && h == 32 && a.nconv == 3 && !with_s2) {  // 8 waves x (64 couts x 128 pixels), one 128 KiB sample per workgroup, 2 x 16 KiB weight ring
    static_assert(CFG_64_WCB == 2 && CFG_64_WC == 1 && CFG_64_GT == 9, "chain_kernel<64,...> reads the packing of the stand-alone 64->64 layer");
    constexpr int RB64 = CFG_CHAIN64_GT == 1 ? 4 : 2, lds64 = 128 * 1024 + 4 * 8 * 1024;
    if (oob_zero) return launch_chain_t(chain_kernel<64, 5, 0, 2, 4, 1, 8, CFG_CHAIN64_GT, RB64, 1, 2, 3, true, false, false, true>, once[6], a, grid_x, 512, lds64, st);
    return launch_chain_t(chain_kernel<64, 5, 0, 2, 4, 1, 8, CFG_CHAIN64_GT, RB64, 1, 2, 3, true, false, false, false>, once[7], a, grid_x, 512, lds64, st);
  }
  if (c == 128 && h == 16) {  // 8 waves x (64 couts x 64 pixels), one sample per workgroup
    if (with_s2 && a.nconv == 3) {
      if (oob_zero) return launch_chain_t(chain_kernel<128, 4, 0, 2, 2, 2, 4, 3, 2, CFG_CHAIN_FD, 2, 3, true, true, true, true>, once[4], a, grid_x, 512, lds, st);
      return launch_chain_t(chain_kernel<128, 4, 0, 2, 2, 2, 4, 3, 2, CFG_CHAIN_FD, 2, 3, true, true, true, false>, once[8], a, grid_x, 512, lds, st);
    }
    if (a.nconv == 3) return launch_chain_t(chain_kernel<128, 4, 0, 2, 2, 2, 4, 3, 2, CFG_CHAIN_FD, 2, 3, CFG_CHAIN_SPLIT != 0>, once[0], a, grid_x, 512, lds, st);
    if (a.nconv == 2) return launch_chain_t(chain_kernel<128, 4, 0, 2, 2, 2, 4, 3, 2, CFG_CHAIN_FD, 2, 2, CFG_CHAIN_SPLIT != 0>, once[1], a, grid_x, 512, lds, st);
  }
  if (c == 256 && h == 8) {   // 8 waves x (64 couts x 32 pixels) x 2 cout passes, two samples per workgroup
    if (with_s2 && a.nconv == 3) {
      if (oob_zero) return launch_chain_t(chain_kernel<256, 3, 1, 2, 1, 2, 4, 3, 2, CFG_CHAIN_FD, 2, 3, true, true, true, true>, once[5], a, grid_x, 512, lds, st);
      return launch_chain_t(chain_kernel<256, 3, 1, 2, 1, 2, 4, 3, 2, CFG_CHAIN_FD, 2, 3, true, true, true, false>, once[9], a, grid_x, 512, lds, st);
    }
    if (a.nconv == 3) return launch_chain_t(chain_kernel<256, 3, 1, 2, 1, 2, 4, 3, 2, CFG_CHAIN_FD, 2, 3, CFG_CHAIN_SPLIT != 0>, once[2], a, grid_x, 512, lds, st);
    if (a.nconv == 2) return launch_chain_t(chain_kernel<256, 3, 1, 2, 1, 2, 4, 3, 2, CFG_CHAIN_FD, 2, 2, CFG_CHAIN_SPLIT != 0>, once[3], a, grid_x, 512, lds, st);
  }
  return hipErrorInvalidValue;
}

hipError_t mlt_launch_stem5(const Stem5Args &a, int nsplit, int grid_x, int lds, hipStream_t st) {
  if (nsplit == 3) hipLaunchKernelGGL(stem5_kernel<3>, dim3(grid_x), dim3(256), lds, st, a);
  else if (nsplit == 2) hipLaunchKernelGGL(stem5_kernel<2>, dim3(grid_x), dim3(256), lds, st, a);
  else hipLaunchKernelGGL(stem5_kernel<1>, dim3(grid_x), dim3(256), lds, st, a);
  return hipGetLastError();
}

hipError_t mlt_launch_block32(const Block32Args &a, bool w2, int grid_x, hipStream_t st) {
  static DeviceOnce once[2];
  if (w2) {  // 8 x 32 tiles: X 12 x 36, T 10 x 34 pixels at 80 B + two weight planes per conv
    constexpr int lds = (12 * 36 + 10 * 34) * 80 + 4 * 18 * 1024;
    if (hipError_t e = ensure_big_lds(block32_kernel<3, true>, once[1]); e != hipSuccess) return e;
    hipLaunchKernelGGL((block32_kernel<3, true>), dim3(grid_x), dim3(512), lds, st, a);
    return hipGetLastError();
  }
  constexpr int lds = (20 * 36 + 18 * 34) * 80 + 2 * 18 * 1024;
  if (hipError_t e = ensure_big_lds(block32_kernel<4, false>, once[0]); e != hipSuccess) return e;
  hipLaunchKernelGGL((block32_kernel<4, false>), dim3(grid_x), dim3(512), lds, st, a);
  return hipGetLastError();
}

hipError_t mlt_launch_stem_block(const StemBlockArgs &a, bool w2, int grid_x, hipStream_t st) {
  // 3 raw + 2 T buffers + conv2 weights (hi+lo-weights form: its LO plane -- the hi plane lives in the consumer waves' registers) + biases +
  // border k-steps (one or two planes) = 149 / 151 KiB
  constexpr int lds = 3 * (39 * 72 * 4 + 16) + 2 * (18 * 34 * 80) + 18 * 1024 + 256 + 2 * 1024;
  static DeviceOnce once[2];
  if (w2) {
    if (hipError_t e = ensure_big_lds(stem_block_kernel<true>, once[1]); e != hipSuccess) return e;
    hipLaunchKernelGGL(stem_block_kernel<true>, dim3(grid_x), dim3(128 * CFG_SB_NWS), lds + 2 * 1024, st, a);
    return hipGetLastError();
  }
  if (hipError_t e = ensure_big_lds(stem_block_kernel<false>, once[0]); e != hipSuccess) return e;
  hipLaunchKernelGGL(stem_block_kernel<false>, dim3(grid_x), dim3(128 * CFG_SB_NWS), lds, st, a);  // one workgroup per CU, two pipeline stages inside
  return hipGetLastError();
}

template <bool F5, bool M16> static hipError_t launch_layer0_stream_t(const Layer0Args &a, int grid_x, hipStream_t st) {
  static DeviceOnce once;
  if (hipError_t e = ensure_big_lds(layer0_stream_kernel<F5, M16>, once); e != hipSuccess) return e;
  hipLaunchKernelGGL((layer0_stream_kernel<F5, M16>), dim3(grid_x), dim3(1024), F5 ? MLT_L0F_LDS_BYTES : MLT_L0_LDS_BYTES, st, a);  // one persistent workgroup per CU slot
  return hipGetLastError();
}
// fuse5: + layer1's stride-2 conv and shortcut as a fifth stage; mfma32: round 5's MFMA shape (MLT_TUNING=1 MLT_L0_MFMA32=1: same-box A/B; same bits).
// The four-stage form (fuse5 = false: weight sets whose layer1 does not open on the single pass) stays on 32x32x16: its 16x16x32 build reloads three of S2's
// weight fragments from scratch every row (S2 carries two accumulator sets; the five-stage form's register allocation comes out without that), and it
// was the five-stage form the A/B measured.
hipError_t mlt_launch_layer0_stream(const Layer0Args &a, bool fuse5, bool mfma32, int grid_x, hipStream_t st) {
  if (fuse5) return mfma32 ? launch_layer0_stream_t<true, false>(a, grid_x, st) : launch_layer0_stream_t<true, true>(a, grid_x, st);
  return launch_layer0_stream_t<false, false>(a, grid_x, st);
}

hipError_t mlt_launch_layer1_stream(const Layer1Args &a, bool mfma32, int grid_x, hipStream_t st) {
  static DeviceOnce once[2];
  if (mfma32) {  // round 5's form on v_mfma_f32_32x32x16_f16 (MLT_TUNING=1 MLT_L1_MFMA32=1: same-box A/B; same bits)
    if (hipError_t e = ensure_big_lds(layer1_stream_kernel<false>, once[0]); e != hipSuccess) return e;
    hipLaunchKernelGGL(layer1_stream_kernel<false>, dim3(grid_x), dim3(1024), MLT_L1_LDS_BYTES, st, a);
    return hipGetLastError();
  }
  if (hipError_t e = ensure_big_lds(layer1_stream_kernel<true>, once[1]); e != hipSuccess) return e;
  hipLaunchKernelGGL(layer1_stream_kernel<true>, dim3(grid_x), dim3(1024), MLT_L1_LDS_BYTES, st, a);
  return hipGetLastError();
}

hipError_t mlt_launch_heads(const HeadArgs &a, int n, hipStream_t st) {
  hipLaunchKernelGGL(heads_kernel, dim3(n), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t mlt_launch_flat_stat(const FlatStatArgs &a, bool aligned8, hipStream_t st) {
  if (aligned8) hipLaunchKernelGGL(flat_stat_kernel<true>, dim3(a.n), dim3(256), 0, st, a);
  else hipLaunchKernelGGL(flat_stat_kernel<false>, dim3(a.n), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t mlt_launch_guard_select(const GuardSelectArgs &a, hipStream_t st) {
  hipLaunchKernelGGL(guard_select_kernel, dim3(1), dim3(1024), 0, st, a);
  return hipGetLastError();
}

hipError_t mlt_launch_guard_gather(const GuardGatherArgs &a, hipStream_t st) {
  hipLaunchKernelGGL(guard_gather_kernel, dim3(a.k, 2), dim3(256), 0, st, a);
  return hipGetLastError();
}

hipError_t mlt_launch_guard_scatter(const GuardScatterArgs &a, hipStream_t st) {
  const int items = a.k * (a.n_logits + 1);
  hipLaunchKernelGGL(guard_scatter_kernel, dim3((items + 255) / 256), dim3(256), 0, st, a);
  return hipGetLastError();
}
