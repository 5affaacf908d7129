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
__device__ __forceinline__ bool quad_near_flat(uint32_t w0, uint32_t w1, uint32_t w2, uint32_t w3) {
  const half2v a = *(half2v *)&w0, b = *(half2v *)&w1, c = *(half2v *)&w2, d = *(half2v *)&w3;
  const half2v mx = __builtin_elementwise_max(__builtin_elementwise_max(a, b), __builtin_elementwise_max(c, d));
  const half2v mn = __builtin_elementwise_min(__builtin_elementwise_min(a, b), __builtin_elementwise_min(c, d));
  const half2v r = mx - mn;
  const half2v d1 = (a + c) - (b + b), d2 = (b + d) - (c + c);
  const half2v l = __builtin_elementwise_max(__builtin_elementwise_max(d1, -d1), __builtin_elementwise_max(d2, -d2));
  const _Float16 R = (_Float16)MLT_FLAT_RANGE, one = (_Float16)1;
  return (r[0] <= R || l[0] <= one) && (r[1] <= R || l[1] <= one);
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
template <int Q> __device__ __forceinline__ half4 act_quad(const float16v &A, float scale, const float4v &b, bool has_res, const half4 &r, bool relu, float *x4 = nullptr) {
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

// ---------------------------------------------------------------------------------------------
// Shared epilogue of the conv kernels: + bias (+ residual) (ReLU) -> fp16 NHWC (16 B per lane and quad pair) and/or
// fp32 GAP partial sums; SC: second output (projection shortcut, no ReLU).  The caller has loaded the biases (bq / bsq)
// and, when a.res is set, the residual (resv / resl) into registers.
// ---------------------------------------------------------------------------------------------
// bias_lds != nullptr: the cout tile's biases ([CT] floats, shortcut biases behind them) are read from LDS one
// 32-channel block at a time instead of from bq / bsq (fewer live registers).
template <int COUT, int CT, int WCB, int WPB, bool SC, int NSPLIT>
__device__ __forceinline__ void conv_epilogue(const ConvArgs &a, int ctile, int wc, int h, int p, const int (&opix)[WPB], const int (&gidx)[WPB],
                                              float16v (&acc)[WCB][WPB], float16v (&acc_sc)[SC ? WCB : 1][SC ? WPB : 1],
                                              const float4v (&bq)[WCB][4], const float4v (&bsq)[SC ? WCB : 1][4],
                                              const uint4v (&resv)[WCB][WPB][2],
                                              const uint4v (&resl)[NSPLIT == 2 ? WCB : 1][NSPLIT == 2 ? WPB : 1][2],
                                              const float *bias_lds = nullptr) {
  const int gl = a.gap_l;  // log2(lanes that share one sample in a 32-pixel block): 0, 2, 4 or 5
#pragma unroll
  for (int i = 0; i < WCB; ++i) {
    const int cbase = ctile * CT + (wc * WCB + i) * 32 + 4 * h;
    float4v bi[4], bsi[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bi[q] = bias_lds ? *(const float4v *)(bias_lds + (wc * WCB + i) * 32 + 4 * h + 8 * q) : bq[i][q];
      if constexpr (SC) bsi[q] = bias_lds ? *(const float4v *)(bias_lds + CT + (wc * WCB + i) * 32 + 4 * h + 8 * q) : bsq[i][q];
    }
#pragma unroll
    for (int j = 0; j < WPB; ++j) {
      const bool ok = opix[j] >= 0;
      const size_t o16 = (size_t)(ok ? opix[j] : 0) * COUT + ctile * CT + (wc * WCB + i) * 32 + 8 * h;  // this lane's 16-byte span of quad pair 0
      float v[16];
      half4 hq[4], hl[4], sq[4], sl[4];
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        half4 ra, rb, rla, rlb;
        if (a.res) {  // uniform
          unpair16(resv[i][j][qq], ra, rb);
          if constexpr (NSPLIT == 2) unpair16(resl[i][j][qq], rla, rlb);
        }
        if constexpr (NSPLIT == 1) {
          static_for<2>([&](auto kk) {
            constexpr int k = decltype(kk)::value;
            static_for<2>([&](auto qc) {
              constexpr int qq_c = decltype(qc)::value;
              if (qq_c == qq) {
                constexpr int q = 2 * qq_c + k;
                hq[q] = act_quad<q>(acc[i][j], a.acc_scale, bi[q], a.res != nullptr, k ? rb : ra, a.relu != 0, &v[4 * q]);
                if constexpr (SC) {
                  const half4 none{};
                  sq[q] = act_quad<q>(acc_sc[i][j], a.acc_scale, bsi[q], false, none, false);
                }
              }
            });
          });
        } else {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int q = 2 * qq + k;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float x = acc[i][j][4 * q + e] * a.acc_scale + bi[q][e];
            if (a.res) {
              x += (float)(k ? rb[e] : ra[e]);
              if constexpr (NSPLIT == 2) x += (float)(k ? rlb[e] : rla[e]);
            }
            if (a.relu) x = fmaxf(x, 0.f);
            v[4 * q + e] = x;
            hq[q][e] = (_Float16)x;
            hl[q][e] = (_Float16)(x - (float)hq[q][e]);
            if constexpr (SC) {
              const float vs = acc_sc[i][j][4 * q + e] * a.acc_scale + bsi[q][e];
              sq[q][e] = (_Float16)vs;
              sl[q][e] = (_Float16)(vs - (float)sq[q][e]);
            }
          }
        }
        }
        // every lane takes part in the swaps; only valid pixels store
        if (a.y) {
          const uint4v w = pair16(hq[2 * qq], hq[2 * qq + 1]);
          if (a.y_c16) {  // chunk-major: [n][COUT / 16][HW][16]; this lane holds channels 8h .. 8h+7 of chunk (cbase + 16 qq) / 16
            const int op = ok ? opix[j] : 0, hw_l = 2 * a.hout_l, nn = op >> hw_l, pix = op & ((1 << hw_l) - 1);
            const int chunk = (ctile * CT + (wc * WCB + i) * 32 + 16 * qq) >> 4;
            if (ok) *(uint4v *)((_Float16 *)a.y + ((((size_t)nn * (COUT / 16) + chunk) << hw_l) + pix) * 16 + 8 * h) = w;
          } else if (ok) *(uint4v *)((_Float16 *)a.y + o16 + 16 * qq) = w;
          if constexpr (NSPLIT == 2) {
            const uint4v wl = pair16(hl[2 * qq], hl[2 * qq + 1]);
            if (ok) *(uint4v *)((char *)((_Float16 *)a.y + o16 + 16 * qq) + a.y_lo_off) = wl;
          }
        }
        if constexpr (SC) {
          const uint4v w = pair16(sq[2 * qq], sq[2 * qq + 1]);
          if (NSPLIT == 1 && a.ysc_c16) {  // chunk-major, as y above
            const int op = ok ? opix[j] : 0, hw_l = 2 * a.hout_l, nn = op >> hw_l, pix = op & ((1 << hw_l) - 1);
            const int chunk = (ctile * CT + (wc * WCB + i) * 32 + 16 * qq) >> 4;
            if (ok) *(uint4v *)((_Float16 *)a.y_sc + ((((size_t)nn * (COUT / 16) + chunk) << hw_l) + pix) * 16 + 8 * h) = w;
          } else if (ok) *(uint4v *)((_Float16 *)a.y_sc + o16 + 16 * qq) = w;
          if constexpr (NSPLIT == 2) {
            const uint4v wl = pair16(sl[2 * qq], sl[2 * qq + 1]);
            if (ok) *(uint4v *)((char *)((_Float16 *)a.y_sc + o16 + 16 * qq) + a.ysc_lo_off) = wl;
          }
        }
      }
      if (a.gap) {
        // Sum over the pixels of one sample inside this 32-pixel block, in fp32, BEFORE any fp16 rounding.
        // Halving butterfly: after the steps for lane bits 0..3 each lane holds ONE channel's sum, channel
        // register index = b0*8 + b1*4 + b2*2 + b3 (b_k = bit k of p).  Fixed order => deterministic.
        if constexpr (NSPLIT == 1) {  // (the packed path left the values of before the ReLU in v)
          if (a.relu) {
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
          }
        }
        if (!ok) {
#pragma unroll
          for (int r = 0; r < 16; ++r) v[r] = 0.f;
        }
        if (gl == 0) {
          if (ok) {
            float *grow = a.gap + (size_t)gidx[j] * COUT + cbase;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
              for (int e = 0; e < 4; ++e) grow[8 * q + e] = v[4 * q + e];
          }
        } else {
          float r8[8], r4[4];
          {
            const bool up = p & 1;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
              float send = up ? v[k] : v[8 + k], keep = up ? v[8 + k] : v[k];
              r8[k] = keep + __shfl_xor(send, 1, 64);
            }
          }
          {
            const bool up = p & 2;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              float send = up ? r8[k] : r8[4 + k], keep = up ? r8[4 + k] : r8[k];
              r4[k] = keep + __shfl_xor(send, 2, 64);
            }
          }
          if (gl == 2) {  // 4 pixels per sample: lane holds 4 channel sums, register index (p&1)*8 + ((p>>1)&1)*4 + k
            if (ok) {
              float *grow = a.gap + (size_t)gidx[j] * COUT + cbase;
              const int ri = (p & 1) * 8 + ((p >> 1) & 1) * 4;
#pragma unroll
              for (int k = 0; k < 4; ++k) grow[8 * ((ri + k) >> 2) + ((ri + k) & 3)] = r4[k];
            }
          } else {
            float r2[2], r1;
            {
              const bool up = p & 4;
#pragma unroll
              for (int k = 0; k < 2; ++k) {
                float send = up ? r4[k] : r4[2 + k], keep = up ? r4[2 + k] : r4[k];
                r2[k] = keep + __shfl_xor(send, 4, 64);
              }
            }
            {
              const bool up = p & 8;
              float send = up ? r2[0] : r2[1], keep = up ? r2[1] : r2[0];
              r1 = keep + __shfl_xor(send, 8, 64);
            }
            if (gl == 5) r1 += __shfl_xor(r1, 16, 64);
            // all pixels of the group belong to one sample: take the row from the group's first lane
            const int grp_lane0 = gl == 5 ? 0 : (p & 16);
            const int g0 = __shfl(gidx[j], (h << 5) + grp_lane0, 64);
            if (g0 >= 0 && (gl == 4 || p < 16)) {
              const int ri = (p & 1) * 8 + ((p >> 1) & 1) * 4 + ((p >> 2) & 1) * 2 + ((p >> 3) & 1);
              a.gap[(size_t)g0 * COUT + cbase + 8 * (ri >> 2) + (ri & 3)] = r1;
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Generic conv (3x3 pad 1 or 1x1 pad 0, stride 1 or 2), NHWC fp16 -> NHWC fp16, fp32 accumulate.
// One workgroup: MT = 32*WPB*WAVES_P output pixels (SPW samples x TH x TW) x CT = 32*WCB*WAVES_C
// output channels.  Per 64-or-32-channel input chunk the (haloed) input patch is staged ONCE in LDS
// and reused by all taps; weights stream through a double-buffered LDS ring by LDS-DMA, GT taps per step.
//   SC   : the block's 1x1 stride-2 projection shortcut (arch:44-50) rides along as weight "tap 9" on the
//          same patch and leaves through a second accumulator / output (y_sc).
//   gap  : fp32 per-channel partial sums of the activated output (global average pooling, arch:282).
// ---------------------------------------------------------------------------------------------
//   NSPLIT = 2 ("exact" mode): every activation and weight is an fp16 (hi, lo) pair with hi + lo == the fp32
//          value to ~2^-22; products are accumulated as Wh*Xh + Wh*Xl + Wl*Xh in fp32 (3 MFMAs), which
//          restores ~fp32 accuracy on the fp16 matrix cores.  Planes: x / y / res / y_sc hold hi at the base
//          pointer and lo at base + a.*_lo_off bytes; the LDS patch and the weight ring are doubled.
//   DMA    (weights-resident stride-1 kernels, fast arithmetic): the patch is not staged through registers but written
//          straight into LDS by LDS-DMA (global_load_lds), double-buffered: while the MFMAs of tile i read buffer A
//          the DMA of tile i+1 fills buffer B, so a workgroup has a whole tile time of patch traffic in flight and
//          no commit phase, and the only barrier per tile is the buffer swap.  The DMA writes 64 x 16 contiguous
//          bytes per wave-instruction, so pixels cannot be padded apart; instead the 16-byte channel slots of a
//          pixel are XOR-swizzled with the pixel index (each lane picks the GLOBAL slot it fetches) which makes the
//          fragment ds_read_b128 conflict-free for 16 consecutive patch pixels, as the padding does in the other modes.
//          Picture-border / batch-tail items are fetched from a zero page (a.zero).
//   CBP    > 0: the weights are packed for cout tiles of CBP 32-channel blocks but this instantiation's tile has fewer
//          (CBT divides CBP): blockIdx.y counts the narrower tiles and the weight steps gather their 1 KiB pieces with
//          the packed stride.  Used by the small-batch ("latency") variants, which spread the couts of a layer over
//          4x more workgroups than the throughput tiling without a second packed copy of the weights.
//   KMAJ   the items of a weight step run k-step-major (all taps of the first 16 channels of the chunk, then of the next 16) instead
//          of tap-major.  Same products, another fp32 accumulation order: the order of chain_kernel's stride-2 phase, which stages
//          16 channels at a time -- so the small-launch variants of the stride-2 convs and the whole-stage kernel agree bit for bit.
template <int CIN, int COUT, int STRIDE, int TAPS, bool SC, int KC, int NSPLIT, int WCB, int WPB, int WAVES_C, int WAVES_P, int GT, int RB, int UN, int MINW, bool DMA, int CBP = 0, bool KMAJ = false>
// MINW = minimum waves per SIMD (second __launch_bounds__ argument, caps the VGPR allocation); 1 = unconstrained
__global__ __launch_bounds__(64 * WAVES_C * WAVES_P, MINW) void conv_mfma_kernel(const ConvArgs a) {
  static_assert(CIN % KC == 0 && (KC == 32 || KC == 64), "cin chunking");
  constexpr int NCHUNK = CIN / KC;
  constexpr int KS = KC / 16;
  constexpr int SLOTS = KC / 8;          // 16-byte slots per pixel
  constexpr int PS = KC * 2 + 16;        // LDS pixel stride (bytes); PS/16 odd -> conflict-free rows
  constexpr int CBT = WCB * WAVES_C;     // 32-channel blocks per workgroup tile
  constexpr int CT = 32 * CBT;
  constexpr int NW = WAVES_C * WAVES_P;
  constexpr int NT = 64 * NW;
  constexpr int TT = TAPS + (SC ? 1 : 0);  // weight steps per chunk (taps + shortcut)
  constexpr int NG = TT / GT;
  constexpr int WCHUNK = GT * KS * CBT * 1024;  // bytes of one weight step (per split plane)
  constexpr int NBUF = NG > 1 ? RB : 1;         // weight ring depth; steps are prefetched NBUF-1 ahead
  constexpr int PFD = NBUF > 1 ? NBUF - 1 : 0;  // prefetch distance (steps)
  constexpr int NPIECE = WCHUNK / 1024;                       // 1 KiB LDS-DMA pieces per step and plane
  // NSPLIT: 1 fast; 2 exact (weights AND activations are hi+lo pairs: Wh*Xh + Wh*Xl + Wl*Xh); 3 weights hi+lo only ((Wh + Wl) * X, 2 MFMAs,
  // single activation planes): WS weight planes in the ring, XS activation planes in the patch / residual / outputs
  constexpr int WS = NSPLIT >= 2 ? 2 : 1, XS = NSPLIT == 2 ? 2 : 1;
  constexpr int PPW = WS * ((NPIECE + NW - 1) / NW);      // LDS-DMA instructions EVERY wave issues per step
  constexpr int PAD = TAPS == 9 ? 1 : 0;
  constexpr int SCW = SC ? WCB : 1, SPB = SC ? WPB : 1;
  static_assert(TT % GT == 0, "tap grouping");
  static_assert(COUT % CT == 0, "cout tiling");
  static_assert(!SC || (STRIDE == 2 && (TAPS == 9 || TAPS == 1)), "shortcut rides on stride-2 convs (3x3, or its centre tap on a 1x1 input)");
  static_assert(RB >= 1 && RB <= 4, "ring depth");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const size_t w_lo = a.w_lo_off;                      // byte offset of the lo weight plane (NSPLIT == 2)

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % WAVES_C, wp = wave / WAVES_C;
  char *patch = smem;                                                  // [NSPLIT][patch_bytes]; DMA: two buffers of patch_bytes
  char *wring = smem + (DMA ? 2 : XS) * a.patch_bytes;                 // [NBUF][WS][WCHUNK]
  constexpr bool ASM_PIPE = CFG_ASM_PIPE && NSPLIT == 1;                // see lds_read128
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);            // LDS byte address of the dynamic segment
  constexpr int PIXROW_L = KC == 64 ? 1 : 2;                           // DMA swizzle: log2(pixels per 256-byte LDS bank row)
  const int p = lane & 31, h = lane >> 5;

  // ---- geometry (all powers of two) ----
  const int tw_l = a.tw_l, th_l = a.th_l, spw_l = a.spw_l;
  const int TW = 1 << tw_l, TH = 1 << th_l;
  const int hout_l = a.hout_l, hin_l = a.hin_l;
  const int Hin = 1 << hin_l;
  const int txs_l = hout_l - tw_l, tys_l = hout_l - th_l;  // tiles per row / column of one sample
  const int PH = a.ph, PW = a.pw, RP = a.rp, HALF = a.half;
  const int m_valid = 1 << (tw_l + th_l + spw_l);
  const int ctile = blockIdx.y;
  const int ntiles = a.ntiles;

  // Workgroups are PERSISTENT over tiles (t = blockIdx.x, += gridDim.x).  XCD-aware order: workgroups are dealt
  // round-robin over the 8 XCDs (private L2 each); logical tile t maps to a physical tile so that every XCD owns a
  // CONTIGUOUS run of tiles (neighbouring tiles share halo rows -> same L2).  Bijective for any ntiles (T1).
  auto tile_decode = [&](int t, int &tx, int &ty, int &n0) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = t & 7;
    const int mt = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
    tx = mt & ((1 << txs_l) - 1);
    ty = (mt >> txs_l) & ((1 << tys_l) - 1);
    n0 = (mt >> (txs_l + tys_l)) << spw_l;
  };

  // ---- patch loader, split into ISSUE (global -> registers, asynchronous) and COMMIT (registers -> LDS) so that
  // the loads of the NEXT stage (next channel chunk, or the next tile's first chunk) fly while THIS stage computes ----
  // UN (template): patch items per lane held in registers across a stage; larger patches take the synchronous tail
  constexpr int PSTEP = NT / SLOTS;                       // pixels advanced per item step; a lane's 16-byte slot is fixed
  const int slot = tid & (SLOTS - 1);
  const int patch_items = (1 << spw_l) * PH * PW * SLOTS;
  const int step_r = udiv_magic(PSTEP, a.pw_magic), step_x = PSTEP - step_r * PW;
  half8 pv[UN], pvl[NSPLIT == 2 ? UN : 1];
  int pdst[UN];  // LDS byte offset; bit 30 set: zero-fill (outside the picture / batch); -1: no item
  // Branch-free: every load is issued (from a clamped, always-valid address, DISTINCT per lane and workgroup -- never one
  // shared hot line) before any result is used; a conditional load would be waited for inside its branch.
  auto load_item = [&](int it, int rr, int px, int n0, int iy0, int ix0, int chunk, half8 &v, half8 &vl, int &dst, bool &live) {
    const int s = udiv_magic(rr, a.ph_magic), py = rr - s * PH;
    const int iy = iy0 + py, ix = ix0 + px;
    const int col = STRIDE == 2 ? ((px & 1) * HALF + (px >> 1)) : px;
    dst = it < patch_items ? (rr * RP + col) * PS + slot * 16 : -1;
    live = it < patch_items && iy >= 0 && iy < Hin && ix >= 0 && ix < Hin && (n0 + s) < a.n;
    const size_t safe_off = ((((size_t)(n0 < a.n ? n0 : 0) << hin_l) << hin_l) * CIN) + (size_t)((tid * 8) & ((CIN << (2 * hin_l)) - 1) & ~7);
    const size_t off = live ? (((((size_t)(n0 + s) << hin_l) + iy) << hin_l) + ix) * CIN + chunk * KC + slot * 8 : safe_off;
    v = *(const half8 *)((const _Float16 *)a.x + off);
    if constexpr (NSPLIT == 2) vl = *(const half8 *)((const char *)((const _Float16 *)a.x + off) + a.x_lo_off);
  };
  auto issue_patch = [&](int t, int chunk) {
    int tx, ty, n0;
    tile_decode(t, tx, ty, n0);
    const int iy0 = ((ty << th_l) * STRIDE) - PAD, ix0 = ((tx << tw_l) * STRIDE) - PAD;
    int rr = udiv_magic(tid / SLOTS, a.pw_magic), px = tid / SLOTS - rr * PW;  // rr = row counter over (sample, py)
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      half8 dummy;
      bool live;
      load_item(tid + u * NT, rr, px, n0, iy0, ix0, chunk, pv[u], NSPLIT == 2 ? pvl[NSPLIT == 2 ? u : 0] : dummy, pdst[u], live);
      if (!live && pdst[u] >= 0) pdst[u] |= 1 << 30;
      px += step_x; rr += step_r;
      if (px >= PW) { px -= PW; ++rr; }
    }
  };
  auto put_item = [&](half8 v, half8 vl, int dst, bool live) {
    if (!live) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { v[e] = (_Float16)0.f; vl[e] = (_Float16)0.f; }
    }
    if (dst >= 0) {
      *(half8 *)(patch + dst) = v;
      if constexpr (NSPLIT == 2) *(half8 *)(patch + a.patch_bytes + dst) = vl;
    }
  };
  auto commit_patch = [&](int t, int chunk) {
#pragma unroll
    for (int u = 0; u < UN; ++u) put_item(pv[u], NSPLIT == 2 ? pvl[NSPLIT == 2 ? u : 0] : pv[u], pdst[u] < 0 ? -1 : (pdst[u] & ~(1 << 30)), !(pdst[u] & (1 << 30)));
    if (patch_items > UN * NT) {  // oversized patch (tiny maps, many samples per tile): the rest synchronously
      int tx, ty, n0;
      tile_decode(t, tx, ty, n0);
      const int iy0 = ((ty << th_l) * STRIDE) - PAD, ix0 = ((tx << tw_l) * STRIDE) - PAD;
      for (int it = tid + UN * NT; it < patch_items; it += NT) {
        const int pix = it / SLOTS, rr = udiv_magic(pix, a.pw_magic), px = pix - rr * PW;
        half8 v, vl;
        int dst;
        bool live;
        load_item(it, rr, px, n0, iy0, ix0, chunk, v, vl, dst, live);
        put_item(v, vl, dst, live);
      }
    }
  };

  // ds_read_b128 is served in 16-lane groups {0-3,12-15,20-27} and {4-11,16-19,28-31} (+32 for the upper half,
  // MI355X_MICROARCH.md LDS table).  With PS/16 odd a group is conflict-free iff its 16 patch-pixel indices are
  // distinct mod 16, so lanes are RANKED such that each group owns 16 consecutive logical pixels: one 16-pixel row
  // segment (TW >= 16) or, for 8-wide maps, two 8-pixel rows 4/STRIDE apart (their patch rows are then
  // 4 * RP = 8 (mod 16) pixels apart because the host keeps RP = 2 (mod 4)).  Only used when a 32-pixel block lies
  // inside one sample (H*W >= 32); smaller maps keep the natural order the GAP butterfly relies on.
  const bool rank_lanes = (tw_l + th_l) >= 5;
  int pr = p;
  if (rank_lanes) pr = p < 4 ? p : p < 12 ? p + 12 : p < 16 ? p - 8 : p < 20 ? p + 8 : p < 28 ? p - 12 : p;
  const bool pair_rows = rank_lanes && tw_l == 3 && th_l >= 3;
  // tile-independent part of the lane -> pixel map, packed: x[0:5) y[5:10) slot[10:18) sample[18:30) ok[30]
  int base[WPB], lmap[WPB];
#pragma unroll
  for (int j = 0; j < WPB; ++j) {
    const int m = (wp * WPB + j) * 32 + pr;
    const bool ok = m < m_valid;
    const int mm = ok ? m : 0;
    const int x = mm & (TW - 1);
    int q = mm >> tw_l;  // row counter over (sample, y)
    if (pair_rows) {
      const int k = q & 7;
      const int kp = STRIDE == 1 ? (((k & 1) << 2) | (k >> 1)) : ((k & 4) | ((k & 1) << 1) | ((k >> 1) & 1));
      q = (q & ~7) | kp;
    }
    const int y = q & (TH - 1), sm = q >> th_l;
    base[j] = DMA ? (sm * PH + y) * RP + x                        // DMA: patch pixel index of tap (0,0) (slots are swizzled per pixel)
                  : ((sm * PH + y * STRIDE) * RP + x) * PS + h * 16;  // LDS byte offset of (pixel, tap (0,0), slot h)
    lmap[j] = x | (y << 5) | (((mm & ((1 << (tw_l + th_l)) - 1)) >> 5) << 10) | (sm << 18) | ((ok ? 1 : 0) << 30);
  }

  constexpr int CBK = CBP ? CBP : CBT;  // 32-channel blocks per PACKED cout tile
  static_assert(CBK % CBT == 0, "packed cout tile is a multiple of this kernel's tile");
  const char *wsrc = (const char *)a.w + (size_t)(ctile / (CBK / CBT)) * NCHUNK * TT * (KS * CBK * 1024) + (size_t)(ctile % (CBK / CBT)) * CBT * 1024;
  auto issue_step = [&](int chunk, int g, int buf) {
    const char *src = wsrc + (size_t)(chunk * TT + g * GT) * (KS * CBK * 1024);
    char *dst = wring + buf * WS * WCHUNK;
    // every wave issues exactly PPW instructions (the counted vmcnt below relies on it); a wave without a piece of
    // its own re-copies the last piece (same bytes to the same place: benign)
#pragma unroll
    for (int sp = 0; sp < WS; ++sp)
#pragma unroll
      for (int k = 0; k < (NPIECE + NW - 1) / NW; ++k) {
        int pi = wave + k * NW;
        pi = pi < NPIECE ? pi : NPIECE - 1;
        const int spi = CBK == CBT ? pi : (pi / CBT) * CBK + pi % CBT;  // piece index in the packed (wider) tile
        glds16(src + sp * w_lo + spi * 1024 + lane * 16, dst + sp * WCHUNK + pi * 1024);
      }
  };
  // weights that fit one step and one chunk stay resident in LDS for the life of the (persistent) workgroup
  constexpr bool W_RESIDENT = NG == 1 && NCHUNK == 1;
  // Cross-stage prefetch keeps ~UN*4 + 16 more VGPRs live through the MFMA loop.  It pays where the loop is short and
  // there is no weight ring competing for registers / the vmcnt queue (the 32-channel layers, measured 0.70 -> 0.50 ms);
  // the ring kernels lose more occupancy than they gain and load each stage's patch right before committing it.
  constexpr bool PREFETCH = W_RESIDENT;
  // Persistence (several tiles per workgroup) only where it pays: a real tile loop makes the compiler keep far more state
  // live (64@32: 79 -> 168 VGPRs, 2 -> 1 workgroups per CU).  Ring kernels run exactly one tile per workgroup (grid.x = ntiles).
  constexpr bool PERSIST = W_RESIDENT;
  static_assert(!DMA || (W_RESIDENT && NSPLIT == 1 && STRIDE == 1 && !SC), "DMA staging: resident weights, stride 1, fast arithmetic");
  if constexpr (W_RESIDENT) issue_step(0, 0, 0);

  // residual prefetch registers (16 B per lane and quad pair, see pair16)
  constexpr int NRES = WCB * WPB * 2 * XS;  // residual loads per lane
  uint4v resv[WCB][WPB][2], resl[NSPLIT == 2 ? WCB : 1][NSPLIT == 2 ? WPB : 1][2];

  // ---- DMA staging: UN LDS-DMA instructions per wave move one tile's patch; LDS position of item it is it * 16 bytes,
  // item = (patch pixel q, position pos) holds channel slot pos ^ ((q >> PIXROW_L) & (SLOTS - 1)) ----
  auto dma_piece = [&](int t, int buf, int k) {  // k-th of this wave's UN pieces of tile t
    int tx, ty, n0;
    tile_decode(t, tx, ty, n0);
    const int iy0 = (ty << th_l) - PAD, ix0 = (tx << tw_l) - PAD;
    const int npiece = a.patch_bytes >> 10, npix = (1 << spw_l) * PH * PW;
    char *dst = smem + buf * a.patch_bytes;
    {
      int piece = wave + k * NW;
      piece = piece < npiece ? piece : npiece - 1;  // every wave issues exactly UN instructions (counted vmcnt); extras re-copy the last KiB
      const int it = piece * 64 + lane;
      const int q = it / SLOTS, pos = it & (SLOTS - 1);
      const int sl = pos ^ ((q >> PIXROW_L) & (SLOTS - 1));
      const int rr = udiv_magic(q, a.pw_magic), px = q - rr * PW;
      const int s = udiv_magic(rr, a.ph_magic), py = rr - s * PH;
      const int iy = iy0 + py, ix = ix0 + px;
      const bool live = q < npix && iy >= 0 && iy < Hin && ix >= 0 && ix < Hin && (n0 + s) < a.n;
      const char *src = live ? (const char *)a.x + ((((((size_t)(n0 + s) << hin_l) + iy) << hin_l) + ix) * CIN + sl * 8) * 2
                             : (const char *)a.zero + ((((int)blockIdx.x * NT + tid) * 16) & 0xFFF0);  // spread over the zero page: no hot line
      glds16(src, dst + piece * 1024);
    }
  };
  auto dma_patch = [&](int t, int buf) {
#pragma unroll
    for (int k = 0; k < UN; ++k) dma_piece(t, buf, k);
  };

  int t = blockIdx.x;
  const int tstep = gridDim.x;
  if (!DMA && PREFETCH && t < ntiles) issue_patch(t, 0);
  int cur = 0;  // DMA: patch buffer the current tile reads
  if constexpr (DMA) {
    if (t < ntiles) dma_patch(t, 0);
    // resident weights + first patch landed (each wave waits for its own LDS-DMA, the barrier publishes them)
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  }

  int lane16 = lane * 16;
  PH_DECL;
  for (; t < ntiles; t = PERSIST ? t + tstep : ntiles) {
    PH_MARK(7);
    // The tile loop must not become a reason to keep every tile-invariant address term in registers: without this the
    // compiler hoists them all out of the loop (64@32: 79 -> 168 VGPRs, one workgroup per CU instead of two).  Passing the
    // few base values through an empty asm makes everything derived from them loop-variant again.
#pragma unroll
    for (int j = 0; j < WPB; ++j) asm volatile("" : "+v"(base[j]));
    asm volatile("" : "+v"(lane16));
    int RPt = RP, HALFt = HALF;
    asm volatile("" : "+s"(RPt), "+s"(HALFt));
    int tx, ty, n0;
    tile_decode(t, tx, ty, n0);
    int opix[WPB];  // output pixel index (flattened n,y,x) or -1
    int gidx[WPB];  // gap partial-sum row (sample * nslots + slot) or -1
#pragma unroll
    for (int j = 0; j < WPB; ++j) {
      const int lxj = lmap[j] & 31, lyj = (lmap[j] >> 5) & 31, lsl = (lmap[j] >> 10) & 255, lsj = (lmap[j] >> 18) & 4095;
      const bool ok = (lmap[j] >> 30) && (n0 + lsj) < a.n;
      const int oy = (ty << th_l) + lyj, ox = (tx << tw_l) + lxj;
      opix[j] = ok ? ((((n0 + lsj) << hout_l) + oy) << hout_l) + ox : -1;
      const int tile_in_sample = (ty << txs_l) + tx;
      gidx[j] = ok ? (n0 + lsj) * a.gap_slots + ((tile_in_sample << (tw_l + th_l)) >> 5) + lsl : -1;
    }
    const int t_next = PERSIST ? t + tstep : ntiles;
    // folded BN biases of this lane's output channels: issued here so the loads fly under the staging / MFMA phase
    // (a load placed in the epilogue is waited for on the spot)
    float4v bq[WCB][4], bsq[SC ? WCB : 1][4];
    auto load_biases = [&]() {
#pragma unroll
      for (int i = 0; i < WCB; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          bq[i][q] = *(const float4v *)(a.bias + ctile * CT + (wc * WCB + i) * 32 + 4 * h + 8 * q);
          if constexpr (SC) bsq[i][q] = *(const float4v *)(a.bias_sc + ctile * CT + (wc * WCB + i) * 32 + 4 * h + 8 * q);
        }
    };
    // early (latency hidden under the MFMA phase, +16..32 live VGPRs) where registers are not the occupancy limiter;
    // the stride-1 ring kernels load them at the epilogue (measured: 128@16 0.41 vs 0.48 ms)
    constexpr bool BIAS_EARLY = CFG_BIAS_EARLY && (W_RESIDENT || (SC && CFG_S2_BIAS_EARLY));
    // 16-wave stride-1 ring kernels: residual loads under the last weight step instead of in the epilogue
    constexpr bool RES_EARLY = CFG_RING_RES_EARLY && !W_RESIDENT && NSPLIT == 1 && NG > 1 && !SC && NW == 16;
    if constexpr (BIAS_EARLY) load_biases();

    float16v acc[WCB][WPB];
    float16v acc_sc[SCW][SPB];
#pragma unroll
    for (int i = 0; i < WCB; ++i)
#pragma unroll
      for (int j = 0; j < WPB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < SCW; ++i)
#pragma unroll
      for (int j = 0; j < SPB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_sc[i][j][r] = 0.f;

    auto load_residual = [&]() {  // residual reads (uniform branch): clamped address when the pixel is invalid
      if (!a.res) return;
#pragma unroll
      for (int i = 0; i < WCB; ++i)
#pragma unroll
        for (int j = 0; j < WPB; ++j) {
          const size_t o = (size_t)(opix[j] >= 0 ? opix[j] : 0) * COUT + ctile * CT + (wc * WCB + i) * 32 + 8 * h;
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            resv[i][j][qq] = *(const uint4v *)((const _Float16 *)a.res + o + 16 * qq);
            if constexpr (NSPLIT == 2) resl[i][j][qq] = *(const uint4v *)((const char *)((const _Float16 *)a.res + o + 16 * qq) + a.res_lo_off);
          }
        }
    };
    // what flies during the LAST weight steps of a chunk: the next stage's patch and, before the epilogue, the residual
    auto prefetch_next = [&](int chunk) {
      if constexpr (!PREFETCH) return;
      if (chunk + 1 < NCHUNK) issue_patch(t, chunk + 1);
      else if (t_next < ntiles) issue_patch(t_next, 0);
      else {  // nothing follows: keep the number of outstanding loads identical (the counted waits depend on it)
#pragma unroll
        for (int u = 0; u < UN; ++u) {
          pv[u] = *(const half8 *)((const _Float16 *)a.x + (size_t)(tid & 255) * 8);
          if constexpr (NSPLIT == 2) pvl[u] = pv[u];
          pdst[u] = -1;
        }
      }
      if (W_RESIDENT && chunk + 1 == NCHUNK) load_residual();
    };
    // loads in flight behind the ring's last counted wait of a chunk (prefetch_next): patch items (+ residual)
    const int PF_PATCH = UN * XS;

    bool all_ok = true;  // DMA: every lane of this wave stores (the counted wait at the end of the tile relies on it)
    if constexpr (DMA) {
#pragma unroll
      for (int j = 0; j < WPB; ++j) all_ok = all_ok && __all(opix[j] >= 0);
      load_residual();  // OLDER than the DMA below: waiting for it in the epilogue does not drain the DMA
      // next tile's patch into the other buffer (all waves passed the previous tile's closing barrier, so nobody reads it);
      // the last tile re-fetches itself so that the number of outstanding operations does not depend on the path
      PH_MARK(0);
      const int t_dma = t_next < ntiles ? t_next : t;
      if constexpr (!(ASM_PIPE && CFG_DMA_SPREAD)) dma_patch(t_dma, cur ^ 1);
      PH_MARK(1);
      const char *pb = smem + cur * a.patch_bytes;
      if constexpr (ASM_PIPE) {
        constexpr int NITEM = TAPS * KS, NR = WCB + WPB;
        const uint32_t wb = lds0 + (uint32_t)(wring - smem) + (wc * WCB) * 1024 + lane16;
        const uint32_t pl = lds0 + cur * a.patch_bytes;
        half8 fa[3][WCB], fb[3][WPB];
        uint32_t rowa[WPB], hs[WPB];  // per tap: byte address of the pixel, (h * 16) ^ (swizzle << 4)
        auto issue = [&](auto ic) {
          constexpr int item = decltype(ic)::value, sl = item % 3, tt = item / KS, ks = item % KS;
          if constexpr (ks == 0) {
            constexpr int dy = tt / 3, dx = tt % 3;
#pragma unroll
            for (int j = 0; j < WPB; ++j) {
              const int q = base[j] + dy * RPt + dx;
              rowa[j] = pl + q * (KC * 2);
              hs[j] = (h * 16) ^ (((q >> PIXROW_L) & (SLOTS - 1)) << 4);
            }
          }
          static_for<WCB>([&](auto ii) {
            constexpr int i = decltype(ii)::value, off = ((tt * KS + ks) * CBT + i) * 1024;
            lds_read128<off & 0xFFFF>(fa[sl][i], wb + (off & ~0xFFFF));
          });
          static_for<WPB>([&](auto jj) { lds_read128<0>(fb[sl][decltype(jj)::value], rowa[decltype(jj)::value] + (hs[decltype(jj)::value] ^ (ks * 32))); });
        };
        issue(std::integral_constant<int, 0>{});
        issue(std::integral_constant<int, 1>{});
        static_for<NITEM>([&](auto ic) {
          constexpr int item = decltype(ic)::value, sl = item % 3;
          // the next tile's patch pieces are issued one at a time under the first MFMAs instead of as a burst in front of
          // them (all eight waves issuing 6 KiB each at once kept the wave at the issue for 18 % of the tile time)
          if constexpr (CFG_DMA_SPREAD && item % CFG_DMA_SPREAD == 0 && item / CFG_DMA_SPREAD < UN) dma_piece(t_dma, cur ^ 1, item / CFG_DMA_SPREAD);
          if constexpr (item + 2 < NITEM) issue(std::integral_constant<int, item + 2>{});
          constexpr int younger = (NITEM - 1 - item < 2 ? NITEM - 1 - item : 2) * NR;
          lds_wait<younger>();
#pragma unroll
          for (int i = 0; i < WCB; ++i) lds_touch(fa[sl][i]);
#pragma unroll
          for (int j = 0; j < WPB; ++j) lds_touch(fb[sl][j]);
#pragma unroll
          for (int i = 0; i < WCB; ++i)
#pragma unroll
            for (int j = 0; j < WPB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[sl][i], fb[sl][j], acc[i][j], 0, 0, 0);
        });
      } else {
        half8 af[2][WCB], bf[2][WPB];
        auto load_frags = [&](int item, int sl) {
          const int tt = item / KS, ks = item - tt * KS;
          const int dy = tt / 3, dx = tt - dy * 3;
#pragma unroll
          for (int i = 0; i < WCB; ++i) af[sl][i] = *(const half8 *)(wring + ((tt * KS + ks) * CBT + wc * WCB + i) * 1024 + lane16);
#pragma unroll
          for (int j = 0; j < WPB; ++j) {
            const int q = base[j] + dy * RPt + dx;
            const int sw = (q >> PIXROW_L) & (SLOTS - 1);
            bf[sl][j] = *(const half8 *)(pb + q * (KC * 2) + (((ks * 2 + h) ^ sw) << 4));
          }
        };
        load_frags(0, 0);
#pragma unroll
        for (int item = 0; item < TAPS * KS; ++item) {
          const int cb = item & 1;
          if (item + 1 < TAPS * KS) load_frags(item + 1, cb ^ 1);
#pragma unroll
          for (int i = 0; i < WCB; ++i)
#pragma unroll
            for (int j = 0; j < WPB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cb][i], bf[cb][j], acc[i][j], 0, 0, 0);
        }
      }
      PH_MARK(3);
    }
    for (int chunk = 0; chunk < (DMA ? 0 : NCHUNK); ++chunk) {
      if constexpr (!PREFETCH) issue_patch(t, chunk);
      PH_MARK(0);
      commit_patch(t, chunk);  // registers -> LDS (the compiler waits for exactly these loads here)
      PH_MARK(1);
      if constexpr (!W_RESIDENT) {
        issue_step(chunk, 0, 0);
#pragma unroll
        for (int d = 1; d < PFD; ++d)
          if (d < NG) issue_step(chunk, d, d);
      }
      if constexpr (NBUF > 1) {
        // step 0 must have landed; the PFD-1 younger steps may stay in flight.  LDS-DMA data is ordered for another
        // wave's ds_read only by the ISSUING wave's vmcnt followed by a barrier; the patch ds_writes need lgkmcnt(0).
        constexpr int INFL0 = (PFD - 1 < NG - 1 ? PFD - 1 : NG - 1) * PPW;
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(INFL0) : "memory");
      } else if constexpr (W_RESIDENT) {
        // The resident weights' LDS-DMA is older than the first patch loads, so the wait commit_patch needed for those
        // already covers it.  Only the patch ds_writes must be visible: lgkmcnt(0) + raw barrier -- a __syncthreads()
        // would add vmcnt(0) and expose the previous tile's epilogue STORES (vmcnt counts stores on CDNA4).
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        prefetch_next(chunk);  // single weight step: the whole MFMA phase + epilogue hide these loads
      } else {
        __syncthreads();  // drains vmcnt (LDS-DMA landed) and makes the patch visible
        prefetch_next(chunk);
      }
      PH_MARK(2);

      int cur_buf = 0;
#pragma unroll 1
      for (int g = 0; g < NG; ++g) {
        char *wcur = wring + cur_buf * WS * WCHUNK;
        // last weight step of the tile: no ring wait follows it any more, so loads issued now fly under its MFMAs without
        // being drained by a counted vmcnt -- the residual arrives before the epilogue needs it
        if constexpr (RES_EARLY) {
          if (chunk == NCHUNK - 1 && g == NG - 1) load_residual();
        }
        if constexpr (NBUF > 1) {
          // buffer (g + PFD) % NBUF == (g - 1) % NBUF was last read in step g-1; every wave has passed that step's barrier
          if (g + PFD < NG) {
            int nb = cur_buf + PFD;
            if (nb >= NBUF) nb -= NBUF;
            issue_step(chunk, g + PFD, nb);
          }
          // after the LAST weight step of this chunk has been issued: start the next stage's loads (they are younger than
          // every ring operation still needed, so the counted waits below simply add them)
          if (g + PFD == NG - 1 || (NG <= PFD && g == 0)) prefetch_next(chunk);
        }
        // Fragment reads run ONE (tap, k-step) ahead of the MFMAs that consume them (register double buffer), so the
        // ds_read latency of item i+1 hides under the MFMAs of item i instead of serialising read -> wait -> MFMA.
        auto tap_off = [&](int tt) -> int {
          const int tp = g * GT + tt;
          if (TAPS != 9) return 0;
          const int te = (SC && tp == TAPS) ? 4 : tp;  // the 1x1 stride-2 shortcut reads the centre tap's pixel
          const int dy = te / 3, dx = te - dy * 3;
          return STRIDE == 2 ? (dy * RPt + (dx & 1) * HALFt + (dx >> 1)) * PS : (dy * RPt + dx) * PS;
        };
        if constexpr (ASM_PIPE) {
          // FDR + 1 fragment slots: the reads of the next FDR items are in flight while the MFMAs of item i run (16-wave
          // workgroups have four waves per SIMD to cover an item and a 128-VGPR budget: one item ahead)
          constexpr int NITEM = GT * KS, NR = WCB + WPB, FDR = (NW == 16 && CFG_RING_FD16) ? CFG_RING_FD16 : 2;
          const uint32_t wb = lds0 + (uint32_t)(wcur - smem) + (wc * WCB) * 1024 + lane16;
          half8 fa[FDR + 1][WCB], fb[FDR + 1][WPB];
          uint32_t pbt[WPB];
          auto issue = [&](auto ic) {
            constexpr int item = decltype(ic)::value, sl = item % (FDR + 1), tt = KMAJ ? item % GT : item / KS, ks = KMAJ ? item / GT : item % KS;
            if constexpr (KMAJ || ks == 0) {
              const int toff = tap_off(tt);
#pragma unroll
              for (int j = 0; j < WPB; ++j) pbt[j] = lds0 + base[j] + toff;
            }
            static_for<WCB>([&](auto ii) {
              constexpr int i = decltype(ii)::value, off = ((tt * KS + ks) * CBT + i) * 1024;
              lds_read128<off & 0xFFFF>(fa[sl][i], wb + (off & ~0xFFFF));
            });
            static_for<WPB>([&](auto jj) { lds_read128<ks * 32>(fb[sl][decltype(jj)::value], pbt[decltype(jj)::value]); });
          };
          issue(std::integral_constant<int, 0>{});
          if constexpr (NITEM > 1 && FDR > 1) issue(std::integral_constant<int, 1>{});
          static_for<NITEM>([&](auto ic) {
            constexpr int item = decltype(ic)::value, sl = item % (FDR + 1);
            if constexpr (item + FDR < NITEM) issue(std::integral_constant<int, item + FDR>{});
            constexpr int younger = (NITEM - 1 - item < FDR ? NITEM - 1 - item : FDR) * NR;
            lds_wait<younger>();
#pragma unroll
            for (int i = 0; i < WCB; ++i) lds_touch(fa[sl][i]);
#pragma unroll
            for (int j = 0; j < WPB; ++j) lds_touch(fb[sl][j]);
            const bool is_sc = SC && (g * GT + (KMAJ ? item % GT : item / KS)) == TAPS;
            if (is_sc) {
              if constexpr (SC) {
#pragma unroll
                for (int i = 0; i < WCB; ++i)
#pragma unroll
                  for (int j = 0; j < WPB; ++j) acc_sc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[sl][i], fb[sl][j], acc_sc[i][j], 0, 0, 0);
              }
            } else {
#pragma unroll
              for (int i = 0; i < WCB; ++i)
#pragma unroll
                for (int j = 0; j < WPB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[sl][i], fb[sl][j], acc[i][j], 0, 0, 0);
            }
          });
        } else {
        static_assert(!KMAJ, "k-step-major items: asm-pipelined path only");
        half8 af[2][WCB], bf[2][WPB], afl[2][WS == 2 ? WCB : 1], bfl[2][NSPLIT == 2 ? WPB : 1];
        auto load_frags = [&](int item, int sl) {
          const int tt = item / KS, ks = item - tt * KS;
          const int toff = tap_off(tt);
#pragma unroll
          for (int i = 0; i < WCB; ++i) {
            af[sl][i] = *(const half8 *)(wcur + ((tt * KS + ks) * CBT + wc * WCB + i) * 1024 + lane16);
            if constexpr (WS == 2) afl[sl][i] = *(const half8 *)(wcur + WCHUNK + ((tt * KS + ks) * CBT + wc * WCB + i) * 1024 + lane16);
          }
#pragma unroll
          for (int j = 0; j < WPB; ++j) {
            bf[sl][j] = *(const half8 *)(patch + base[j] + toff + ks * 32);
            if constexpr (NSPLIT == 2) bfl[sl][j] = *(const half8 *)(patch + a.patch_bytes + base[j] + toff + ks * 32);
          }
        };
        load_frags(0, 0);
#pragma unroll
        for (int item = 0; item < GT * KS; ++item) {
          const int cur = item & 1;
          if (item + 1 < GT * KS) load_frags(item + 1, cur ^ 1);
          const bool is_sc = SC && (g * GT + item / KS) == TAPS;
          if (is_sc) {
            if constexpr (SC) {
#pragma unroll
              for (int i = 0; i < WCB; ++i)
#pragma unroll
                for (int j = 0; j < WPB; ++j) {
                  acc_sc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cur][i], bf[cur][j], acc_sc[i][j], 0, 0, 0);
                  if constexpr (NSPLIT == 2) acc_sc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cur][i], bfl[cur][j], acc_sc[i][j], 0, 0, 0);
                  if constexpr (WS == 2) acc_sc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afl[cur][i], bf[cur][j], acc_sc[i][j], 0, 0, 0);
                }
            }
          } else {
#pragma unroll
            for (int i = 0; i < WCB; ++i)
#pragma unroll
              for (int j = 0; j < WPB; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
                if constexpr (NSPLIT == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[cur][i], bfl[cur][j], acc[i][j], 0, 0, 0);
                if constexpr (WS == 2) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(afl[cur][i], bf[cur][j], acc[i][j], 0, 0, 0);
              }
          }
        }
        }  // !ASM_PIPE
        PH_MARK(3);
        if constexpr (NBUF > 1) {
          if (g + 1 < NG) {
            // step g+1 must have landed; steps g+2 .. min(g+PFD, NG-1) may stay in flight, and so may the next stage's
            // loads once they have been issued (they are younger): counted vmcnt + raw barrier (__syncthreads() would
            // drain the whole queue with vmcnt(0))
            const int last = g + PFD < NG - 1 ? g + PFD : NG - 1;
            const int infl = last - (g + 1);
            const bool pf_out = PREFETCH && g + PFD >= NG - 1;  // prefetch_next already issued
            const bool with_res = false;  // ring kernels read the residual in the epilogue (register budget)
            if (!pf_out) {
              if (infl >= 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * PPW) : "memory");
              else if (infl == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(1 * PPW) : "memory");
              else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            } else if (with_res) {  // infl == 0 here: every ring step has been issued before the prefetch
              asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(UN * XS + NRES) : "memory");
            } else {
              asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(UN * XS) : "memory");
            }
            if (++cur_buf == NBUF) cur_buf = 0;
          }
        }
      }
      PH_MARK(4);
      // everyone is done reading this stage's patch / weights before the next commit overwrites them
      if (chunk + 1 < NCHUNK) __builtin_amdgcn_s_barrier();
      PH_MARK(4);
    }
    (void)PF_PATCH;
  // ---- epilogue: + bias (+ residual) (ReLU) -> fp16 NHWC (8 B per register quad) and/or fp32 GAP partials ----
  if constexpr (!BIAS_EARLY) load_biases();
  if constexpr (!W_RESIDENT && !RES_EARLY) {
    if (a.res) {  // all residual reads together (16 B per lane and quad pair), clamped address when the pixel is invalid
#pragma unroll
      for (int i = 0; i < WCB; ++i)
#pragma unroll
        for (int j = 0; j < WPB; ++j) {
          const size_t o = (size_t)(opix[j] >= 0 ? opix[j] : 0) * COUT + ctile * CT + (wc * WCB + i) * 32 + 8 * h;
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            resv[i][j][qq] = *(const uint4v *)((const _Float16 *)a.res + o + 16 * qq);
            if constexpr (NSPLIT == 2) resl[i][j][qq] = *(const uint4v *)((const char *)((const _Float16 *)a.res + o + 16 * qq) + a.res_lo_off);
          }
        }
    }
  }
  conv_epilogue<COUT, CT, WCB, WPB, SC, XS>(a, ctile, wc, h, p, opix, gidx, acc, acc_sc, bq, bsq, resv, resl);
    // the next tile's commit overwrites the patch: every wave must be done reading it (and the GAP / stores above
    // do not touch LDS)
    PH_MARK(5);
    if constexpr (DMA) {
      if (t_next < ntiles) {
        // the next tile's patch (this wave's DMA share) has landed, and -- after the barrier -- every wave is done reading the
        // current buffer.  The output stores above are younger than the DMA and may stay in flight when their number is known.
        if (a.gap || !all_ok) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(WCB * WPB * 2) : "memory");
        cur ^= 1;
      }
    } else if (t_next < ntiles) __builtin_amdgcn_s_barrier();
    PH_MARK(6);
  }
  PH_FLUSH((CIN / 32) + (STRIDE == 2 ? 8 : 0));
}


// ---------------------------------------------------------------------------------------------
// Weight-ring conv with LDS-DMA patch staging (fast arithmetic; the 64..256-channel layers of the 128x128 model).
// Same tiling, weight packing, lane ranking and epilogue as conv_mfma_kernel, but
//   * the input patch of a stage (tile, 32/64-channel chunk) is written by LDS-DMA into one of two unpadded,
//     XOR-swizzled buffers (see the DMA mode of conv_mfma_kernel) while the previous stage computes: no register
//     staging, no commit phase, a whole stage of patch traffic in flight;
//   * vmcnt retires in order, so a wave that had both kinds of DMA in its queue would drain the long-lived patch DMA
//     at every weight-step wait.  The waves therefore split the ROLES: the first half issues the weight ring (and
//     waits for it every step), the second half issues the patch pieces (and waits once per stage).  All waves compute;
//   * the weight ring runs continuously across chunks and tiles (no drain / refill per chunk), workgroups are
//     persistent over tiles, fragment reads are asm-pipelined (lds_read128).
// ---------------------------------------------------------------------------------------------
//   * NWL > 0: the LDS-DMA is issued by NWL extra LOADER waves that do nothing else (first half: weight ring, second
//     half: patch).  The compute waves' vmcnt queues then hold only their own residual prefetch and output stores, so
//     the residual can be prefetched a chunk ahead and a store never sits in front of a ring wait; NWL == 0: the
//     compute waves double as loaders (first half ring, second half patch).
//   * FD: fragment reads run FD (1 or 2) items ahead of the MFMAs (FD + 1 register slots of WCB + WPB fragments).
template <int CIN, int COUT, int STRIDE, bool SC, int KC, int WCB, int WPB, int WAVES_C, int WAVES_P, int GT, int RB, int UNP, int MINW, int NWL, int FD>
__global__ __launch_bounds__(64 * (WAVES_C * WAVES_P + NWL), MINW) void conv_ring_dma_kernel(const ConvArgs a) {
  constexpr int TAPS = 9, NSPLIT = 1;
  constexpr int NCHUNK = CIN / KC, KS = KC / 16, SLOTS = KC / 8, PIXROW_L = KC == 64 ? 1 : 2;
  constexpr int CBT = WCB * WAVES_C, CT = 32 * CBT, NW = WAVES_C * WAVES_P, NT = 64 * NW;
  constexpr int TT = TAPS + (SC ? 1 : 0), NG = TT / GT;
  constexpr int WCHUNK = GT * KS * CBT * 1024, NBUF = RB, PFD = RB - 1, NPIECE = WCHUNK / 1024;
  constexpr int NWR = NWL ? NWL / 2 : NW / 2, NWP = NWL ? NWL - NWR : NW - NWR;  // ring waves / patch waves
  constexpr int LW0 = NWL ? NW : 0;              // first loader wave: ring waves [LW0, LW0 + NWR), patch waves follow
  static_assert(NWL == 0 || NWL >= 2, "loader waves: at least one ring and one patch wave");
  static_assert(FD == 1 || FD == 2, "fragment prefetch distance");
  constexpr int PPWR = (NPIECE + NWR - 1) / NWR;  // ring LDS-DMA instructions per ring wave and step
  constexpr int PPS = (UNP + NG - 1) / NG;        // patch pieces a patch wave issues per weight step
  constexpr int SCW = SC ? WCB : 1, SPB = SC ? WPB : 1;
  static_assert(CIN % KC == 0 && (KC == 32 || KC == 64) && TT % GT == 0 && COUT % CT == 0, "tiling");
  static_assert(NG >= 2 && RB >= 2 && RB <= 4 && PFD <= NG && NW >= 2, "ring");
  static_assert(!SC || STRIDE == 2, "shortcut rides on stride-2 convs");
  static_assert(WCHUNK <= 65536, "fragment offsets are ds_read immediates");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *ring = smem + 2 * a.patch_bytes;
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % WAVES_C, wp = wave / WAVES_C;
  const int p = lane & 31, h = lane >> 5;
  const bool loader = NWL && wave >= NW;                         // a pure loader wave (no MFMA, no epilogue)
  const bool ring_wave = wave >= LW0 && wave < LW0 + NWR;
  const bool patch_wave = wave >= LW0 + NWR && wave < LW0 + NWR + NWP;
  const int lane16 = lane * 16;

  const int tw_l = a.tw_l, th_l = a.th_l, spw_l = a.spw_l, hout_l = a.hout_l, hin_l = a.hin_l;
  const int TW = 1 << tw_l, TH = 1 << th_l, Hin = 1 << hin_l;
  const int txs_l = hout_l - tw_l, tys_l = hout_l - th_l;
  const int PH = a.ph, PW = a.pw, RP = a.rp, HALF = a.half;
  const int m_valid = 1 << (tw_l + th_l + spw_l);
  const int ctile = blockIdx.y, ntiles = a.ntiles;
  auto tile_decode = [&](int t, int &tx, int &ty, int &n0) {  // XCD-contiguous tile order, see conv_mfma_kernel
    const int q = ntiles >> 3, r = ntiles & 7, xcd = t & 7;
    const int mt = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
    tx = mt & ((1 << txs_l) - 1);
    ty = (mt >> txs_l) & ((1 << tys_l) - 1);
    n0 = (mt >> (txs_l + tys_l)) << spw_l;
  };

  // lane -> pixel map (lane ranking: see conv_mfma_kernel); base[j] = patch pixel index of tap (0,0)
  const bool rank_lanes = (tw_l + th_l) >= 5;
  int pr = p;
  if (rank_lanes) pr = p < 4 ? p : p < 12 ? p + 12 : p < 16 ? p - 8 : p < 20 ? p + 8 : p < 28 ? p - 12 : p;
  const bool pair_rows = rank_lanes && tw_l == 3 && th_l >= 3;
  int base[WPB], lmap[WPB];
#pragma unroll
  for (int j = 0; j < WPB; ++j) {
    const int m = (wp * WPB + j) * 32 + pr;
    const bool ok = m < m_valid;
    const int mm = ok ? m : 0;
    const int x = mm & (TW - 1);
    int q = mm >> tw_l;
    if (pair_rows) {
      const int k = q & 7;
      const int kp = STRIDE == 1 ? (((k & 1) << 2) | (k >> 1)) : ((k & 4) | ((k & 1) << 1) | ((k >> 1) & 1));
      q = (q & ~7) | kp;
    }
    const int y = q & (TH - 1), sm = q >> th_l;
    base[j] = (sm * PH + y * STRIDE) * RP + x;
    lmap[j] = x | (y << 5) | (((mm & ((1 << (tw_l + th_l)) - 1)) >> 5) << 10) | (sm << 18) | ((ok ? 1 : 0) << 30);
  }

  // ---- weight ring: global step sequence (chunk, g) cycles over the chunks whatever the tile ----
  int t = blockIdx.x;
  const int tstep = gridDim.x;
  if (t >= ntiles) return;
#if defined(CFG_STAGGER_PHASES)
  stagger_start(CFG_STAGGER_PHASES, CFG_STAGGER_UNITS);
#endif
  int steps_to_issue = ((ntiles - 1 - t) / tstep + 1) * NCHUNK * NG;
  int ci = 0, gi = 0, slot_wr = 0;
  int ahead = 0;  // ring steps issued and not yet computed (the current one included)
  const char *wsrc = (const char *)a.w + (size_t)ctile * NCHUNK * TT * (KS * CBT * 1024);
  auto issue_ring = [&]() {  // next ring step, if any is left (all waves keep the counters, ring waves move the data)
    if (steps_to_issue <= 0) return false;
    if (ring_wave) {
      const char *src = wsrc + (size_t)(ci * TT + gi * GT) * (KS * CBT * 1024);
      char *dst = ring + slot_wr * WCHUNK;
#pragma unroll
      for (int k = 0; k < PPWR; ++k) {
        int pi = (wave - LW0) + k * NWR;
        pi = pi < NPIECE ? pi : NPIECE - 1;  // every ring wave issues exactly PPWR instructions per step (counted vmcnt)
        glds16(src + pi * 1024 + lane16, dst + pi * 1024);
      }
    }
    --steps_to_issue;
    ++ahead;
    if (++gi == NG) { gi = 0; if (++ci == NCHUNK) ci = 0; }
    if (++slot_wr == NBUF) slot_wr = 0;
    return true;
  };

  // ---- patch pieces (patch waves): LDS position of item it is it * 16 bytes; item = (patch pixel q = row * RP + col,
  // position pos) holds channel slot pos ^ ((q >> PIXROW_L) & (SLOTS - 1)); stride 2 keeps even / odd columns in halves ----
  const int npiece = a.patch_bytes >> 10;
  auto dma_piece = [&](int tp, int chunk, int buf, int k) {
    const int piece = (wave - LW0 - NWR) + k * NWP;
    if (piece >= npiece) return;  // wave-uniform; patch waves wait with vmcnt(0), nothing counts their instructions
    int tx, ty, n0;
    tile_decode(tp, tx, ty, n0);
    const int iy0 = ((ty << th_l) * STRIDE) - 1, ix0 = ((tx << tw_l) * STRIDE) - 1;
    const int it = piece * 64 + lane;
    const int q = it / SLOTS, pos = it & (SLOTS - 1);
    const int sl = pos ^ ((q >> PIXROW_L) & (SLOTS - 1));
    const int rr = udiv_magic(q, a.rp_magic), col = q - rr * RP;
    const int px = STRIDE == 2 ? (col < HALF ? 2 * col : 2 * (col - HALF) + 1) : col;
    const int s = udiv_magic(rr, a.ph_magic), py = rr - s * PH;
    const int iy = iy0 + py, ix = ix0 + px;
    const bool live = s < (1 << spw_l) && px < PW && iy >= 0 && iy < Hin && ix >= 0 && ix < Hin && (n0 + s) < a.n;
    const char *src = live ? (const char *)a.x + ((((((size_t)(n0 + s) << hin_l) + iy) << hin_l) + ix) * CIN + chunk * KC + sl * 8) * 2
                           : (const char *)a.zero + ((((int)blockIdx.x * NT + tid) * 16) & 0xFFF0);
    glds16(src, smem + buf * a.patch_bytes + piece * 1024);
  };

  // ---- prologue: first stage's patch, first PFD ring steps ----
  if constexpr (NWL > 0) {
    for (int c = tid; c < CT; c += NT + 64 * NWL) {
      ((float *)(ring + NBUF * WCHUNK))[c] = a.bias[ctile * CT + c];
      if constexpr (SC) ((float *)(ring + NBUF * WCHUNK))[CT + c] = a.bias_sc[ctile * CT + c];
    }
  }
  if (patch_wave) {
#pragma unroll
    for (int k = 0; k < UNP; ++k) dma_piece(t, 0, 0, k);
  }
#pragma unroll
  for (int d = 0; d < PFD; ++d) issue_ring();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  int cur = 0, slot_rd = 0;
  uint4v resv[WCB][WPB][2], resl[1][1][2];
  float4v bq[WCB][4], bsq[SC ? WCB : 1][4];
  auto load_biases = [&]() {
#pragma unroll
    for (int i = 0; i < WCB; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bq[i][q] = *(const float4v *)(a.bias + ctile * CT + (wc * WCB + i) * 32 + 4 * h + 8 * q);
        if constexpr (SC) bsq[i][q] = *(const float4v *)(a.bias_sc + ctile * CT + (wc * WCB + i) * 32 + 4 * h + 8 * q);
      }
  };
  // loader-wave variant: the biases of this (persistent) workgroup's cout tile sit in LDS behind the ring, written before
  // the prologue barrier, and are read in the epilogue (32 VGPRs less than keeping them in registers)
  const float *bias_lds = (const float *)(ring + NBUF * WCHUNK);
  int opix[WPB], gidx[WPB];
  auto load_residual = [&]() {
    if (!a.res) return;
#pragma unroll
    for (int i = 0; i < WCB; ++i)
#pragma unroll
      for (int j = 0; j < WPB; ++j) {
        const size_t o = (size_t)(opix[j] >= 0 ? opix[j] : 0) * COUT + ctile * CT + (wc * WCB + i) * 32 + 8 * h;
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) resv[i][j][qq] = *(const uint4v *)((const _Float16 *)a.res + o + 16 * qq);
      }
  };
  for (; t < ntiles; t += tstep) {
    int tx, ty, n0;
    tile_decode(t, tx, ty, n0);
#pragma unroll
    for (int j = 0; j < WPB; ++j) {
      const int lxj = lmap[j] & 31, lyj = (lmap[j] >> 5) & 31, lsl = (lmap[j] >> 10) & 255, lsj = (lmap[j] >> 18) & 4095;
      const bool ok = (lmap[j] >> 30) && (n0 + lsj) < a.n;
      const int oy = (ty << th_l) + lyj, ox = (tx << tw_l) + lxj;
      opix[j] = ok ? ((((n0 + lsj) << hout_l) + oy) << hout_l) + ox : -1;
      const int tile_in_sample = (ty << txs_l) + tx;
      gidx[j] = ok ? (n0 + lsj) * a.gap_slots + ((tile_in_sample << (tw_l + th_l)) >> 5) + lsl : -1;
    }
    const int t_next = t + tstep;
    float16v acc[WCB][WPB], acc_sc[SCW][SPB];
#pragma unroll
    for (int i = 0; i < WCB; ++i)
#pragma unroll
      for (int j = 0; j < WPB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < SCW; ++i)
#pragma unroll
      for (int j = 0; j < SPB; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc_sc[i][j][r] = 0.f;

#pragma unroll 1
    for (int chunk = 0; chunk < NCHUNK; ++chunk) {
      const bool in_tile = chunk + 1 < NCHUNK;
      const bool has_next = in_tile || t_next < ntiles;
      const int tn = in_tile ? t : t_next, cn = in_tile ? chunk + 1 : 0;
      const uint32_t pl = lds0 + cur * a.patch_bytes;
#pragma unroll 1
      for (int g = 0; g < NG; ++g) {
        issue_ring();  // step PFD ahead of the one computed now (into the slot read in step g-1)
        if (patch_wave && has_next) {
#pragma unroll
          for (int k = 0; k < PPS; ++k)
            if (g * PPS + k < UNP) dma_piece(tn, cn, cur ^ 1, g * PPS + k);
        }
        if (NWL && !loader && chunk == NCHUNK - 1 && g == 0) load_residual();  // a whole chunk ahead of the epilogue
        // ---- MFMAs of this step: fragment reads two items ahead, counted waits ----
        if (!loader) {
          constexpr int NITEM = GT * KS, NR = WCB + WPB;
          const uint32_t wb = lds0 + 2 * a.patch_bytes + slot_rd * WCHUNK + (wc * WCB) * 1024 + lane16;
          half8 fa[FD + 1][WCB], fb[FD + 1][WPB];
          uint32_t rowa[WPB], hs[WPB];
          auto issue = [&](auto ic) {
            constexpr int item = decltype(ic)::value, sl = item % (FD + 1), tt = item / KS, ks = item % KS;
            if constexpr (ks == 0) {
              const int tp = g * GT + tt;
              const int te = (SC && tp == TAPS) ? 4 : tp;  // the 1x1 stride-2 shortcut reads the centre tap's pixel
              const int dy = te / 3, dx = te - dy * 3;
              const int tq = STRIDE == 2 ? dy * RP + (dx & 1) * HALF + (dx >> 1) : dy * RP + dx;
#pragma unroll
              for (int j = 0; j < WPB; ++j) {
                const int q = base[j] + tq;
                rowa[j] = pl + q * (KC * 2);
                hs[j] = (h * 16) ^ (((q >> PIXROW_L) & (SLOTS - 1)) << 4);
              }
            }
            static_for<WCB>([&](auto ii) {
              constexpr int i = decltype(ii)::value;
              lds_read128<((tt * KS + ks) * CBT + i) * 1024>(fa[sl][i], wb);
            });
            static_for<WPB>([&](auto jj) { lds_read128<0>(fb[sl][decltype(jj)::value], rowa[decltype(jj)::value] + (hs[decltype(jj)::value] ^ (ks * 32))); });
          };
          issue(std::integral_constant<int, 0>{});
          if constexpr (NITEM > 1 && FD > 1) issue(std::integral_constant<int, 1>{});
          static_for<NITEM>([&](auto ic) {
            constexpr int item = decltype(ic)::value, sl = item % (FD + 1);
            if constexpr (item + FD < NITEM) issue(std::integral_constant<int, item + FD>{});
            constexpr int younger = (NITEM - 1 - item < FD ? NITEM - 1 - item : FD) * NR;
            lds_wait<younger>();
#pragma unroll
            for (int i = 0; i < WCB; ++i) lds_touch(fa[sl][i]);
#pragma unroll
            for (int j = 0; j < WPB; ++j) lds_touch(fb[sl][j]);
            const bool is_sc = SC && (g * GT + item / KS) == TAPS;
            if (is_sc) {
              if constexpr (SC) {
#pragma unroll
                for (int i = 0; i < WCB; ++i)
#pragma unroll
                  for (int j = 0; j < WPB; ++j) acc_sc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[sl][i], fb[sl][j], acc_sc[i][j], 0, 0, 0);
              }
            } else {
#pragma unroll
              for (int i = 0; i < WCB; ++i)
#pragma unroll
                for (int j = 0; j < WPB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[sl][i], fb[sl][j], acc[i][j], 0, 0, 0);
            }
          });
        }
        // ---- end of step: the next ring step has landed (ring waves; LDS-DMA is published by the issuing wave's vmcnt +
        // a barrier); at the end of a stage also the next patch (patch waves) ----
        // of the `ahead` issued steps this one is done and the next must have landed: the ahead - 2 younger ones may fly
        // (anything else in the queue -- epilogue stores -- only makes the counted wait stricter)
        if (ring_wave) {
          if (PFD >= 3 && ahead >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPWR) : "memory");
          else if (PFD >= 2 && ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPWR) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (patch_wave && g == NG - 1) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        asm volatile("s_barrier" ::: "memory");
        --ahead;
        if (++slot_rd == NBUF) slot_rd = 0;
      }
      cur ^= 1;
    }

    // ---- epilogue.  Without loader waves the biases and the residual are read here (the CU's other workgroup covers the
    // latency); with them the biases were loaded once per kernel and the residual a chunk ago ----
    if (loader) continue;
    if constexpr (NWL == 0) {
      load_biases();
      load_residual();
    }
    conv_epilogue<COUT, CT, WCB, WPB, SC, NSPLIT>(a, ctile, wc, h, p, opix, gidx, acc, acc_sc, bq, bsq, resv, resl, NWL ? bias_lds : nullptr);
  }
}



// A kernel argument re-read from the kernarg segment AT THE POINT OF USE (one s_load, base pointer made opaque): for pointers that are
// needed once per tile in a kernel whose scalar registers are exhausted -- kept live they are spilled to a VGPR lane and from there to
// scratch, and every scratch reload costs a round trip plus an s_waitcnt vmcnt(0) that drains the LDS-DMA in flight.
template <class T> __device__ __forceinline__ T karg_reload(size_t off) {
  const __attribute__((address_space(4))) char *ka = (const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(ka));
  return *(const __attribute__((address_space(4))) T *)(ka + off);
}
#define KARG(type, member) karg_reload<type>(offsetof(ChainArgs, member))

// ---------------------------------------------------------------------------------------------
// Fused chain of stride-1 3x3 convs on WHOLE samples (the BasicBlock tail of the 128@16 stage, arch:52-57):
//     b0 = relu(bn2(conv2(t)) + sc) ;  t1 = relu(bn1(conv1(b0))) ;  out = relu(bn2(conv2(t1)) + b0)  (+ GAP, arch:288)
// One 16-wave workgroup owns one sample: the 16 x 16 x 128 activation (64 KiB fp16) lives in LDS for the whole chain,
// every conv reads it from there and its epilogue writes the next activation back IN PLACE (all waves have passed the
// last weight step's barrier, so nobody reads the buffer any more); b0 -- the residual of the last conv -- stays in the
// registers of the wave that produced it (a wave's output tile covers the same pixels x channels in every conv).  HBM
// sees t and sc in, out (or only the GAP sums) out: no intermediate is written or re-read, no patch is staged per
// conv, and three launches become one.
//   * no halo: a tile is a whole sample, so a tap that leaves the map is conv padding.  Such a lane reads an address beyond the
//     workgroup's LDS allocation, which returns zeros on gfx950 (OOBZ; probed at mlt_init, which selects the masked form of the kernel otherwise) -- the earlier form, the
//     lane's own pixel ANDed with a zero mask, cost 8 VALU ops per k-step in the loop where issue slots are scarce (4-6 %);
//     the buffer is unpadded, 16-byte channel slots XOR-swizzled with the pixel index (conflict-free ds_read_b128 for
//     16 consecutive pixels, as in the LDS-DMA kernels above);
//   * weights stream through the same LDS-DMA ring as conv_ring_dma_kernel (first half of the waves), continuously
//     across convs and samples; 64 KiB activation + 2 x 48 KiB ring = the whole 160 KiB;
//   * the next sample's input is fetched by LDS-DMA (second half of the waves) into each 64-channel region of the buffer
//     as soon as the LAST conv has finished reading it: region c during chunk c+1, the last region during the final
//     epilogue and the next sample's first chunk.
// Bit-identical to the three stand-alone launches (same k order, same rounding points: fp16 activations between convs).
// ---------------------------------------------------------------------------------------------
//   * C = 256 (8 x 8 maps, two samples per workgroup): the couts are computed in NPASS = 2 passes of 128 (the stand-alone
//     layer's weight packing); a pass cannot overwrite the buffer while the other still reads it, so pass 0 holds its
//     activated tile in registers (16 VGPRs) until pass 1 has finished.
//   * S2 (whole stage in one launch): the stage's first conv -- 3x3 stride 2 (+ bn1 + ReLU) and its 1x1 stride-2 projection
//     shortcut (+ BN), arch:44-55 -- runs in front of the chain ON THE SAME TILE: its input (C/2 channels at 2H x 2H) is staged
//     16 channels at a time as a zero-padded, column-parity-split patch ((2H+1)^2 pixels x 32 B, slot-swizzled by bit 3 of the
//     pixel index) into the then idle activation buffer, double-buffered; 10 weight "taps" (9 + shortcut) of a chunk are one
//     40 KiB ring step.  t goes straight into the activation buffer, sc into the residual registers: neither touches HBM.
//     LDS during this phase: patches [0, 2 x PBYTES), ring steps at ACT + 8 KiB and ACT + WCHUNK + 8 KiB -- chosen so that
//     the first chain step can be prefetched during the last stride-2 step and vice versa (see the hazards at issue_ring).
//   * KEEP = false (64 channels @ 32 x 32: a whole sample is 128 KiB, weights stream 1 tap = 8 KiB per step through a 4-deep ring,
//     a wave owns 64 couts x 128 pixels = 8 accumulators = 128 VGPRs): no room to keep b0 in registers -- conv 0 also writes it
//     to HBM (ChainConv.y) and the last conv reads it back (L2-hot) as an ordinary residual, both loaded AFTER the last step's
//     MFMA loop.  One channel chunk only, so the next sample's input can be fetched only after the last step of the last conv.
template <int C, int HL, int SPW_L, int WCB, int WPB, int WAVES_C, int WAVES_P, int GT, int RB, int FD, int MINW, int NCONV, bool SPLIT_ROLES, bool S2 = false, bool KEEP = true, bool OOBZ = false>
__global__ __launch_bounds__(64 * WAVES_C * WAVES_P, MINW) void chain_kernel(const ChainArgs a) {
  constexpr int KC = 64, NCHUNK = C / KC, KS = KC / 16, SLOTS = KC / 8, TAPS = 9, NG = (TAPS + GT - 1) / GT, GT_LAST = TAPS - (NG - 1) * GT;  // a last, shorter tap group when GT does not divide 9
  constexpr int CBT = WCB * WAVES_C, CT = 32 * CBT, NPASS = C / CT, NW = WAVES_C * WAVES_P;
  constexpr int H = 1 << HL, HW = H * H, M = HW << SPW_L;
  static_assert(C % CT == 0 && NPASS <= 2 && M == 32 * WAVES_P * WPB && HW >= 32 && C % KC == 0, "tiling");
  static_assert(GT_LAST == GT || RB == 2, "a shorter last tap group issues fewer ring pieces: only with the uncounted vmcnt(0) wait of a 2-deep ring");
  constexpr int WCHUNK = GT * KS * CBT * 1024, NPIECE = WCHUNK / 1024, NBUF = RB, PFD = RB - 1;
  // SPLIT_ROLES: first half of the waves issues the ring DMA (and waits for it every step), second half the activation DMA;
  // otherwise every wave issues its share of both and waits for everything it issued at the end of a step
  constexpr int NWR = SPLIT_ROLES ? NW / 2 : NW, NWP = SPLIT_ROLES ? NW - NWR : NW, WP0 = SPLIT_ROLES ? NWR : 0;
  constexpr int PPWR = (NPIECE + NWR - 1) / NWR;
  constexpr int REGION = M * KC * 2, ACT = NCHUNK * REGION, RPIECE = REGION / 1024, PPR = (RPIECE + NWP - 1) / NWP;
  static_assert(ACT + NBUF * WCHUNK <= 160 * 1024, "LDS");
  static_assert(RB >= 2 && RB <= 4, "ring depth");
  static_assert(SPLIT_ROLES || RB == 2, "unified issue waits with vmcnt(0)");
  static_assert(!S2 || KEEP, "the whole-stage variant keeps sc / b0 in registers");
  static_assert(FD >= 1 && FD <= 3 && FD * (WCB + WPB) <= 15, "fragment prefetch distance");
  static_assert(((GT * KS - 1) * CBT + WCB - 1) * 1024 < 65536, "fragment offsets are ds_read immediates");
  // stride-2 front conv (S2)
  constexpr int CIN = C / 2, HIN = 2 * H, KCS = 16, NCS = CIN / KCS, TAPS_S = 10, NS = S2 ? NPASS * NCS : 0;
  constexpr int NCH = NCONV * NPASS * NCHUNK * NG;                       // chain steps per tile
  constexpr int WSTEP_S = TAPS_S * CBT * 1024, NPIECE_S = WSTEP_S / 1024;
  constexpr int PRS = 2 * H + 1, PCW = 2 * H + 1, PPS = PRS * PCW, PPIX = PPS << SPW_L;  // patch rows, columns (H+1 odd + H even), pixels
  constexpr int PBYTES = (PPIX * KCS * 2 + 1023) / 1024 * 1024, PPIECE = PBYTES / 1024;
  constexpr int SOFF0 = ACT + 8 * 1024, SOFF1 = ACT + WCHUNK + 8 * 1024;  // LDS offsets of the two stride-2 weight steps
  // patch buffers: three where they fit below the first weight step (8 x 8 maps: 19 KiB patches) -> a patch is requested two steps
  // before it is read; else two.  Either way the LAST step of a tile reads buffer 0, the only one the first chain step's weights
  // (landing in [ACT, ACT + WCHUNK) meanwhile) cannot overlap.
  constexpr int NPB = 3 * PBYTES <= SOFF0 ? 3 : 2, PLA = NPB - 1;
  auto pbuf = [](int i) { return NPB == 3 ? i % 3 : (i + 1) & 1; };
  static_assert(!S2 || (NPB == 3 ? (NS - 1) % 3 == 0 : (NS - 1) % 2 == 1), "S2: the last stride-2 step must read patch buffer 0");
  static_assert(!S2 || (SPLIT_ROLES && RB == 2 && NS % 2 == 0 && NCH % 2 == 0 && (NCS - 1) % 2 == 1), "S2: step parities");
  static_assert(!S2 || (2 * PBYTES <= SOFF0 && SOFF0 + WSTEP_S <= ACT + WCHUNK && SOFF1 + WSTEP_S <= 160 * 1024), "S2: LDS layout");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *ring = smem + ACT;
  const uint32_t lds0 = (uint32_t)(uintptr_t)LDS_PTR(smem);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc = wave % WAVES_C, wp = wave / WAVES_C;
  const int p = lane & 31, h = lane >> 5;
  const bool ring_wave = wave < NWR, patch_wave = wave >= WP0;
  const int lane16 = lane * 16;

  // lane -> pixels: EXACTLY the map of the stand-alone kernels (lane ranking; on 8-wide maps a 16-lane ds_read_b128 group
  // owns rows a and a + 4 of a sample), so that the fp32 GAP butterfly adds the same pixels in the same order and the
  // logits are bit-identical whichever kernel variant serves a batch.  mj: logical position (GAP slot), pj: pixel
  // (sample, y, x) = its LDS slot.
  const int pr = p < 4 ? p : p < 12 ? p + 12 : p < 16 ? p - 8 : p < 20 ? p + 8 : p < 28 ? p - 12 : p;
  int mj[WPB], pj[WPB];
#pragma unroll
  for (int j = 0; j < WPB; ++j) {
    mj[j] = (wp * WPB + j) * 32 + pr;
    if constexpr (HL == 3) {
      const int q = mj[j] >> 3, k = q & 7;
      pj[j] = ((q & ~7) | ((k & 1) << 2) | (k >> 1)) * 8 + (mj[j] & 7);
    } else pj[j] = mj[j];
  }
  // maps wider than 8: a lane's pixel blocks are 32 pixels apart and position == pixel, so ONE register carries all of them (the
  // opaque copies that keep address terms from being hoisted would otherwise pin 2 * WPB registers: the 64-channel chain is at the cap)
  auto opaque_pixels = [&]() {
    if constexpr (HL != 3 && WPB > 1) {
#pragma unroll
      for (int j = 1; j < WPB; ++j) { pj[j] = pj[0] + 32 * j; mj[j] = pj[j]; }
      mj[0] = pj[0];
    }
  };
  // 16-byte channel slot swizzle of pixel q: two pixels share a 256-byte bank row; on 8-wide maps bit 5 of q (row a vs
  // a + 4) is folded in so that the two rows of a lane group use disjoint columns
  // (16-wide maps: bit 4 of q -- the row parity -- is folded in as well, so that the epilogue's ds_write_b128, whose 8-lane groups
  // cover columns 0-3 of two consecutive rows, is conflict-free too)
#ifndef CFG_CHAIN_SWZ16
#define CFG_CHAIN_SWZ16 1
#endif
  // OOBZ: a tap outside the map reads an LDS address beyond the allocation (DS reads out of range return 0 on gfx950) instead of the
  // lane's own pixel ANDed with a zero mask (8 VALU ops per k-step and wave: knock-out -4..6 % per launch).  Both forms are built for the
  // three kernels of the default path; the context picks per device (mlt_probe_lds_oob at mlt_init).
#ifndef CFG_CHAIN_SWZ32  // 32-wide maps: 0 = no extra bit.  A read window of a dx = +-1 tap crosses a 16-pixel boundary there, and any
#define CFG_CHAIN_SWZ32 0  // term that depends on bit 4 of q then maps two same-parity pixels of the window to one slot (2-way conflict on 6 of 9 taps)
#endif
  auto swz = [](int q) {
    return ((q >> 1) & (SLOTS - 1)) ^ (HL == 3 ? ((q >> 5) & 1) << 2 : HL == 4 ? (CFG_CHAIN_SWZ16 ? ((q >> 4) & 1) << 1 : 0) : (CFG_CHAIN_SWZ32 ? ((q >> 4) & 1) << 1 : 0));
  };

  const int ntiles = (a.n + (1 << SPW_L) - 1) >> SPW_L;
  int t = blockIdx.x;
  const int tstep = gridDim.x;
  if (t >= ntiles) return;

  // ---- weight ring: one global step sequence (sample, [stride-2: cout pass, chunk], conv, cout pass, chunk, tap group) ----
  // S2 hazards (step i+1 is DMA'd while step i computes; steps of both kinds alternate slots from 0 in every tile):
  //   last chain step (reads chain slot 1 = [ACT+WCHUNK, ..)) -> next tile's first stride-2 step lands in SOFF0 (inside chain slot 0: free);
  //   last stride-2 step (odd index: reads SOFF1 and patch buffer 0) -> first chain step lands in chain slot 0 = [ACT, ACT+WCHUNK):
  //   overlaps only patch buffer 1's tail and SOFF0, both last read one step earlier.
  int steps_to_issue = ((ntiles - 1 - t) / tstep + 1) * (NS + NCH);
  int r_pos = 0, r_cv = 0, r_ps = 0, r_ci = 0, r_gi = 0, ahead = 0;  // r_pos: position in the tile's NS + NCH steps
  int slot_wr = 0;  // ring slot of the next chain step (S2: NCH is even and RB == 2, so every tile starts at slot 0 again)
  auto issue_ring = [&]() {
    if (steps_to_issue <= 0) return;
    const bool s_step = S2 && r_pos < NS;
    if (ring_wave) {
      if (s_step) {
      } else {
        const char *wsrc = (const char *)(r_cv == 0 ? a.cv[0].w : r_cv == 1 ? a.cv[1].w : a.cv[NCONV > 2 ? 2 : 1].w);
        const char *src = wsrc + (size_t)((r_ps * NCHUNK + r_ci) * TAPS + r_gi * GT) * (KS * CBT * 1024);
        char *dst = ring + slot_wr * WCHUNK;
        const int npiece = (GT_LAST != GT && r_gi == NG - 1) ? GT_LAST * KS * CBT : NPIECE;  // (never read behind the layer's last tap)
#pragma unroll
        for (int k = 0; k < PPWR; ++k) {
          int pi = wave + k * NWR;
          pi = pi < npiece ? pi : npiece - 1;  // every ring wave issues exactly PPWR instructions per step (counted vmcnt)
          glds16(src + pi * 1024 + lane16, dst + pi * 1024);
        }
      }
    }
    if (s_step) {  // stride-2 weight steps: every wave issues its share (all waves wait with vmcnt(0) at the end of those steps anyway)
      const char *src = (const char *)a.s2_w + (size_t)r_pos * WSTEP_S;
      char *dst = smem + ((r_pos & 1) ? SOFF1 : SOFF0);
#pragma unroll
      for (int k = 0; k < (NPIECE_S + NW - 1) / NW; ++k) {
        const int pi = wave + k * NW;
        if (pi < NPIECE_S) glds16(src + pi * 1024 + lane16, dst + pi * 1024);
      }
    }
    --steps_to_issue;
    ++ahead;
    if (!s_step) {
      if (++slot_wr == NBUF) slot_wr = 0;
      if (++r_gi == NG) { r_gi = 0; if (++r_ci == NCHUNK) { r_ci = 0; if (++r_ps == NPASS) { r_ps = 0; if (++r_cv == NCONV) r_cv = 0; } } }
    }
    if (++r_pos == NS + NCH) r_pos = 0;
  };
  // ---- S2: input patch of stride-2 step i (16 channels, chunk i % NCS) -> patch buffer (i + 1) & 1 (patch waves).  Patch pixel
  // qS = (sample * PRS + r) * PCW + k holds input pixel (r - 1, k <= H ? 2k - 1 : 2(k - H - 1)); its two 16-byte channel slots are
  // swapped when bit 3 of qS is set (16 consecutive patch pixels span two 256-byte bank rows).  Outside the picture: zero page.
  // Addressing: everything that depends only on (lane, piece) is packed ONCE into 32 bits per piece -- byte offset inside the
  // (sample, 16-channel chunk) | sample-in-tile << 30 | "outside the picture" << 31 -- and the 64-bit source address is formed at the
  // point of issue from an opaque copy.  Left to itself the compiler precomputes the 64-bit addresses of all pieces at the top of a
  // tile, runs out of registers, and reloads them from scratch BETWEEN the DMA instructions: every reload is followed by
  // s_waitcnt vmcnt(0), i.e. by a full round trip of the piece just requested (seen in the ISA: 280 B of scratch, 3 serialised round
  // trips per stride-2 step and 5 in front of the last epilogue's stores).
  constexpr int PPW = (PPIECE + NW - 1) / NW;
  uint32_t pvo[S2 ? PPW : 1];
  if constexpr (S2) {
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
      const int piece = wave + k * NW;
      const int it = piece * 64 + lane, qS = it >> 1, pos = it & 1;
      const int sl = pos ^ ((qS >> 3) & 1);
      const int sm = qS / PPS, rem = qS - sm * PPS, r = rem / PCW, kk = rem - r * PCW;
      const int ri = r - 1, ci = kk <= H ? 2 * kk - 1 : 2 * (kk - H - 1);
      const bool live = qS < PPIX && ri >= 0 && ri < HIN && ci >= 0 && ci < HIN;
      const uint32_t off = a.x_c16 ? ((ri * HIN + ci) * KCS + sl * 8) * 2   // [n][CIN/16][HIN*HIN][16]
                                   : ((ri * HIN + ci) * CIN + sl * 8) * 2;  // NHWC
      pvo[k] = live ? off | ((uint32_t)sm << 30) : 0x80000000u;
    }
  }
  const uint32_t chunk_stride = a.x_c16 ? HIN * HIN * KCS * 2 : KCS * 2;
  auto patch_dma = [&](int tile, int i) {  // every wave issues its share
    const int chunk = i % NCS;
    char *dst = smem + pbuf(i) * PBYTES;
    const char *xb = (const char *)a.x + (size_t)chunk * chunk_stride;
    uint32_t zo = ((int)blockIdx.x * 64 * NW + tid) * 16;
    asm volatile("" : "+v"(zo));
    const char *zsrc = (const char *)a.zero + (zo & 0xFFF0);
#pragma unroll
    for (int k = 0; k < PPW; ++k) {
      const int piece = wave + k * NW;
      if (piece >= PPIECE) break;  // wave-uniform
      uint32_t v = pvo[k];
      asm volatile("" : "+v"(v));
      int n = (tile << SPW_L) + (int)((v >> 30) & 1);
      n = n < a.n ? n : a.n - 1;
      const char *src = (int)v < 0 ? zsrc : xb + (size_t)n * (size_t)(CIN * HIN * HIN * 2) + (v & 0x3FFFFFFFu);
      glds16(src, dst + piece * 1024);
    }
  };
  // ---- activation regions (patch waves): item it = (pixel q, position pos) at byte it * 16 of the region holds channel slot
  // pos ^ ((q >> 1) & 7) of that pixel ----
  auto dma_region = [&](int tile, int c) {
    if (!patch_wave) return;
    // per-piece source offsets are formed here (a few bit operations) from a rematerialised lane id, not hoisted and kept live
    int ln = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int k = 0; k < PPR; ++k) {
      const int piece = (wave - WP0) + k * NWP;
      if (piece >= RPIECE) break;  // wave-uniform
      const int it = piece * 64 + ln, q = it / SLOTS, pos = it & (SLOTS - 1);
      const int sl = pos ^ swz(q);
      int n = (tile << SPW_L) + (q >> (2 * HL));
      n = n < a.n ? n : a.n - 1;  // ragged last tile: a valid sample again (its outputs are masked)
      // (NHWC only.  A chunk-major t for the 64-channel chain was tried: its writer's stores gain what this DMA -- four slots of a pixel per
      // lane quad = two cache lines instead of one -- loses, and the second address form cost this loop a scratch reload per piece)
      const char *src = (const char *)a.x + (((size_t)n * HW + (q & (HW - 1))) * C + c * KC + sl * 8) * 2;
      glds16(src, smem + c * REGION + piece * 1024);
    }
  };

  // ---- prologue ----
  if constexpr (S2) {
#pragma unroll
    for (int i = 0; i < NPB; ++i) patch_dma(t, i);
  } else {
#pragma unroll
    for (int c = 0; c < NCHUNK; ++c) dma_region(t, c);
  }
#pragma unroll
  for (int d = 0; d < PFD; ++d) issue_ring();
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  int slot_rd = 0;
  PHC_DECL;
  // Biases live in registers for the life of the (persistent) workgroup: lane l holds the bias of cout (wc * WCB) * 32 + l of every
  // conv and cout pass (5 VGPRs for the 128 stage, 10 for the 256 stage); an epilogue fetches the 16 values of its accumulator rows
  // with ds_bpermute (crossbar only, no LDS memory).  A bias LOADED in an epilogue costs an exposed L2 round trip four times per
  // tile, and the compiler's s_waitcnt vmcnt(0) in front of its first use also waits for whatever LDS-DMA was issued before it --
  // in the last epilogue that is the next tile's complete input patch.
  constexpr bool PBIAS = WCB == 2;
  float pb_cv[NCONV][NPASS], pb_t[NPASS], pb_sc[NPASS];
  if constexpr (PBIAS) {
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
      for (int c = 0; c < NCONV; ++c) pb_cv[c][ps] = a.cv[c].bias[ps * CT + (wc * WCB) * 32 + lane];
      if constexpr (S2) {
        pb_t[ps] = a.s2_bias[ps * CT + (wc * WCB) * 32 + lane];
        pb_sc[ps] = a.s2_bias_sc[ps * CT + (wc * WCB) * 32 + lane];
      }
    }
  }
  const int pb_lane = 16 * h;  // byte address of lane 4 * h for ds_bpermute
  auto bias_rows = [&](float v, int i, float4v(&out)[4]) {  // out[q][e] = bias of accumulator row 8 * q + 4 * h + e of cout block i
    int base = pb_lane;
    asm volatile("" : "+v"(v), "+v"(base));  // (or the 16 results / lane addresses, tile-invariant, are hoisted out of the tile loop and stay live)
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 4; ++e)
        out[q][e] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(base + (i * 32 + 8 * q + e) * 4, __builtin_bit_cast(int, v)));
  };
  uint4v keep[NPASS][WCB][WPB][2];  // residual tile saved by an earlier conv of the chain (packed fp16, pair16 layout)
  uint4v hold0[WCB][WPB][2];        // S2, two cout passes: pass 0's t tile while pass 1 still reads the input patches
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps)
#pragma unroll
    for (int i = 0; i < WCB; ++i)
#pragma unroll
      for (int j = 0; j < WPB; ++j)
#pragma unroll
        for (int qq = 0; qq < 2; ++qq) keep[ps][i][j][qq] = uint4v{0u, 0u, 0u, 0u};

  for (; t < ntiles; t += tstep) {
    // keep the tile loop from turning every tile-invariant address term into a live register (see conv_mfma_kernel)
#pragma unroll
    for (int j = 0; j < WPB; ++j) asm volatile("" : "+v"(pj[j]));
    opaque_pixels();
    const bool has_next = t + tstep < ntiles;
    int opix[WPB], gidx[WPB];  // flattened (n, y, x) of this lane's output pixels, GAP partial-sum rows
    auto out_index = [&]() {
#pragma unroll
      for (int j = 0; j < WPB; ++j) {
        const int nn = (t << SPW_L) + (pj[j] >> (2 * HL));
        const bool ok = nn < a.n;
        opix[j] = ok ? nn * HW + (pj[j] & (HW - 1)) : -1;
        gidx[j] = ok ? nn * a.gap_slots + ((mj[j] & (HW - 1)) >> 5) : -1;
      }
    };
    // whole-stage variant: only the last epilogue needs them -- computed there from opaque copies, or the store addresses derived
    // from them are formed up here and live (= spilled) across the whole tile
    if constexpr (!S2 && KEEP) out_index();
    auto out_index_late = [&]() {
#pragma unroll
      for (int j = 0; j < WPB; ++j) asm volatile("" : "+v"(pj[j]), "+v"(mj[j]));
      opaque_pixels();
      out_index();
    };

    if constexpr (!S2 && NCHUNK == 1) {  // single-region buffer: the next sample's input was requested only after the previous tile's last step
      if (patch_wave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      PHC_MARK(1);
      asm volatile("s_barrier" ::: "memory");
      PHC_MARK(0);
    }
    if constexpr (S2) {
      // ================= stride-2 conv + projection shortcut of the stage, on this tile =================
      // patches 0 and 1 and the first weight step of this tile (issued a tile ago / in the prologue; every wave has a share)
      asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      PHC_MARK(0);
      int bS[WPB];  // patch pixel of tap (0, 0) for this lane's output pixels
#pragma unroll
      for (int j = 0; j < WPB; ++j) bS[j] = ((pj[j] >> (2 * HL)) * PRS + 2 * ((pj[j] >> HL) & (H - 1))) * PCW + (pj[j] & (H - 1));
      auto put_t = [&](int pass, int i, int j, int qq, uint4v w) {  // this lane: channels cb .. cb+7 of pixel pj[j] -> activation buffer
        const int cb = pass * CT + (wc * WCB + i) * 32 + 16 * qq + 8 * h;
        *(uint4v *)(smem + (cb / KC) * REGION + pj[j] * (KC * 2) + ((((cb % KC) / 8) ^ swz(pj[j])) << 4)) = w;
      };
      static_for<NPASS>([&](auto kp) {
        constexpr int ps = decltype(kp)::value;
        float16v acc[WCB][WPB], accs[WCB][WPB];
#pragma unroll
        for (int i = 0; i < WCB; ++i)
#pragma unroll
          for (int j = 0; j < WPB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; accs[i][j][r] = 0.f; }
#pragma unroll 1
        for (int chunk = 0; chunk < NCS; ++chunk) {
          const int si = ps * NCS + chunk;  // stride-2 step index in this tile
          if (si >= 1 && si + PLA < NS) patch_dma(t, si + PLA);  // (HBM: first) into the buffer step si - 1 has finished with
          issue_ring();
          {
            const uint32_t wb = lds0 + ((si & 1) ? SOFF1 : SOFF0) + (wc * WCB) * 1024 + lane16;
            const uint32_t pb = lds0 + pbuf(si) * PBYTES;
            half8 fa[2][WCB], fb[2][WPB];
            auto issue = [&](auto ic) {
              constexpr int tt = decltype(ic)::value, sl = tt & 1;
              constexpr int te = tt == 9 ? 4 : tt, dy = te / 3, dx = te % 3;  // the 1x1 stride-2 shortcut reads the centre tap's pixel
              constexpr int off = dy * PCW + (dx == 1 ? H + 1 : dx / 2);
              static_for<WCB>([&](auto ii) {
                constexpr int i = decltype(ii)::value;
                lds_read128<(tt * CBT + i) * 1024>(fa[sl][i], wb);
              });
              static_for<WPB>([&](auto jj) {
                constexpr int j = decltype(jj)::value;
                const int qS = bS[j] + off;
                lds_read128<0>(fb[sl][j], pb + qS * (KCS * 2) + ((h ^ ((qS >> 3) & 1)) << 4));
              });
            };
            issue(std::integral_constant<int, 0>{});
            static_for<TAPS_S>([&](auto ic) {
              constexpr int tt = decltype(ic)::value, sl = tt & 1;
              if constexpr (tt + 1 < TAPS_S) issue(std::integral_constant<int, tt + 1>{});
              lds_wait<(tt + 1 < TAPS_S ? WCB + WPB : 0)>();
#pragma unroll
              for (int i = 0; i < WCB; ++i) lds_touch(fa[sl][i]);
#pragma unroll
              for (int j = 0; j < WPB; ++j) lds_touch(fb[sl][j]);
#pragma unroll
              for (int i = 0; i < WCB; ++i)
#pragma unroll
                for (int j = 0; j < WPB; ++j) {
                  if constexpr (tt == 9) accs[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[sl][i], fb[sl][j], accs[i][j], 0, 0, 0);
                  else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[sl][i], fb[sl][j], acc[i][j], 0, 0, 0);
                }
            });
          }
          PHC_MARK(1);
          asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");  // next weight step (ring waves) and next patch (patch waves) landed
          PHC_MARK(2);
          --ahead;
        }
        // t = relu(bn1(conv)) -> activation buffer (pass 0 of 2: held until pass 1 has read its patches -- they overlap the buffer);
        // sc = bn(shortcut) -> residual registers of conv 0
#pragma unroll
        for (int i = 0; i < WCB; ++i) {
          float4v b1[4], bs[4];
          if constexpr (PBIAS) {
            bias_rows(pb_t[ps], i, b1);
            bias_rows(pb_sc[ps], i, bs);
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              b1[q] = *(const float4v *)(a.s2_bias + ps * CT + (wc * WCB + i) * 32 + 4 * h + 8 * q);
              bs[q] = *(const float4v *)(a.s2_bias_sc + ps * CT + (wc * WCB + i) * 32 + 4 * h + 8 * q);
            }
          }
#pragma unroll
          for (int j = 0; j < WPB; ++j)
            static_for<2>([&](auto qc) {
              constexpr int qq = decltype(qc)::value;
              half4 ht[2], hc[2];
              const half4 none{};
              ht[0] = act_quad<2 * qq>(acc[i][j], a.s2_scale, b1[2 * qq], false, none, true);
              ht[1] = act_quad<2 * qq + 1>(acc[i][j], a.s2_scale, b1[2 * qq + 1], false, none, true);
              hc[0] = act_quad<2 * qq>(accs[i][j], a.s2_scale, bs[2 * qq], false, none, false);
              hc[1] = act_quad<2 * qq + 1>(accs[i][j], a.s2_scale, bs[2 * qq + 1], false, none, false);
              keep[ps][i][j][qq] = pair16(hc[0], hc[1]);
              const uint4v wt = pair16(ht[0], ht[1]);
              if constexpr (NPASS == 2 && ps == 0) hold0[i][j][qq] = wt;  // pass 1 still reads the patches, which overlap the buffer
              else put_t(ps, i, j, qq, wt);
            });
        }
        if constexpr (NPASS == 2 && ps == 1) {
#pragma unroll
          for (int i = 0; i < WCB; ++i)
#pragma unroll
            for (int j = 0; j < WPB; ++j)
#pragma unroll
              for (int qq = 0; qq < 2; ++qq) put_t(0, i, j, qq, hold0[i][j][qq]);
        }
      });
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      PHC_MARK(3);
    }

    static_for<NCONV>([&](auto kc) {
      constexpr int cvi = decltype(kc)::value;
      constexpr bool lastc = cvi == NCONV - 1;
      const ChainConv &cv = a.cv[cvi];
      // the two chains there are (host: run_chain): NCONV == 3: conv2(+sc from HBM, saved) -> conv1 -> conv2(+saved tile);
      // NCONV == 2: conv1 -> conv2(+x from HBM); every conv is followed by a ReLU
      // (S2: conv 0's residual sc is not in HBM but in the registers the stride-2 phase left it in)
      // (KEEP = false: b0 goes through HBM instead of the registers: conv 0 stores it, the last conv loads it as cv.res)
      constexpr int RES = NCONV == 3 ? (cvi == 0 ? (S2 ? 2 : 1) : cvi == 2 ? (KEEP ? 2 : 1) : 0) : (cvi == 1 ? 1 : 0);
      constexpr bool SAVE = KEEP && NCONV == 3 && cvi == 0;
      constexpr bool RES_LATE = !KEEP;  // residual loads after the last step's MFMA loop (the fragment registers are free then)
      uint4v hold[WCB][WPB][2];  // NPASS == 2: pass 0's activated tile until pass 1 has finished reading the buffer
      static_for<NPASS>([&](auto kp) {
        constexpr int ps = decltype(kp)::value;
        constexpr bool last = lastc && ps == NPASS - 1;  // the very last K loop over the buffer for this sample
        float16v acc[WCB][WPB];
#pragma unroll
        for (int i = 0; i < WCB; ++i)
#pragma unroll
          for (int j = 0; j < WPB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        uint4v resv[WCB][WPB][2];

#pragma unroll 1
        for (int chunk = 0; chunk < NCHUNK; ++chunk) {
          const uint32_t pl = lds0 + chunk * REGION;
          auto do_step = [&](int g, auto nitem_c) {
            issue_ring();  // step PFD ahead of the one computed now (into the slot read in step g-1)
            // the last K loop has finished reading region chunk-1: the next sample's input may land there
            if (!S2 && last && has_next && g == 0 && chunk >= 1) dma_region(t + tstep, chunk - 1);
            if (RES == 1 && !RES_LATE && chunk == NCHUNK - 1 && g == NG - 1) {  // HBM residual: flies under the last weight step
#pragma unroll
              for (int i = 0; i < WCB; ++i)
#pragma unroll
                for (int j = 0; j < WPB; ++j) {
                  const size_t o = (size_t)(opix[j] >= 0 ? opix[j] : 0) * C + ps * CT + (wc * WCB + i) * 32 + 8 * h;
#pragma unroll
                  for (int qq = 0; qq < 2; ++qq) resv[i][j][qq] = *(const uint4v *)((const _Float16 *)cv.res + o + 16 * qq);
                }
            }
            {
              constexpr int NITEM = decltype(nitem_c)::value, NR = WCB + WPB;
              const uint32_t wb = lds0 + ACT + slot_rd * WCHUNK + (wc * WCB) * 1024 + lane16;
              half8 fa[FD + 1][WCB], fb[FD + 1][WPB];
              uint32_t rowa[WPB], hs[WPB], mcur[WPB], mvs[FD + 1][WPB];
              auto issue = [&](auto ic) {
                constexpr int item = decltype(ic)::value, sl = item % (FD + 1), tt = item / KS, ks = item % KS;
                if constexpr (ks == 0) {
                  const int tp = g * GT + tt, dy = tp / 3, dx = tp - dy * 3;
#pragma unroll
                  for (int j = 0; j < WPB; ++j) {
                    const int yy = ((pj[j] >> HL) & (H - 1)) + dy - 1, xx = (pj[j] & (H - 1)) + dx - 1;
                    const bool ok = (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)H;  // else: conv padding
                    const int q = ok ? pj[j] + (dy - 1) * H + (dx - 1) : pj[j];
                    rowa[j] = pl + q * (KC * 2);
                    if constexpr (OOBZ) {
                      if (!ok) rowa[j] = 0x100000u;  // beyond the LDS: the read returns zeros, no mask needed
                    }
                    hs[j] = (h * 16) ^ (swz(q) << 4);
                    mcur[j] = ok ? 0xFFFFFFFFu : 0u;
                  }
                }
#pragma unroll
                for (int j = 0; j < WPB; ++j) mvs[sl][j] = mcur[j];
                static_for<WCB>([&](auto ii) {
                  constexpr int i = decltype(ii)::value;
                  lds_read128<((tt * KS + ks) * CBT + i) * 1024>(fa[sl][i], wb);
                });
                static_for<WPB>([&](auto jj) { lds_read128<0>(fb[sl][decltype(jj)::value], rowa[decltype(jj)::value] + (hs[decltype(jj)::value] ^ (ks * 32))); });
              };
              issue(std::integral_constant<int, 0>{});
              if constexpr (NITEM > 1 && FD > 1) issue(std::integral_constant<int, 1>{});
              if constexpr (NITEM > 2 && FD > 2) issue(std::integral_constant<int, 2>{});
              static_for<NITEM>([&](auto ic) {
                constexpr int item = decltype(ic)::value, sl = item % (FD + 1);
                if constexpr (item + FD < NITEM) issue(std::integral_constant<int, item + FD>{});
                constexpr int younger = (NITEM - 1 - item < FD ? NITEM - 1 - item : FD) * NR;
                lds_wait<younger>();
#pragma unroll
                for (int i = 0; i < WCB; ++i) lds_touch(fa[sl][i]);
                half8 bm[WPB];
#pragma unroll
                for (int j = 0; j < WPB; ++j) {
                  lds_touch(fb[sl][j]);
                  bm[j] = fb[sl][j];
                  if constexpr (!OOBZ) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ((uint32_t *)&bm[j])[e] &= mvs[sl][j];
                  }
                }
#pragma unroll
                for (int i = 0; i < WCB; ++i)
#pragma unroll
                  for (int j = 0; j < WPB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[sl][i], bm[j], acc[i][j], 0, 0, 0);
              });
            }
            if constexpr (RES == 1 && RES_LATE) {
              // (defined in every iteration by the empty asm: not live across any step's MFMA loop, only from here to the epilogue)
#pragma unroll
              for (int i = 0; i < WCB; ++i)
#pragma unroll
                for (int j = 0; j < WPB; ++j)
#pragma unroll
                  for (int qq = 0; qq < 2; ++qq) asm volatile("" : "=v"(resv[i][j][qq]));
              if (chunk == NCHUNK - 1 && g == NG - 1) {
                out_index_late();
                const _Float16 *resb = (const _Float16 *)KARG(const void *, cv[cvi].res);
                const bool res0_c16 = cvi == 0 && KARG(int, res0_c16) != 0;
#pragma unroll
                for (int i = 0; i < WCB; ++i)
#pragma unroll
                  for (int j = 0; j < WPB; ++j) {
                    // cvi == 0: sc, written NHWC by the stride-2 kernel; else b0, written by this wave below in its own order
                    // (lane-linear 1 KiB blocks: a quad of lanes touches ONE 64-byte span instead of four cache lines)
                    const int op0 = opix[j] >= 0 ? opix[j] : 0;
                    const size_t o = cvi != 0       ? (((size_t)t * NW + wave) * (WCB * WPB * 2) + (i * WPB + j) * 2) * 512 + lane * 8
                                     : res0_c16 ? ((((size_t)(op0 >> (2 * HL)) * (C / 16) + ((ps * CT + (wc * WCB + i) * 32) >> 4)) << (2 * HL)) + (op0 & (HW - 1))) * 16 + 8 * h
                                                : (size_t)op0 * C + ps * CT + (wc * WCB + i) * 32 + 8 * h;
                    // (asm loads: the compiler does not track them, so the epilogue's first use is not preceded by its
                    // s_waitcnt vmcnt(0) -- which would also drain the next sample's input DMA issued in between; the wait is below)
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq) {
                      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(resv[i][j][qq]) : "v"(resb + o + (cvi != 0 ? 512 : res0_c16 ? HW * 16 : 16) * qq) : "memory");
                    }
                  }
              }
            }
            // ---- end of step: next ring step landed (ring waves); next sample's regions landed (patch waves) where the
            // following step reads them ----
            PHC_MARK(4 + 3 * cvi);
            if constexpr (SPLIT_ROLES) {
              if (ring_wave) {  // the step after this one has landed; the ahead - 2 younger ones may still fly
                if (PFD >= 3 && ahead >= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPWR) : "memory");
                else if (PFD >= 2 && ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPWR) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              } else {
                const bool before_next_sample = NCHUNK > 1 && last && chunk == NCHUNK - 1 && g == NG - 1;   // regions 0 .. NCHUNK-2
                const bool before_last_region = cvi == 0 && ps == 0 && chunk == NCHUNK - 2 && g == NG - 1;  // region NCHUNK-1
                if (before_next_sample || before_last_region) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
              }
            } else {
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if constexpr (RES == 1 && RES_LATE) {  // the residual tile requested above (every wave)
              if (chunk == NCHUNK - 1 && g == NG - 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            asm volatile("s_barrier" ::: "memory");
            PHC_MARK(5 + 3 * cvi);
            --ahead;
            if (++slot_rd == NBUF) slot_rd = 0;
          };
          // (a shorter last tap group is peeled: both item counts inside one loop body, selected by a branch, cost 900 B of scratch)
#pragma unroll 1
          for (int g = 0; g < (GT_LAST != GT ? NG - 1 : NG); ++g) do_step(g, std::integral_constant<int, GT * KS>{});
          if constexpr (GT_LAST != GT) do_step(NG - 1, std::integral_constant<int, GT_LAST * KS>{});
        }
        // every wave is past the last step's barrier
        if constexpr (lastc) {
          if constexpr (last) {  // nobody reads the activation buffer any more
            if constexpr (S2) {
              if (has_next) {  // all patch buffers of the next tile
#pragma unroll
                for (int i = 0; i < NPB; ++i) patch_dma(t + tstep, i);
              }
            } else if (has_next) dma_region(t + tstep, NCHUNK - 1);
          }
          if constexpr (S2 || !KEEP) out_index_late();
          auto std_epi = [&]() {
          float4v bq[WCB][4], bsq[1][4];
#pragma unroll
          for (int i = 0; i < WCB; ++i) {
            if constexpr (PBIAS) bias_rows(pb_cv[cvi][ps], i, bq[i]);
            else {
#pragma unroll
              for (int q = 0; q < 4; ++q) bq[i][q] = *(const float4v *)(cv.bias + ps * CT + (wc * WCB + i) * 32 + 4 * h + 8 * q);
            }
          }
          ConvArgs ea{};
          ea.y = a.y; ea.gap = a.gap; ea.gap_slots = a.gap_slots; ea.gap_l = a.gap_l; ea.acc_scale = cv.acc_scale; ea.relu = 1;
          ea.y_c16 = a.y_c16; ea.hout_l = HL;
          ea.res = RES ? (const void *)a.x : nullptr;  // non-NULL = "add resv"
          float16v acc_sc[1][1];
          uint4v resl[1][1][2];
          if constexpr (RES == 2) {
#pragma unroll
            for (int i = 0; i < WCB; ++i)
#pragma unroll
              for (int j = 0; j < WPB; ++j)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) resv[i][j][qq] = keep[ps][i][j][qq];
          }
          int he = h, pe = p;
          asm volatile("" : "+v"(he), "+v"(pe));  // nothing of the epilogue's lane arithmetic is hoisted out of the tile loop
          conv_epilogue<C, CT, WCB, WPB, false, 1>(ea, ps, wc, he, pe, opix, gidx, acc, acc_sc, bq, bsq, resv, resl);
          };
          // !KEEP (8 accumulators + the residual tile in registers): the same epilogue one 32-cout block at a time, so that only one
          // block's 16 bias values are live (both blocks' at once, as above, spilled)
          auto blk_epi = [&]() {
            ConvArgs ea{};
            ea.y = KARG(void *, y); ea.gap = KARG(float *, gap); ea.gap_slots = KARG(int, gap_slots); ea.gap_l = KARG(int, gap_l);
            ea.acc_scale = KARG(float, cv[cvi].acc_scale); ea.relu = 1;
            ea.y_c16 = KARG(int, y_c16); ea.hout_l = HL;
            ea.res = RES ? (const void *)ea.y : nullptr;  // (only its non-NULLness is looked at)
            float16v acc_sc[1][1];
            uint4v resl[1][1][2];
            float4v bsq[1][4];
#pragma unroll
            for (int i = 0; i < WCB; ++i) {
              float4v bq1[1][4];
              bias_rows(pb_cv[cvi][ps], i, bq1[0]);
              if (i) out_index_late();  // (recomputed per block: two 64-bit GAP row offsets kept across block 0 were spilled)
              int he = h, pe = p;
              asm volatile("" : "+v"(he), "+v"(pe));
              conv_epilogue<C, CT, 1, WPB, false, 1>(ea, ps, wc * WCB + i, he, pe, opix, gidx, reinterpret_cast<float16v(&)[1][WPB]>(acc[i]), acc_sc, bq1, bsq,
                                                     reinterpret_cast<const uint4v(&)[1][WPB][2]>(resv[i]), resl);
            }
          };
          if constexpr (!KEEP && PBIAS) blk_epi();
          else std_epi();
          PHC_MARK(6 + 3 * cvi);
        } else {
          // + bias (+ residual) (ReLU) -> fp16 -> back into the activation buffer (input of the next conv)
          _Float16 *yb0 = nullptr;  // !KEEP: HBM copy of this conv's output (the later residual), private layout (see the loads)
          if constexpr (!KEEP) yb0 = (_Float16 *)KARG(void *, cv[cvi].y);
          // (lane half from a rematerialised lane id: h itself, kept live over the step loops, was spilled in the 64-channel chain and
          // its reload's s_waitcnt vmcnt(0) drained the weight ring prefetch)
          const int hm = KEEP ? h : (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) >> 5;
          auto put = [&](int pass, int i, int j, int qq, uint4v w) {  // this lane: channels cb .. cb+7 of pixel pj[j]
            const int cb = pass * CT + (wc * WCB + i) * 32 + 16 * qq + 8 * hm;
            *(uint4v *)(smem + (cb / KC) * REGION + pj[j] * (KC * 2) + ((((cb % KC) / 8) ^ swz(pj[j])) << 4)) = w;
          };
#pragma unroll
          for (int i = 0; i < WCB; ++i) {
            float4v bi[4];
            if constexpr (PBIAS) bias_rows(pb_cv[cvi][ps], i, bi);
            else {
#pragma unroll
              for (int q = 0; q < 4; ++q) bi[q] = *(const float4v *)(cv.bias + ps * CT + (wc * WCB + i) * 32 + 4 * h + 8 * q);
            }
#pragma unroll
            for (int j = 0; j < WPB; ++j)
              static_for<2>([&](auto qc) {
                constexpr int qq = decltype(qc)::value;
                half4 ra{}, rb{}, hq[2];
                if constexpr (RES != 0) unpair16(RES == 2 ? keep[ps][i][j][qq] : resv[i][j][qq], ra, rb);
                hq[0] = act_quad<2 * qq>(acc[i][j], cv.acc_scale, bi[2 * qq], RES != 0, ra, true);
                hq[1] = act_quad<2 * qq + 1>(acc[i][j], cv.acc_scale, bi[2 * qq + 1], RES != 0, rb, true);
                const uint4v w = pair16(hq[0], hq[1]);
                if constexpr (SAVE) keep[ps][i][j][qq] = w;
                if constexpr (!KEEP) {  // the later residual goes through HBM
                  if (yb0) *(uint4v *)(yb0 + (((size_t)t * NW + wave) * (WCB * WPB * 2) + (i * WPB + j) * 2 + qq) * 512 + lane * 8) = w;
                }
                if constexpr (ps == NPASS - 1) put(ps, i, j, qq, w);
                else hold[i][j][qq] = w;
              });
          }
          if constexpr (ps == NPASS - 1) {
            if constexpr (NPASS == 2) {
#pragma unroll
              for (int i = 0; i < WCB; ++i)
#pragma unroll
                for (int j = 0; j < WPB; ++j)
#pragma unroll
                  for (int qq = 0; qq < 2; ++qq) put(0, i, j, qq, hold[i][j][qq]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
          }
          PHC_MARK(6 + 3 * cvi);
        }
      });
    });
  }
  PHC_FLUSH((C == 256 ? 2 : 0) + (S2 ? 1 : 0));
}

// ---------------------------------------------------------------------------------------------
// Fused identity BasicBlock for the 32-channel stage (layer0.1, arch:52-57 with identity shortcut):
//     out = relu( bn2(conv2( relu(bn1(conv1(x))) )) + x )
// One workgroup owns a 16 x 32 output tile: the input patch (tile + 2-pixel halo) is staged once, conv1 is evaluated on
// tile + 1-pixel halo and its activation goes to LDS as fp16 (zero outside the picture = conv2's padding), conv2 reads it
// from LDS and takes the residual from the input patch that is still in LDS.  Per CU this reads x once (1.4x with halo,
// mostly L2 hits) and writes out once: 616 KiB instead of the 1280 KiB of two separate conv launches; the intermediate
// activation never touches HBM.  Both weight sets (2 x 18 KiB) stay resident in LDS; workgroups are persistent over tiles
// and prefetch the next tile's patch into registers while computing (issue-early / commit-late).  Fast arithmetic only.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void block32_kernel(const Block32Args a) {
  constexpr int TH = 16, TW = 32, C = 32, PS = 80;          // pixel stride: 64 B of channels + 16 B pad (PS/16 odd)
  constexpr int XH = TH + 4, XW = TW + 4, T_H = TH + 2, T_W = TW + 2;
  constexpr int XBYTES = XH * XW * PS, TBYTES = T_H * T_W * PS, WBYTES = 18 * 1024;
  constexpr int NT = 512, NW = 8, UN = 6;                   // 20*36*4 = 2880 patch items <= 6 * 512
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *X = smem, *T = smem + XBYTES, *W1 = T + TBYTES, *W2 = W1 + WBYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 31, h = lane >> 5;
  const int h_l = a.h_l, H = 1 << h_l;
  const int txs_l = h_l - 5, tys_l = h_l - 4;               // tiles per row (H/32) and per column (H/16)
  const int ntiles = a.ntiles;
  auto tile_decode = [&](int t, int &tx, int &ty, int &n) {
    const int q = ntiles >> 3, r = ntiles & 7, xcd = t & 7;  // XCD-contiguous tile order (see conv_mfma_kernel)
    const int mt = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
    tx = mt & ((1 << txs_l) - 1);
    ty = (mt >> txs_l) & ((1 << tys_l) - 1);
    n = mt >> (txs_l + tys_l);
  };
  // resident weights: 18 one-KiB LDS-DMA pieces per conv
  for (int pi = wave; pi < 18; pi += NW) {
    glds16((const char *)a.w1 + pi * 1024 + lane * 16, W1 + pi * 1024);
    glds16((const char *)a.w2 + pi * 1024 + lane * 16, W2 + pi * 1024);
  }
  // ---- input patch: issue (global -> registers) / commit (registers -> LDS) ----
  half8 pv[UN];
  int pdst[UN];  // LDS byte offset, bit 30: zero-fill, -1: none
  auto issue_patch = [&](int t) {
    int tx, ty, n;
    tile_decode(t, tx, ty, n);
    const int iy0 = ty * TH - 2, ix0 = tx * TW - 2;
    const size_t nbase = (((size_t)n << h_l) << h_l) * C;
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int it = tid + u * NT, slot = it & 3, pix = it >> 2;
      const int row = pix / XW, col = pix - row * XW;
      const int iy = iy0 + row, ix = ix0 + col;
      const bool in_items = pix < XH * XW;
      const bool live = in_items && iy >= 0 && iy < H && ix >= 0 && ix < H;
      pdst[u] = in_items ? ((row * XW + col) * PS + slot * 16) | (live ? 0 : 1 << 30) : -1;
      // masked lanes load a valid, lane-distinct address inside this sample (never one shared hot line)
      const size_t off = live ? nbase + ((((size_t)iy << h_l) + ix) * C) + slot * 8 : nbase + (size_t)((tid * 8) & ((C << (2 * h_l)) - 8));
      pv[u] = *(const half8 *)((const _Float16 *)a.x + off);
    }
  };
  auto commit_patch = [&]() {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      half8 v = pv[u];
      if (pdst[u] & (1 << 30)) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (_Float16)0.f;
      }
      if (pdst[u] >= 0) *(half8 *)(X + (pdst[u] & ~(1 << 30))) = v;
    }
  };
  // One 32-pixel block of a 3x3 conv.  The conv's 18 weight (A) fragments sit in REGISTERS for the whole phase (LDS
  // capacity already limits this kernel to one workgroup per CU, so VGPRs are free): only the activation (B) fragments
  // are read from LDS, 1 ds_read_b128 per MFMA instead of 2, software-pipelined one item ahead.
  half8 wf[18];
  auto load_weights = [&](const char *Wres) {
#pragma unroll
    for (int item = 0; item < 18; ++item) wf[item] = *(const half8 *)(Wres + item * 1024 + lane * 16);
  };
  auto conv_block = [&](const char *patch, int base, int pitch) -> float16v {
    float16v acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto frag = [&](int item) -> half8 {
      const int tp = item >> 1, ks = item & 1, dy = tp / 3, dx = tp - dy * 3;
      return *(const half8 *)(patch + base + (dy * pitch + dx) * PS + ks * 32);
    };
    half8 bf[3];  // activation fragments run TWO items ahead of the MFMA that consumes them (LDS latency ~ 2 MFMAs)
    bf[0] = frag(0);
    bf[1] = frag(1);
#pragma unroll
    for (int item = 0; item < 18; ++item) {  // 9 taps x 2 k-steps
      if (item + 2 < 18) bf[(item + 2) % 3] = frag(item + 2);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[item], bf[item % 3], acc, 0, 0, 0);
    }
    return acc;
  };
  // folded BN biases of this lane's 16 output channels, once per kernel (a load inside the per-block epilogue is waited
  // for on the spot: four serialised memory round trips per 32-pixel block)
  float4v b1r[4], b2r[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    b1r[q] = *(const float4v *)(a.bias1 + 4 * h + 8 * q);
    b2r[q] = *(const float4v *)(a.bias2 + 4 * h + 8 * q);
  }

#pragma unroll
  for (int q = 0; q < 4; ++q) { asm volatile("" ::"v"(b1r[q])); asm volatile("" ::"v"(b2r[q])); }  // wait for the biases HERE, not inside the tile loop (see stem_block_kernel)
  int t = blockIdx.x;
  if (t < ntiles) issue_patch(t);
  PH_DECL;
  for (; t < ntiles; t += gridDim.x) {
    int tx, ty, n;
    tile_decode(t, tx, ty, n);
    PH_MARK(7);
    commit_patch();  // waits for exactly the prefetched loads (the resident weights' DMA is older, hence landed too)
    PH_MARK(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    PH_MARK(1);
    const int t_next = t + gridDim.x;
    if (t_next < ntiles) issue_patch(t_next);
    PH_MARK(2);

    // ---- conv1 + bn1 + relu on tile + 1-pixel halo (18 x 34 = 612 pixels = 20 blocks) -> T (fp16, LDS) ----
    load_weights(W1);
    // Blocks 0..17: one T row each, columns 0..31 (32 consecutive patch pixels: conflict-free fragment reads; a block that
    // wrapped over the row end would shift part of its lanes by XW - T_W pixels).  Blocks 18, 19: columns 32, 33 of all rows.
    for (int pb = wave; pb < T_H + 2; pb += NW) {
      const int m = (pb - T_H) * 32 + p;  // blocks 18, 19: index into the 2 x 18 leftover pixels
      const bool ok = pb < T_H || m < 2 * T_H;
      const int y1 = pb < T_H ? pb : (ok ? m >> 1 : 0), x1 = pb < T_H ? p : 32 + (m & 1);
      const float16v acc = conv_block(X, (y1 * XW + x1) * PS + h * 16, XW);
      // positions outside the picture are conv2's zero padding, not conv1 of padded input
      const int gy = ty * TH - 1 + y1, gx = tx * TW - 1 + x1;
      const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < H;
      half4 o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) o[q][e] = inside ? (_Float16)fmaxf(acc[4 * q + e] + b1r[q][e], 0.f) : (_Float16)0.f;
      // 16-byte stores (pair16: lanes 0-31 take channels 16qq..+7, lanes 32-63 the next 8): ds_write_b128 at an 80-byte
      // pixel pitch is conflict-free, the four ds_write_b64 it replaces were 2-way conflicted
      char *dst = T + (y1 * T_W + x1) * PS + 16 * h;
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const uint4v w = pair16(o[2 * qq], o[2 * qq + 1]);  // every lane takes part in the swap
        if (ok) *(uint4v *)(dst + 32 * qq) = w;
      }
    }
    PH_MARK(3);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    PH_MARK(4);

    // ---- conv2 + bn2 + residual (input patch centre) + relu on the 16 x 32 tile (16 blocks) -> HBM ----
    load_weights(W2);
    for (int pb = wave; pb < TH * TW / 32; pb += NW) {
      const int y = pb, x = p;  // one tile row per block (TW = 32)
      const float16v acc = conv_block(T, (y * T_W + x) * PS + h * 16, T_W);
      const char *rsrc = X + ((y + 2) * XW + x + 2) * PS + 16 * h;  // residual = centre of the input patch, 16-byte reads
      half4 hq[4];
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        half4 r[2];
        unpair16(*(const uint4v *)(rsrc + 32 * qq), r[0], r[1]);
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int e = 0; e < 4; ++e) hq[2 * qq + k][e] = (_Float16)fmaxf(acc[8 * qq + 4 * k + e] + b2r[2 * qq + k][e] + (float)r[k][e], 0.f);
      }
      const size_t ob = ((((size_t)n << h_l) + ty * TH + y) << h_l) * C + (size_t)(tx * TW + x) * C + 8 * h;
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) *(uint4v *)((_Float16 *)a.y + ob + 16 * qq) = pair16(hq[2 * qq], hq[2 * qq + 1]);
    }
    // the next commit overwrites X, the next conv1 overwrites T
    PH_MARK(5);
    if (t_next < ntiles) __builtin_amdgcn_s_barrier();
    PH_MARK(6);
  }
  PH_FLUSH(13);
}

// ---------------------------------------------------------------------------------------------
// First layer, composed.  The stem conv has neither BN nor ReLU (arch:277-278: out = conv1(x), then layer0), so
//   t  = relu(bn1(conv3x3_s2(stem(x))))   is ONE linear 5x5 stride-2 conv of the 2 raw channels (+ bias, ReLU), and
//   sc = bn(conv1x1_s2(stem(x)))          is ONE linear 3x3 stride-2 conv of the raw channels.
// The only place the two-step form differs is conv1's ZERO padding of the stem activation (row -1 / column -1 of the
// stem map): the composed conv would see stem(padded input) there.  That affects output row 0 and column 0 only and
// is cancelled exactly by extra K entries whose B operand is the raw input row 0 / column 0 (/ pixel (0,0)), masked to
// those output pixels (weights: mlt_model.cpp pack_stem5; algebra checked to 1e-15 against the two-step form).
// GEMM per 32-pixel block: K = 40 dword slots (25 taps + 5 top + 5 left + 1 corner + 4 pad) x 2 channels = 5 k-steps for
// t, 2 k-steps for sc, B fragments gathered with ds_read_b32 straight from the raw (org, |org-pred|) LDS patch.
// HBM traffic per CU: 64 KiB of Pel planes in, 2 x 256 KiB (S = 128) out; the stem activation never exists.
// ---------------------------------------------------------------------------------------------
template <int NSPLIT>
__global__ __launch_bounds__(256) void stem5_kernel(const Stem5Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint32_t *raw = (uint32_t *)smem;
  constexpr int NT = 256, WPB = 2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 31, h = lane >> 5;
  const int tw_l = a.tw_l, th_l = a.th_l, spw_l = a.spw_l, hout_l = a.hout_l;
  const int TW = 1 << tw_l, TH = 1 << th_l, S = 1 << a.s_l;
  const int txs_l = hout_l - tw_l, tys_l = hout_l - th_l;
  int mtile;
  {
    const int nwg = gridDim.x, bid = blockIdx.x, q = nwg >> 3, r = nwg & 7, xcd = bid & 7;
    mtile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  }
  const int tx = mtile & ((1 << txs_l) - 1);
  const int ty = (mtile >> txs_l) & ((1 << tys_l) - 1);
  const int n0 = (mtile >> (txs_l + tys_l)) << spw_l;
  const int RH = a.rh, RW = a.rw, HW = a.halfw, RP = 2 * HW;  // raw patch rows / cols, parity-split row pitch
  const int iy0 = 2 * (ty << th_l) - 2, ix0 = 2 * (tx << tw_l) - 2;

  // ---- raw (org,resi) patch -> LDS; even columns first, odd columns at +HW (stride-2 taps become unit stride) ----
  {
    const int raw_items = (1 << spw_l) * RH * RW;
    const int step_r = udiv_magic(NT, a.rw_magic), step_x = NT - step_r * RW;
    int rr = udiv_magic(tid, a.rw_magic), rx = tid - rr * RW;
    constexpr int UR = 6;
    for (int it0 = tid; it0 < raw_items; it0 += UR * NT) {
      int16_t vo[UR], vp[UR];
      bool in[UR];
      int dst[UR];
#pragma unroll
      for (int u = 0; u < UR; ++u) {
        const int it = it0 + u * NT;
        const int s = udiv_magic(rr, a.rh_magic), ry = rr - s * RH;
        const int iy = iy0 + ry, ix = ix0 + rx;
        in[u] = it < raw_items && iy >= 0 && iy < S && ix >= 0 && ix < S && (n0 + s) < a.n;
        dst[u] = it < raw_items ? rr * RP + (rx & 1) * HW + (rx >> 1) : -1;
        const size_t oo = in[u] ? (size_t)(n0 + s) * a.org_cu_stride + (size_t)iy * a.org_row_stride + ix : (size_t)n0 * a.org_cu_stride + (tid & (S - 1));
        const size_t po = in[u] ? (size_t)(n0 + s) * a.pred_cu_stride + (size_t)iy * a.pred_row_stride + ix : (size_t)n0 * a.pred_cu_stride + (tid & (S - 1));
        vo[u] = a.org[oo];
        vp[u] = a.pred[po];
        rx += step_x; rr += step_r;
        if (rx >= RW) { rx -= RW; ++rr; }
      }
#pragma unroll
      for (int u = 0; u < UR; ++u)
        if (dst[u] >= 0) raw[dst[u]] = in[u] ? prep_pair(vo[u], vp[u]) : 0u;
    }
  }
  // ---- A fragments (weights) stay in registers: 5 k-steps for t, 2 for sc, per split plane ----
  constexpr int WS = NSPLIT >= 2 ? 2 : 1;  // weight planes (NSPLIT == 3: hi+lo weights, single outputs)
  half8 am[WS][5], as[WS][2];
#pragma unroll
  for (int sp = 0; sp < WS; ++sp) {
    const char *w = (const char *)a.w + sp * a.w_lo_off;
#pragma unroll
    for (int k = 0; k < 5; ++k) am[sp][k] = *(const half8 *)(w + k * 1024 + lane * 16);
#pragma unroll
    for (int k = 0; k < 2; ++k) as[sp][k] = *(const half8 *)(w + (5 + k) * 1024 + lane * 16);
  }
  __syncthreads();

  // biases once per kernel (a load inside the per-block epilogue would be waited for on the spot)
  float4v sb1[4], sbs[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    sb1[q] = *(const float4v *)(a.bias + 4 * h + 8 * q);
    sbs[q] = *(const float4v *)(a.bias_sc + 4 * h + 8 * q);
  }
  // raw offset of tap (u, v) relative to the pixel origin (row 2y, column 2x): columns are parity-split
  auto tap = [&](int u, int v) { return u * RP + (v & 1) * HW + (v >> 1); };
  auto main_off = [&](int slot) { return tap(slot / 5, slot % 5); };
  const int m_valid = 1 << (tw_l + th_l + spw_l);
#pragma unroll
  for (int j = 0; j < WPB; ++j) {
    const int m = (wave * WPB + j) * 32 + p;
    bool ok = m < m_valid;
    const int mm = ok ? m : 0;
    const int x = mm & (TW - 1), y = (mm >> tw_l) & (TH - 1), s = mm >> (tw_l + th_l);
    const int gy = (ty << th_l) + y, gx = (tx << tw_l) + x;
    ok = ok && (n0 + s) < a.n;
    const int o = (s * RH + 2 * y) * RP + x;          // pixel origin = input (2y-2, 2x-2)
    const bool top = gy == 0, left = gx == 0;
    const int orow0 = (s * RH + 2) * RP;              // raw row of input row 0 (tiles with ty == 0)
    // value of K slot `slot` for this pixel (slot is a compile-time constant at every call site)
    auto slot_val = [&](int slot) -> uint32_t {
      if (slot < 25) return raw[o + main_off(slot)];
      if (slot < 30) { const uint32_t v = raw[top ? orow0 + x + tap(0, slot - 25) : o]; return top ? v : 0u; }                 // in(0, 2x-2+v)
      if (slot < 35) { const uint32_t v = raw[left ? o + (slot - 30) * RP + 1 : o]; return left ? v : 0u; }                    // in(2y-2+u, 0)
      if (slot == 35) { const uint32_t v = raw[(top && left) ? orow0 + 1 : o]; return (top && left) ? v : 0u; }                // in(0, 0)
      return 0u;
    };
    float16v acc, accs;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accs[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
      half8 b;
      uint32_t *bw = (uint32_t *)&b;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (ks < 3) bw[e] = raw[o + (h ? main_off(8 * ks + 4 + e) : main_off(8 * ks + e))];  // both halves are plain taps
        else { const uint32_t v0 = slot_val(8 * ks + e), v1 = slot_val(8 * ks + 4 + e); bw[e] = h ? v1 : v0; }
      }
#pragma unroll
      for (int sp = 0; sp < WS; ++sp) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[sp][ks], b, acc, 0, 0, 0);
    }
    // shortcut: 3x3 stride-2 taps (b_y, b_x) at raw offset tap(1 + b_y, 1 + b_x); slots 0..8 of 16
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      half8 b;
      uint32_t *bw = (uint32_t *)&b;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int t0 = 8 * ks + e, t1 = 8 * ks + 4 + e;  // slot for h = 0 / h = 1
        const uint32_t v0 = t0 < 9 ? raw[o + tap(1 + t0 / 3, 1 + t0 % 3)] : 0u;
        const uint32_t v1 = t1 < 9 ? raw[o + tap(1 + t1 / 3, 1 + t1 % 3)] : 0u;
        bw[e] = h ? v1 : v0;
      }
#pragma unroll
      for (int sp = 0; sp < WS; ++sp) accs = __builtin_amdgcn_mfma_f32_32x32x16_f16(as[sp][ks], b, accs, 0, 0, 0);
    }
    {
      const size_t ob = ok ? ((((size_t)(n0 + s) << hout_l) + gy) << hout_l) * 32 + (size_t)gx * 32 + 8 * h : 0;  // 16-byte span of quad pair 0
      half4 t[4], tl[4], c[4], cl[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float vt = fmaxf(acc[4 * q + e] * a.acc_scale + sb1[q][e], 0.f);  // relu(bn1(conv1(stem)))
          const float vc = accs[4 * q + e] * a.acc_scale + sbs[q][e];             // bn(shortcut conv(stem))
          t[q][e] = (_Float16)vt; tl[q][e] = (_Float16)(vt - (float)t[q][e]);
          c[q][e] = (_Float16)vc; cl[q][e] = (_Float16)(vc - (float)c[q][e]);
        }
      }
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {  // all lanes swap, valid pixels store 16 B per lane and quad pair
        const uint4v wt = pair16(t[2 * qq], t[2 * qq + 1]), wc = pair16(c[2 * qq], c[2 * qq + 1]);
        if (ok) {
          *(uint4v *)((_Float16 *)a.y + ob + 16 * qq) = wt;
          *(uint4v *)((_Float16 *)a.y_sc + ob + 16 * qq) = wc;
        }
        if constexpr (NSPLIT == 2) {
          const uint4v wtl = pair16(tl[2 * qq], tl[2 * qq + 1]), wcl = pair16(cl[2 * qq], cl[2 * qq + 1]);
          if (ok) {
            *(uint4v *)((char *)((_Float16 *)a.y + ob + 16 * qq) + a.y_lo_off) = wtl;
            *(uint4v *)((char *)((_Float16 *)a.y_sc + ob + 16 * qq) + a.ysc_lo_off) = wcl;
          }
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Whole first BasicBlock (layer0.0) from the raw Pel planes, fast arithmetic, output maps >= 32 x 32:
//     t  = relu(bn1(conv1(stem x)))         composed 5x5 stride-2 conv + border k-steps   (see stem5_kernel)
//     b0 = relu(bn2(conv2(t)) + bn(shortcut(stem x)))
// One workgroup owns a 16 x 32 tile of b0: the raw (org, |org-pred|) patch for tile + 1-pixel halo of t is staged in LDS
// (39 x 71 pixels, 11 KiB), t is evaluated on 18 x 34 pixels and kept in LDS as fp16 (zero outside the picture = conv2's
// padding), conv2 reads it from LDS, and the shortcut is evaluated for the same pixel block and added in fp32 (it is never
// rounded to fp16).  HBM sees 64 KiB of Pel planes in and 256 KiB of b0 out per 128x128 CU -- t and sc (2 x 256 KiB written
// and read back by the two-kernel form) never exist.  conv2's weights are resident in LDS (and in registers during the
// phase), the composed first-layer weights live in registers; persistent workgroups prefetch the next raw patch.
// Round 3: the raw patch is ROW-MAJOR (one dword = the (org, resi) fp16 pair of a pixel, pitch 72 dwords) and the K order of the
// composed weights (mlt_model.cpp: pack_stem_b) makes a lane's B fragment 16 contiguous bytes of it -- window row dy, columns
// 2x-2+4h .. +3 -- so a 32-pixel block of t costs 5 ds_read2_b64 instead of 20-25 gathered ds_read_b32 (round 2: 22.8 % LDS bank
// conflicts, MFMA busy 29 %); the three surplus columns of a row carry zero weights.  The border corrections (conv1 pads the STEM
// map, the composed conv the INPUT: output row 0 / column 0 only) are two extra k-steps whose B operands are picked out of the
// main fragments already in registers, executed only by blocks that contain such pixels.
// ---------------------------------------------------------------------------------------------
// Round 3, second step: the kernel is a two-stage PIPELINE inside one workgroup.  Phase stamps of the serial form (commit -> barrier
// -> phase 1 -> barrier -> phase 2 -> barrier, every wave doing everything) showed 10.8 k cycles per tile against an MFMA floor of
// 3.4 k: phase 1 -- 5 MFMAs per 32-pixel block between an LDS round trip and a VALU epilogue -- is a latency chain that took as long
// as phase 2 with its 8x more MFMAs, and commit + barriers another 30 %.  Now waves 0-3 run phase 1 of tile k while waves 4-7 run
// phase 2 of tile k-1 (every SIMD hosts one wave of each kind: the latency-bound producer fills the issue slots the MFMA-dense
// consumer leaves), every wave commits its share of the raw patch of tile k+1 at the top of the interval, and ONE barrier per tile
// separates the intervals.  LDS: three raw buffers (tile k+1 being written, k read by phase 1, k-1 read by phase 2's shortcut), two T
// buffers, conv2's weights (the consumer waves keep their 18 A fragments in registers), biases, border k-steps = 149 KiB.
// ---------------------------------------------------------------------------------------------
#ifndef CFG_SB_NWS
#define CFG_SB_NWS 4   // waves per pipeline stage (4: 8 waves per workgroup, 2 per SIMD; 8: 16 waves, 4 per SIMD at <= 128 VGPRs)
#endif
#ifndef CFG_SB_PD
#define CFG_SB_PD 8
#endif
__global__ __launch_bounds__(128 * CFG_SB_NWS) void stem_block_kernel(const StemBlockArgs a) {
  constexpr int TH = 16, TW = 32, PS = 80;
  constexpr int T_H = TH + 2, T_W = TW + 2;                 // t region: tile + 1-pixel halo
  constexpr int RH = 2 * T_H + 3, RW = 2 * T_W + 3;         // raw patch 39 x 71
  constexpr int RP = 72;                                    // row pitch in dwords (pixels): quads of 4 pixels are 16-byte aligned
  constexpr int RAWBYTES = RH * RP * 4 + 16, TBYTES = T_H * T_W * PS;  // (+16: the zero-weight columns of the last row's reads)
  constexpr int NWS = CFG_SB_NWS, NT = 128 * NWS;           // waves per pipeline stage, threads
  constexpr int QW = RP / 4, UR = (RH * QW + NT - 1) / NT;  // raw rows are fetched as 18 quads of 4 pixels: 702 items, 2 per lane
  static_assert(RW <= RP && RP % 4 == 0 && RAWBYTES % 16 == 0, "raw pitch");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char *Tb = smem + 3 * RAWBYTES, *W2 = Tb + 2 * TBYTES;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int p = lane & 31, h = lane >> 5;
  const int h_l = a.hout_l, H = 1 << h_l, S = 2 * H;        // output map H x H, picture S x S
  const int txs_l = h_l - 5, tys_l = h_l - 4;
  const int ntiles = a.ntiles;
  if ((int)blockIdx.x >= ntiles) return;
  const int nloc = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;  // tiles of this workgroup
  auto tile_decode = [&](int k, int &tx, int &ty, int &n) {
    const int t = (int)blockIdx.x + k * (int)gridDim.x;
    const int q = ntiles >> 3, r = ntiles & 7, xcd = t & 7;  // XCD-contiguous tile order
    const int mt = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (t >> 3);
    tx = mt & ((1 << txs_l) - 1);
    ty = (mt >> txs_l) & ((1 << tys_l) - 1);
    n = mt >> (txs_l + tys_l);
  };
  auto rawbuf = [&](int k) { return (uint32_t *)(smem + (k % 3) * RAWBYTES); };
  for (int pi = wave; pi < 18; pi += 2 * NWS) glds16((const char *)a.w2 + pi * 1024 + lane * 16, W2 + pi * 1024);
  // biases in LDS: BL[0..31] = bias of conv1 (bn1), BL[32..63] = bias of conv2 (bn2) + bias of the shortcut (only ever added together)
  float *BL = (float *)(W2 + 18 * 1024);
  char *WB = (char *)(BL + 64);  // the two border k-steps' A fragments (2 KiB): read only by blocks that contain border pixels
  if (wave < 2) glds16((const char *)a.w + (7 + wave) * 1024 + lane * 16, WB + wave * 1024);
  if (tid < 32) {
    BL[tid] = a.bias[tid];
    BL[32 + tid] = a.bias2[tid] + a.bias_sc[tid];
  }
  if (tid < 12) ((uint32_t *)(smem + (tid >> 2) * RAWBYTES))[RH * RP + (tid & 3)] = 0u;
  // ---- raw patch: issue (global -> registers) / commit (registers -> LDS) ----
  // One item = 4 horizontally adjacent pixels of both planes (two 8-byte loads).  The patch starts at column 64*tx - 4 and the
  // planes are dense 2H x 2H int16 with 8-byte aligned rows (host guarantees it), so quads are aligned and lie entirely
  // inside or outside the picture.  (Pixel-wise 2-byte loads made ISSUING the next tile's loads 42 % of the tile time.)
  typedef uint32_t uint2v __attribute__((ext_vector_type(2)));
  uint2v vo[UR], vp[UR];
  int rdst[UR];  // dword index in the raw buffer of the quad's first pixel, bit 30: zero-fill, bit 29: own quad (guard statistic), -1: none
  auto issue_raw = [&](int k) {
    int tx, ty, n;
    tile_decode(k, tx, ty, n);
    const int iy0 = 2 * (ty * TH - 1) - 2, ix0 = 2 * (tx * TW - 1) - 2;
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const int it = tid + u * NT;
      const int ry = it / QW, qx = it - ry * QW;
      const int iy = iy0 + ry, ix = ix0 + 4 * qx;
      const bool in_items = it < RH * QW;
      const bool live = in_items && iy >= 0 && iy < S && ix >= 0 && ix < S;
      // bit 29: the quad lies in the part of the raw patch only THIS tile owns (rows / columns of its 32 x 64 input
      // pixels without the halo) -> counted once for the flat-content guard
      const bool own = in_items && ry >= 4 && ry < 4 + 2 * TH && qx >= 1 && qx <= 2 * TW / 4;
      rdst[u] = in_items ? (ry * RP + 4 * qx) | (live ? 0 : 1 << 30) | (own ? 1 << 29 : 0) : -1;
      const size_t oo = live ? (size_t)n * a.org_cu_stride + (size_t)iy * a.org_row_stride + ix : (size_t)n * a.org_cu_stride + ((tid * 4) & (S - 1));
      const size_t po = live ? (size_t)n * a.pred_cu_stride + (size_t)iy * a.pred_row_stride + ix : (size_t)n * a.pred_cu_stride + ((tid * 4) & (S - 1));
      vo[u] = *(const uint2v *)(a.org + oo);
      vp[u] = *(const uint2v *)(a.pred + po);
    }
  };
  auto commit_raw = [&](int k) {
    int tx, ty, n;
    tile_decode(k, tx, ty, n);
    uint32_t *raw = rawbuf(k);
    int nflat = 0;  // wave-uniform: own quads that are coherent in both planes the network sees (flat_stat_kernel's statistic)
#pragma unroll
    for (int u = 0; u < UR; ++u) {
      const bool item = rdst[u] >= 0;
      const bool zf = rdst[u] & (1 << 30);
      const int d = rdst[u] & ~(3 << 29);
      uint4v w;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int16_t o = (int16_t)(vo[u][j >> 1] >> (16 * (j & 1))), q = (int16_t)(vp[u][j >> 1] >> (16 * (j & 1)));
        w[j] = zf ? 0u : prep_pair(o, q);
      }
      if (a.flat) nflat += __builtin_popcountll(__ballot(item && (rdst[u] & (1 << 29)) && quad_near_flat(w[0], w[1], w[2], w[3])));
      if (!item) continue;
      *(uint4v *)(raw + d) = w;  // row-major: one 16-byte store per quad
    }
    if (a.flat && nflat && lane == 0) atomicAdd(a.flat + n, nflat);  // integer adds: order-independent, deterministic
  };
  // 16 contiguous bytes at an 8-byte aligned LDS address (two ds_read_b64 / one ds_read2_b64)
  auto read16 = [&](const uint32_t *raw, uint32_t dword_index) -> half8 {
    const uint2v lo = *(const uint2v *)(raw + dword_index), hi = *(const uint2v *)(raw + dword_index + 2);
    uint4v v;
    v[0] = lo[0]; v[1] = lo[1]; v[2] = hi[0]; v[3] = hi[1];
    return *(half8 *)&v;
  };

  // ---- prologue: raw patch of tile 0 committed, loads of tile 1 in flight ----
  issue_raw(0);
  commit_raw(0);
  if (nloc > 1) issue_raw(1);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0);  // (the resident-weight / border DMA of this wave as well; the prefetch of tile 1 is re-waited by its commit)
  __builtin_amdgcn_s_barrier();

  if (wave < NWS) {
    // ================= producer waves: phase 1 of tile k =================
    // composed first-layer weights (5 k-steps): registers, once per kernel; first use in front of the loop (a first use inside it would
    // make the compiler drain the raw-plane prefetch with s_waitcnt vmcnt(0) every iteration)
    half8 am[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) am[k] = *(const half8 *)((const char *)a.w + k * 1024 + lane * 16);
#pragma unroll
    for (int k = 0; k < 5; ++k) asm volatile("" ::"v"(am[k]));
    for (int k = 0; k <= nloc; ++k) {
      if (k + 1 < nloc) commit_raw(k + 1);
      if (k + 2 < nloc) issue_raw(k + 2);
      if (k < nloc) {
        int tx, ty, n;
        tile_decode(k, tx, ty, n);
        const uint32_t *raw = rawbuf(k);
        char *T = Tb + (k & 1) * TBYTES;
        // t on tile + halo (18 x 34 = 612 pixels = 20 blocks), composed 5x5 stride-2 conv -> T (fp16, LDS).  Blocks 0..17: one T row each,
        // columns 0..31 (the lanes' 16-byte reads tile 512 contiguous bytes of a raw row: conflict-free).  Blocks 18, 19: columns 32, 33.
        // Software-pipelined: the five fragment reads of the wave's NEXT block are issued before the current block's epilogue, so the
        // LDS round trip hides under ~100 VALU instructions instead of stalling the wave (the producer waves run 1 per SIMD).
        auto block_pos = [&](int pb, bool &ok, int &y1, int &x1) {
          const int m = (pb - T_H) * 32 + p;  // blocks 18, 19: index into the 2 x 18 leftover pixels
          ok = pb < T_H || m < 2 * T_H;
          y1 = pb < T_H ? pb : (ok ? m >> 1 : 0);
          x1 = pb < T_H ? p : 32 + (m & 1);
        };
        half8 bk[5];
        auto fetch = [&](int pb) {
          bool ok; int y1, x1;
          block_pos(pb, ok, y1, x1);
          const uint32_t o = (2 * y1) * RP + 2 * x1 + 4 * h;  // window origin = input (2gy-2, 2gx-2); this lane: columns +4h .. +4h+3
#pragma unroll
          for (int dy = 0; dy < 5; ++dy) bk[dy] = read16(raw, o + dy * RP);
        };
        fetch(wave);
        for (int pb = wave; pb < T_H + 2; pb += NWS) {
          bool ok; int y1, x1;
          block_pos(pb, ok, y1, x1);
          const int gy = ty * TH - 1 + y1, gx = tx * TW - 1 + x1;  // position in the H x H map
          float16v acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int dy = 0; dy < 5; ++dy) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(am[dy], bk[dy], acc, 0, 0, 0);
          // border corrections (tiles on the picture's top row / left column only; wave-uniform tests)
          if (ty == 0 || tx == 0) {
            const bool topb = ok && gy == 0, leftb = ok && gx == 0;
            if (__ballot(leftb) != 0ull) {  // Left[u] x in(2y-2+u, 0) = window (dy = u, dx = 2); Corner x in(0, 0)
              const uint32_t l4 = raw[(2 * y1 + 4) * RP + 2 * x1 + 2], c0 = raw[4 * RP + 4];
              uint4v b;
              if (h == 0) {
                b[0] = ((const uint32_t *)&bk[0])[2]; b[1] = ((const uint32_t *)&bk[1])[2];
                b[2] = ((const uint32_t *)&bk[2])[2]; b[3] = ((const uint32_t *)&bk[3])[2];
              } else {
                b[0] = l4; b[1] = topb ? c0 : 0u; b[2] = 0u; b[3] = 0u;
              }
              if (!leftb) b = uint4v{0u, 0u, 0u, 0u};
              acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8 *)(WB + lane * 16), *(half8 *)&b, acc, 0, 0, 0);
            }
            if (__ballot(topb) != 0ull) {  // Top[v] x in(0, 2x-2+v) = window (dy = 2, dx = v)
              uint4v b = *(const uint4v *)&bk[2];
              if (h == 1) { b[1] = 0u; b[2] = 0u; b[3] = 0u; }
              if (!topb) b = uint4v{0u, 0u, 0u, 0u};
              acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*(const half8 *)(WB + 1024 + lane * 16), *(half8 *)&b, acc, 0, 0, 0);
            }
          }
          if (pb + NWS < T_H + 2) fetch(pb + NWS);  // next block's fragments fly under this block's epilogue
          const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < H;  // outside: conv2's zero padding
          const uint32_t keep_mask = inside ? 0xFFFFFFFFu : 0u;
          half4 ov[4];
          static_for<4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const float4v b1q = *(const float4v *)(BL + 4 * h + 8 * q);
            const half4 none{};
            ov[q] = act_quad<q>(acc, a.acc_scale, b1q, false, none, true);  // relu(acc * scale + bias), packed arithmetic
            ((uint32_t *)&ov[q])[0] &= keep_mask;
            ((uint32_t *)&ov[q])[1] &= keep_mask;
          });
          // 16-byte stores (pair16: lanes 0-31 take channels 16qq..+7, lanes 32-63 the next 8): conflict-free at an 80-byte pixel pitch
          char *dst = T + (y1 * T_W + x1) * PS + 16 * h;
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) {
            const uint4v w = pair16(ov[2 * qq], ov[2 * qq + 1]);  // every lane takes part in the swap
            if (ok) *(uint4v *)(dst + 32 * qq) = w;
          }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  } else {
    // ================= consumer waves: phase 2 of tile k - 1 =================
    // conv2's 18 A fragments and the shortcut's 2 live in this wave's registers for the whole kernel: only activation fragments are
    // read from LDS in the MFMA loop
    half8 wf[18], as[2];
#pragma unroll
    for (int i = 0; i < 18; ++i) wf[i] = *(const half8 *)(W2 + i * 1024 + lane * 16);
#pragma unroll
    for (int k = 0; k < 2; ++k) as[k] = *(const half8 *)((const char *)a.w + (5 + k) * 1024 + lane * 16);
#pragma unroll
    for (int k = 0; k < 2; ++k) asm volatile("" ::"v"(as[k]));
    const int cw = wave - NWS;
    for (int k = 0; k <= nloc; ++k) {
      if (k + 1 < nloc) commit_raw(k + 1);
      if (k + 2 < nloc) issue_raw(k + 2);
      if (k >= 1) {
        int tx, ty, n;
        tile_decode(k - 1, tx, ty, n);
        const uint32_t *raw = rawbuf(k - 1);
        const char *T = Tb + ((k - 1) & 1) * TBYTES;
        // conv2(t) from T + shortcut (composed 3x3 stride-2 conv of the raw planes, fp32) + relu -> b0.  The consumer waves run one per
        // SIMD, so LDS latency is hidden by depth, not by other waves: activation fragments are read PD items ahead of the MFMA that
        // consumes them, and the first PD fragments of the wave's next block are issued before the current block's epilogue.
        constexpr int PD = CFG_SB_PD;
        half8 bf[PD], sb0, sb1;
        auto frag = [&](int pb, int item) -> half8 {
          const int base = (pb * T_W + p) * PS + h * 16;
          const int tp = item >> 1, ks = item & 1, dy = tp / 3, dx = tp - dy * 3;
          return *(const half8 *)(T + base + (dy * T_W + dx) * PS + ks * 32);
        };
        auto prefetch = [&](int pb) {
          // shortcut taps (by, bx) at input (2gy+by-1, 2gx+bx-1) = raw (2y+3+by, 2x+3+bx): lanes h = 0 / 1 read rows by = 0 / 1 (k-step 5),
          // every lane row by = 2 (k-step 6, upper half: zero weights), columns 2x+2 .. 2x+5 (the first one carries a zero weight)
          sb0 = read16(raw, (2 * pb + 3 + h) * RP + 2 * p + 2);
          sb1 = read16(raw, (2 * pb + 5) * RP + 2 * p + 2);
#pragma unroll
          for (int i = 0; i < PD; ++i) bf[i] = frag(pb, i);
        };
        prefetch(cw);
        for (int pb = cw; pb < TH * TW / 32; pb += NWS) {
          const int y = pb, x = p;
          float16v acc, accs;
#pragma unroll
          for (int r = 0; r < 16; ++r) { acc[r] = 0.f; accs[r] = 0.f; }
          accs = __builtin_amdgcn_mfma_f32_32x32x16_f16(as[0], sb0, accs, 0, 0, 0);
          accs = __builtin_amdgcn_mfma_f32_32x32x16_f16(as[1], sb1, accs, 0, 0, 0);
#pragma unroll
          for (int item = 0; item < 18; ++item) {
            const half8 cur = bf[item % PD];
            if (item + PD < 18) bf[item % PD] = frag(pb, item + PD);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[item], cur, acc, 0, 0, 0);
          }
          if (pb + NWS < TH * TW / 32) prefetch(pb + NWS);  // next block's first fragments fly under this block's epilogue
          half4 hq[4];
          static_for<4>([&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const float4v bq = *(const float4v *)(BL + 32 + 4 * h + 8 * q);
#pragma unroll
            for (int ep = 0; ep < 2; ++ep) {  // relu(conv2 + (shortcut * scale + bias)), the sum in fp32, packed arithmetic
              const float2v c2 = {acc[4 * q + 2 * ep], acc[4 * q + 2 * ep + 1]}, s2 = {accs[4 * q + 2 * ep], accs[4 * q + 2 * ep + 1]};
              const float2v b2 = {bq[2 * ep], bq[2 * ep + 1]};
              const float2v x2 = c2 + (s2 * a.acc_scale + b2);
              half2v h2 = __builtin_convertvector(x2, half2v);
              h2 = __builtin_elementwise_max(h2, (half2v){(_Float16)0, (_Float16)0});
              hq[q][2 * ep] = h2[0];
              hq[q][2 * ep + 1] = h2[1];
            }
          });
          const size_t ob = ((((size_t)n << h_l) + ty * TH + y) << h_l) * 32 + (size_t)(tx * TW + x) * 32 + 8 * h;
#pragma unroll
          for (int qq = 0; qq < 2; ++qq) *(uint4v *)((_Float16 *)a.y + ob + 16 * qq) = pair16(hq[2 * qq], hq[2 * qq + 1]);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Heads + argmax (arch:282-297, EncCu.cpp:913-921).  One workgroup per CU; fp32 throughout.
// feat = (sum of the GAP partial sums written by the stage's last conv) / HW
// logits_k = W_k . [feat (C floats), poc, qp] + b_k ; split = first maximal index (torch.argmax).
// Every class of a head runs the identical operation sequence, so identical rows tie exactly.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void heads_kernel(const HeadArgs a) {
  __shared__ float feat[MLT_MAX_HEADS_K][256 + 2];
  __shared__ float lg[MLT_MAX_LOGITS_K];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float fpoc = (float)a.poc[n], fqp = (float)a.qp[n];  // EncCu.cpp:881-882 (int -> float, exact)
  // GAP features of ALL heads first (one barrier instead of two per head: this kernel is pure latency, and it is a
  // visible share of the one-CU-per-call path).  The partial sums are added in slot order as before -- same rounding --
  // but eight loads are in flight at a time instead of one.
  for (int hd = 0; hd < a.n_heads; ++hd) {
    const int C = a.c[hd], slots = a.slots[hd];
    if (tid < C) {
      const float *g = a.gap[hd] + (size_t)n * slots * C + tid;
      float s = 0.f;
      for (int i = 0; i < slots; i += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = i + u < slots ? g[(size_t)(i + u) * C] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (i + u < slots) s += v[u];
      }
      feat[hd][tid] = s / (float)a.hw[hd];
    }
    if (tid == 0) { feat[hd][C] = fpoc; feat[hd][C + 1] = fqp; }
  }
  __syncthreads();
  int lo = 0;
  for (int hd = 0; hd < a.n_heads; ++hd) {
    const int C = a.c[hd], K = a.classes[hd];
    for (int k = wave; k < K; k += 4) {
      const float *w = a.w[hd] + (size_t)k * (C + 2);
      float s = 0.f;
      for (int i = lane; i < C + 2; i += 64) s += w[i] * feat[hd][i];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
      if (lane == 0) lg[lo + k] = s + a.b[hd][k];
    }
    lo += K;
  }
  __syncthreads();
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
// Parity guard of the fast arithmetic.  fp16 rounding noise is averaged away by the global pooling only where
// neighbouring pixels DIFFER: on exactly-constant areas every pixel carries the same rounding error (measured: a constant
// 128x128 CU reaches |dlogit| 1.3e-3, ordinary content 2-7e-4), and nearly so where they differ by a few LSB (dither, low
// contrast, gentle ramps: rms 1.2-1.5x the textured one, measured per content class in round 3).  flat_stat_kernel counts,
// per CU, the aligned 4-pixel quads whose org values and whose |org - pred| values each span <= MLT_FLAT_RANGE; guard_select_kernel lists the CUs whose count reaches the
// threshold (and, optionally, those whose decision-head margin is small); the host runtime re-evaluates exactly those
// CUs with the exact (hi, lo) arithmetic (gather -> exact network -> scatter).  Integer statistics: deterministic.
// ---------------------------------------------------------------------------------------------
template <bool ALIGNED>
__global__ __launch_bounds__(256) void flat_stat_kernel(const FlatStatArgs a) {
  __shared__ int wsum[4];
  const int n = blockIdx.x, tid = threadIdx.x, S = 1 << a.s_l, qrow_l = a.s_l - 2;  // quads per row = S / 4
  const int nquads = S << qrow_l;
  const int16_t *o = a.org + (size_t)n * a.org_cu_stride, *p = a.pred + (size_t)n * a.pred_cu_stride;
  int cnt = 0;
  for (int q = tid; q < nquads; q += 256) {
    const int y = q >> qrow_l, x = (q & ((1 << qrow_l) - 1)) * 4;
    int16_t vo[4], vp[4];
    if constexpr (ALIGNED) {
      typedef uint32_t uint2v __attribute__((ext_vector_type(2)));
      const uint2v wo = *(const uint2v *)(o + (size_t)y * a.org_row_stride + x), wp = *(const uint2v *)(p + (size_t)y * a.pred_row_stride + x);
#pragma unroll
      for (int j = 0; j < 4; ++j) { vo[j] = (int16_t)(wo[j >> 1] >> (16 * (j & 1))); vp[j] = (int16_t)(wp[j >> 1] >> (16 * (j & 1))); }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) { vo[j] = o[(size_t)y * a.org_row_stride + x + j]; vp[j] = p[(size_t)y * a.pred_row_stride + x + j]; }
    }
    // what the network sees (uint16 cast, absdiff, clip)
    cnt += quad_near_flat(prep_pair(vo[0], vp[0]), prep_pair(vo[1], vp[1]), prep_pair(vo[2], vp[2]), prep_pair(vo[3], vp[3])) ? 1 : 0;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
  if ((tid & 63) == 0) wsum[tid >> 6] = cnt;
  __syncthreads();
  if (tid == 0) a.flat[n] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// one workgroup; ascending index list by a block-wide prefix scan (deterministic order)
__global__ __launch_bounds__(1024) void guard_select_kernel(const GuardSelectArgs a) {
  __shared__ int part[1024];
  const int tid = threadIdx.x, per = (a.n + 1023) / 1024, lo = tid * per, hi = lo + per < a.n ? lo + per : a.n;
  auto selected = [&](int i) -> bool {
    bool s = a.flat && a.flat[i] >= a.flat_thr;
    if (a.logits && a.margin > 0.f) {
      const float *l = a.logits + (size_t)i * a.n_logits + a.head_off;
      float t1 = -3.4e38f, t2 = -3.4e38f;
      for (int c = 0; c < a.head_classes; ++c) {
        if (l[c] > t1) { t2 = t1; t1 = l[c]; } else if (l[c] > t2) t2 = l[c];
      }
      s = s || !(t1 - t2 >= a.margin);  // also catches NaN
    }
    return s;
  };
  int c = 0;
  for (int i = lo; i < hi; ++i) c += selected(i) ? 1 : 0;
  part[tid] = c;
  __syncthreads();
  for (int off = 1; off < 1024; off <<= 1) {  // inclusive Hillis-Steele scan
    const int v = tid >= off ? part[tid - off] : 0;
    __syncthreads();
    part[tid] += v;
    __syncthreads();
  }
  int pos = part[tid] - c;
  for (int i = lo; i < hi; ++i)
    if (selected(i)) a.idx[pos++] = i;
  if (tid == 1023) *a.count = part[1023];
}

// grid (k, 2): plane blockIdx.y of the blockIdx.x-th selected CU -> dense staging
__global__ __launch_bounds__(256) void guard_gather_kernel(const GuardGatherArgs a) {
  const int j = blockIdx.x, src = a.idx[j], S = 1 << a.s_l, tid = threadIdx.x;
  const bool is_pred = blockIdx.y == 1;
  const int16_t *s = is_pred ? a.pred + (size_t)src * a.pred_cu_stride : a.org + (size_t)src * a.org_cu_stride;
  const long rs = is_pred ? a.pred_row_stride : a.org_row_stride;
  int16_t *d = (is_pred ? a.g_pred : a.g_org) + ((size_t)j << (2 * a.s_l));
  for (int i = tid; i < S * S; i += 256) d[i] = s[(size_t)(i >> a.s_l) * rs + (i & (S - 1))];
  if (tid == 0 && !is_pred) { a.g_poc[j] = a.poc[src]; a.g_qp[j] = a.qp[src]; }
}

__global__ __launch_bounds__(256) void guard_scatter_kernel(const GuardScatterArgs a) {
  const int t = blockIdx.x * 256 + threadIdx.x, j = t / (a.n_logits + 1), e = t - j * (a.n_logits + 1);
  if (j >= a.k) return;
  const int dst = a.idx[j];
  if (e == a.n_logits) a.split[dst] = a.g_split[j];
  else if (a.logits) a.logits[(size_t)dst * a.n_logits + e] = a.g_logits[(size_t)j * a.n_logits + e];
}

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
  const int patch_lds = (DMA ? 2 : (NSPLIT == 2 ? 2 : 1)) * a.patch_bytes;
  const int lds = patch_lds + NBUF * (NSPLIT >= 2 ? 2 : 1) * GT * (KC / 16) * CBT * 1024 + extra_lds;
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
#ifndef CFG_GTE_S2
#define CFG_GTE_S2 1
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
      out->lat = (!exact && r.lat && r.wcb * r.wc == (r.cout == 64 ? 2 : 4)) ? 1 : 0;
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
    return launch_conv_t<CIN, COUT, STRIDE, 9, SCF, KCE, 2, WCB, WPB, WC, WPE, GTE, RBE, UNE, 1, false>(a, grid_x, extra_lds, st);          \
  }
#define CONV_CASE(CIN, COUT, STRIDE, SCF, KCF, KCE, WCB, WPB, WC, WP, GTF, GTE, RBF, RBE, UNF, UNE, MWF) \
  CONV_CASE2(CIN, COUT, STRIDE, SCF, KCF, KCE, WCB, WPB, WC, WP, WP, GTF, GTE, RBF, RBE, UNF, UNE, MWF)

// stride-2 convs always carry their block's projection shortcut (layer0.0's lives in stem5_kernel).
bool mlt_conv_has_centre_variant(int cin, int cout) { return (cin == cout && (cin == 128 || cin == 256)) || (cin == 128 && cout == 256); }

// nsplit: 1 fast, 2 exact (weights and activations hi+lo), 3 weights hi+lo only (the tiling of the exact kernels, 2 MFMAs, single activation planes)
hipError_t mlt_launch_conv(int cin, int cout, int stride, int nsplit, int variant, const ConvArgs &a, int grid_x, int extra_lds, hipStream_t st) {
  const bool exact = nsplit >= 2;
  const bool dma = variant == MLT_CONV_DMA;
  if (variant == MLT_CONV_CENTRE && nsplit == 3) return hipErrorInvalidValue;  // (1x1 maps: small-CU models only, which do not use this tier)
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
  CONV_CASE(32, 32, 1, false, 32, 32, 1, CFG_32_WPB, 1, CFG_32_WP, 9, 3, 1, 2, 5, 3, CFG_32_MINW)
  CONV_CASE(32, 64, 2, true, 32, 32, CFG_3264_WCB, 1, CFG_3264_WC, CFG_3264_WP, CFG_3264_GT, CFG_GTE_S2, CFG_3264_RB, 2, CFG_3264_UN, 3, CFG_32_MINW)
  CONV_CASE(64, 64, 1, false, 64, 32, CFG_64_WCB, CFG_64_WPB, CFG_64_WC, CFG_64_WP, CFG_64_GT, CFG_GTE_S1, CFG_S1_RB, 2, 6, 3, CFG_S1_MINW)
  CONV_CASE(64, 128, 2, true, 32, 32, CFG_S2BIG_WCB, CFG_S2BIG_WPB, CFG_S2BIG_WC, CFG_S2BIG_WP, 2, CFG_GTE_S2, 2, 2, 5, 3, CFG_S2_MINW)
  CONV_CASE2(128, 128, 1, false, 64, 32, CFG_BIG_WCB, CFG_128_WPB, CFG_BIG_WC, CFG_BIG_WP, CFG_BIG_WP_EXACT, CFG_BIG_GT, CFG_GTE_S1, CFG_S1_RB, 2, 3, 2, CFG_S1_MINW)
  CONV_CASE(128, 256, 2, true, 32, 32, CFG_S2BIG_WCB, CFG_S2BIG_WPB, CFG_S2BIG_WC, CFG_S2BIG_WP, CFG_S2B_GT, CFG_GTE_S2, 2, 2, 5, 3, CFG_S2_MINW)
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
template <class K> static hipError_t launch_chain_t(K kern, DeviceOnce &once, const ChainArgs &a, int grid_x, int threads, int lds, hipStream_t st) {
  if (hipError_t e = ensure_big_lds(kern, once); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(grid_x), dim3(threads), lds, st, a);
  return hipGetLastError();
}
hipError_t mlt_launch_chain(int c, int h, bool with_s2, bool oob_zero, const ChainArgs &a, int grid_x, hipStream_t st) {
  static_assert(CFG_BIG_GT == 3 && CFG_BIG_WCB == 2 && CFG_BIG_WC == 2, "chain_kernel reads the packing of the stand-alone 128->128 / 256->256 layers");
  constexpr int lds = 64 * 1024 + 2 * (CFG_BIG_GT * 4 * 4 * 1024);  // 64 KiB activation + two 48 KiB weight steps = all of the LDS
  static DeviceOnce once[10];
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

hipError_t mlt_launch_block32(const Block32Args &a, int grid_x, hipStream_t st) {
  constexpr int lds = (20 * 36 + 18 * 34) * 80 + 2 * 18 * 1024;
  static DeviceOnce once;
  if (hipError_t e = ensure_big_lds(block32_kernel, once); e != hipSuccess) return e;
  hipLaunchKernelGGL(block32_kernel, dim3(grid_x), dim3(512), lds, st, a);
  return hipGetLastError();
}

hipError_t mlt_launch_stem_block(const StemBlockArgs &a, int grid_x, hipStream_t st) {
  constexpr int lds = 3 * (39 * 72 * 4 + 16) + 2 * (18 * 34 * 80) + 18 * 1024 + 256 + 2 * 1024;  // 3 raw + 2 T buffers + conv2 weights + biases + border k-steps = 149 KiB
  static DeviceOnce once;
  if (hipError_t e = ensure_big_lds(stem_block_kernel, once); e != hipSuccess) return e;
  hipLaunchKernelGGL(stem_block_kernel, dim3(grid_x), dim3(128 * CFG_SB_NWS), lds, st, a);  // one workgroup per CU, two pipeline stages inside
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
