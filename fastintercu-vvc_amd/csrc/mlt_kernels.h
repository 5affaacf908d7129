// mlt_kernels.h -- launch interface between the host runtime (mlt_api.cpp) and mlt_kernels.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MLT_MAX_HEADS_K 4
#define MLT_MAX_LOGITS_K 16
// Flat-content guard statistic (round 3: widened from "exactly constant"): an aligned 4-pixel quad is NEAR-FLAT when the RANGE (max - min)
// of its four org values AND of its four |org - pred| values -- as the network sees them: uint16 cast, absdiff, clip to 10 bits --
// is <= MLT_FLAT_RANGE -- +-1 LSB dither, sensor noise on smooth areas, every exactly-constant area (sky, letterbox bars,
// screen content) -- or the four values are LINEAR to within one step (ramps of any slope); per plane either test may hold.  Content
// on which neighbouring pixels carry correlated fp16 rounding errors that the global pooling cannot average away.
// Round 4: a quad is EXACTLY FLAT when each plane is constant or exactly linear (range 0, or both second differences 0).  A CU is flagged
// when >= 1/8 of its quads are exactly flat (identical activations at every pixel: the error mechanism proper, 1.2 - 2.3e-3 measured on
// constant CUs whatever the weight arithmetic) OR >= 1/2 are near-flat.  With the single 1/8 rule on near-flat quads, 22 % of the CUs of
// the natural-statistics class (synth.natural_patches: smooth areas + sensor noise) were re-run exactly although their single-pass error
// is no larger than that of textured CUs (profiles/r04*_natural_probe.txt); the two-level rule flags 3 % of them, and the calibration
// set carries the content it lets through (a near-flat band over 40-48 % of the quads).
// Round 6: the range is 6, not 8.  Measured on nine seeded weight sets and the trained families in their shipped tiers WITHOUT the guard
// (scripts/r06_flat_guard_need_probe.py, profiles/r06i_flat_guard_need*.txt): what needs the guard is content whose variation stays near the
// first activations' fp16 step -- constant areas (1.5 - 2.7e-3), +-1 LSB dither (heavier tail, up to 9.5e-4), low-contrast texture of amplitude
// <= 4 (like texture for most sets, but ONE logit of 405,504 at 1.01e-3 in the most marginal tier when a range of 4 let that class through:
// profiles/r06j_tail_probe.txt) -- while the 2.3 - 3 % of the natural-statistics CUs the range of 8 flagged (near-flat fractions 0.50 - 0.72,
// made of quads of range 7 - 8: slow gradients + sensor noise) err like their unflagged neighbours (max 6.1e-4 against 7.4e-4): 125 exact
// re-runs per 4096-CU batch of natural scenes for nothing (1.02 M -> 0.83 M CU/s).  With 6 every class the rule was built for stays guarded
// (dither 100 % near-flat, amplitude-4 texture 69 - 73 %, ramps, constants) and natural scenes are flagged at 0 % (largest near-flat fraction 0.47).
#define MLT_FLAT_RANGE 6
#define MLT_FLAT_EXACT_SHIFT 16   // the per-CU statistic packs both counts: near-flat quads in bits 0-15, exactly flat quads in bits 16-31

struct ConvArgs {
  const void *x;      // input  [n][Hin][Hin][CIN]  fp16 NHWC
  void *y;            // output [n][Hout][Hout][COUT] fp16 NHWC
  const void *w;      // packed fp16 weights (mlt_model.cpp: pack_conv)
  const float *bias;  // folded BN bias per output channel
  const void *res;    // optional residual, same shape as y (NULL: none)
  const void *zero;   // >= 64 KiB of zeros (DMA staging fetches padding from here)
  int n;
  int ntiles;                  // logical tiles (grid.x may be smaller: workgroups are persistent over tiles)
  int hin_l, hout_l;           // log2 of input / output height (= width)
  int tw_l, th_l, spw_l;       // log2 of tile width, tile height, samples per workgroup
  int ph, pw, rp, half;        // patch rows, cols, row pitch (pixels), parity-split half width
  uint32_t pw_magic, ph_magic, rp_magic; // ceil(2^32 / d), d >= 2
  int patch_bytes;             // LDS bytes reserved for the patch (multiple of 1024)
  int relu;
  int y_c16;                   // 1: y is written chunk-major, [n][COUT/16][Hout*Hout][16] -- the layout the next stage's whole-stage
                               // kernel stages its 16-channel input patches from (contiguous 32-byte pixels instead of 32-byte pieces
                               // of 128/256-byte NHWC pixels: measured 2.2x HBM over-fetch on those)
  int ysc_c16;                 // 1: y_sc chunk-major as well (its reader is the 64-channel chain's residual load: one cache line per
                               // lane quad instead of four)
  float acc_scale;             // accumulators are multiplied by this before the bias (weights are stored * 2^s)
  // SC variant: second output = bn(conv1x1_stride2(x)) (projection shortcut, arch:44-50)
  void *y_sc;
  const float *bias_sc;
  // fp32 global-average-pool partial sums [n][gap_slots][COUT] (NULL: none); gap_l = log2(lanes per sample)
  float *gap;
  int gap_slots, gap_l;
  // exact mode (NSPLIT == 2): byte offsets from each hi plane to its lo plane
  size_t x_lo_off, y_lo_off, res_lo_off, ysc_lo_off, w_lo_off;  // exact arithmetic: byte offsets hi plane -> lo plane (x_lo_off / res_lo_off == 0: that input has no lo plane, its lo part is zero)
  int lo8_scale;               // NSPLIT == 5 (hi fp16 + FP8 lo plane): E8M0 scale byte of the lo plane, replicated (mlt_model.h: lo8_exp)
  int xl_sa0, xl_sa1, xl_sb1;  // NSPLIT == 6 (exact-lite): replicated E8M0 bytes -- A operand lanes 0-31 (e4m3 Wl: 127 - xl_ewl) / lanes 32-63 (e4m3 Wh: 127 - xl_ewh),
                               // B operand lanes 32-63 (e4m3 Xl: 127 - MLT_XL_LO_EXP); B lanes 0-31 (e4m3 Xh) are unscaled
};

struct ConvCfg { int kc, ct, mt, gt, dma, mt_dma, lat, gt_w2; };  // gt_w2 (exact packing only): taps per weight step of the hi+lo-WEIGHTS tier (NSPLIT = 3)  // cin chunk, couts / pixels per workgroup, taps per weight step, LDS-DMA staging variant (0 none, 1 resident, 2 ring) and its pixels per workgroup

struct Stem5Args {
  const int16_t *org, *pred;   // Pel planes
  long org_row_stride, org_cu_stride, pred_row_stride, pred_cu_stride;  // in elements
  const void *w;               // [plane][7 k-steps][64 lanes][8 halves]: 5 composed 5x5(+border) steps, 2 shortcut steps
  const float *bias, *bias_sc;
  void *y, *y_sc;              // t and sc, [n][S/2][S/2][32] fp16
  size_t y_lo_off, ysc_lo_off, w_lo_off;
  float acc_scale;
  int n, s_l, hout_l, tw_l, th_l, spw_l;
  int rh, rw, halfw;           // raw patch rows, cols, half row pitch (columns are parity-split)
  uint32_t rw_magic, rh_magic;
};

struct Block32Args {
  const void *x;               // block input  [n][H][H][32] fp16 (also the residual)
  void *y;                     // block output [n][H][H][32] fp16
  const void *w1, *w2;         // packed fp16 weights of conv1 / conv2 (18 KiB each, fast mode layout)
  const float *bias1, *bias2;  // folded BN biases
  int n, h_l, ntiles;          // H = 1 << h_l (>= 32); tiles of 16 x 32 output pixels (hi+lo-weights form: 8 x 32)
  // hi+lo-weights form: byte offsets hi plane -> lo plane, accumulator scales (weights are stored * 2^s)
  size_t w1_lo_off, w2_lo_off;
  float scale1, scale2;
};

struct StemBlockArgs {
  const int16_t *org, *pred;   // Pel planes
  long org_row_stride, org_cu_stride, pred_row_stride, pred_cu_stride;  // in elements
  const void *w;               // composed first-layer weights (Stem5Args.w layout, fast plane)
  const void *w2;              // layer0.0.conv2 packed weights (18 KiB)
  const float *bias, *bias_sc, *bias2;
  void *y;                     // b0 [n][H][H][32] fp16
  int32_t *flat;               // NULL, or [n] zero-initialised: += number of near-constant (MLT_FLAT_RANGE) 4-pixel quads of each CU
                               // (flat-content guard; same statistic as flat_stat_kernel)
  float acc_scale;             // composed-weight storage scale (2^-12)
  int n, hout_l, ntiles;       // H = 1 << hout_l (>= 32), picture 2H x 2H; tiles of 16 x 32 output pixels
  // hi+lo-weights form: byte offsets hi plane -> lo plane of w / w2, accumulator scale of conv2
  size_t w_lo_off, w2_lo_off;
  float scale2;
};

// All of layer0 of 128 x 128 CUs in one launch (layer0_stream_kernel, round 5): stem_block_kernel's and block32_kernel's arithmetic, streamed row by
// row through LDS rings by one persistent 16-wave workgroup per CU -- b0 never reaches HBM, no halo is staged or computed twice.
#define MLT_L0F_LDS_BYTES 161040 /* ... with the stride-2 conv of layer1 as a fifth stage: rings of 6 / 8 / 6 / 6 rows, 64 + 64 more biases */
#define MLT_L0_LDS_BYTES 145872  /* 3 rings x 8 map rows + zero row (66 px x 80 B), 16 + 1 raw rows (136 dwords), biases, 4 KiB of first-layer k-steps, 2 counters */
struct Layer0Args {
  const int16_t *org, *pred;   // Pel planes, 128 x 128 per CU
  long org_row_stride, org_cu_stride, pred_row_stride, pred_cu_stride;  // in elements (8-byte aligned quads: see stem_block_kernel)
  const void *w;               // composed first-layer weights (StemBlockArgs.w)
  const void *w2, *w3, *w4;    // layer0.0.conv2, layer0.1.conv1, layer0.1.conv2: packed fp16, 18 KiB each (fast plane)
  const float *bias, *bias_sc, *bias2, *bias3, *bias4;
  void *y;                     // layer0 output [n][64][64][32] fp16
  int32_t *flat;               // NULL, or [n] zero-initialised (flat-content guard statistic, as StemBlockArgs.flat)
  float acc_scale;             // composed-weight storage scale
  int n;
  // fifth stage (layer0_stream_kernel<true>): layer1.0.conv1 + shortcut, packed as for conv_mfma_kernel<32, 64, 2, ...> (ct 64, kc 32, 10 taps)
  const void *w5;
  const float *bias5, *bias5_sc;
  void *y_t, *y_sc;            // t [n][32][32][64] fp16 NHWC; sc chunk-major [n][4][1024][16] (ConvArgs.ysc_c16: what the 64-channel chain reads)
  float scale5;
};

// The three stride-1 convs of layer1 (64 channels at 32 x 32) as a streaming pipeline (layer1_stream_kernel, round 5): chain_kernel<64>'s
// arithmetic, weights resident in registers, rows through LDS rings, b0 never in HBM.
#define MLT_L1_LDS_BYTES 159008  /* rings of 4 / 8 / 4 map rows + zero row (34 px x 144 B), 2 sc rows, 16 accumulator hand-over blocks of 4 KiB, biases */
struct Layer1Args {
  const void *t;               // [n][32][32][64] fp16 NHWC: relu(bn1(conv1 x)) of layer1.0
  const void *sc;              // the projection shortcut, chunk-major [n][4][1024][16] fp16 (ConvArgs.ysc_c16)
  const void *w[3];            // layer1.0.conv2, layer1.1.conv1, layer1.1.conv2: packed fp16 (ct 64, kc 64: 72 KiB each)
  const float *bias[3];
  float scale[3];
  void *y;                     // stage output [n][32][32][64] fp16, NHWC or chunk-major (y_c16)
  int y_c16;
  float *gap;                  // fp32 GAP partial sums [n][gap_slots][64]
  int gap_slots, n;
};

// Fused chain of stride-1 3x3 convs on whole samples (chain_kernel): the BasicBlock tail of a stage,
//   b0 = relu(bn2(conv2(t)) + sc) ; t1 = relu(bn1(conv1(b0))) ; out = relu(bn2(conv2(t1)) + b0)      (arch:52-57)
// with every intermediate kept on chip (activations in LDS, b0 as the residual in registers).
struct ChainConv {
  const void *w;       // packed fp16 weights (same packing as the stand-alone conv of this layer)
  size_t w_lo_off;     // hi+lo-weights form (chain_kernel<..., W2>): byte offset of the lo plane
  int lo8_scale;       // ... with an FP8 lo plane (LO8): the E8M0 scale byte 127 - lo8_exp, replicated into all four bytes
  const float *bias;   // folded BN bias
  float acc_scale;
  int relu;
  int res_mode;        // 0 none, 1 from `res` (HBM, same layout as the output), 2 from the tile saved by an earlier conv
  int save;            // 1: keep this conv's activated output tile in registers as a later conv's residual
  const void *res;
  void *y;             // non-final conv: also write the activated output to HBM (same bytes as NHWC, but in the writing wave's own
                       // order: it is read back only by that wave) -- the later residual when the chain does not keep it in
                       // registers (64-channel stage: 128 accumulator registers leave no room)
};
struct ChainArgs {
  const void *x;       // [n][H][H][C] fp16: input of the first conv; S2 variant: the STAGE input [n][2H][2H][C/2]
  // S2 variant (whole stage in one launch): the stage's first conv -- 3x3 stride 2 + its 1x1 stride-2 projection shortcut
  // (arch:44-55) -- runs in front of the chain; its weights are packed for 16-channel chunks (mlt_model.cpp: conv1_s2c)
  const void *s2_w;
  const float *s2_bias, *s2_bias_sc;
  float s2_scale;
  const void *zero;    // >= 64 KiB of zeros (padding source of the patch DMA)
  ChainConv cv[3];
  int nconv;           // 2 or 3
  void *y;             // [n][H][H][C] fp16 output of the last conv, or NULL (only the GAP sums are needed)
  int y_c16;           // y chunk-major (ConvArgs.y_c16)
  int x_c16;           // S2: the stage input x is chunk-major
  int res0_c16;        // cv[0].res (the shortcut sc from HBM) is chunk-major (ConvArgs.ysc_c16)
  float *gap;          // fp32 GAP partial sums of the last conv's output [n][gap_slots][C], or NULL
  int gap_slots, gap_l;
  int n;
};
hipError_t mlt_launch_chain(int c, int h, bool with_s2, bool oob_zero, bool w2, const ChainArgs &a, int grid_x, hipStream_t st);  // oob_zero: see mlt_probe_lds_oob; w2: hi+lo weights (two planes, no stride-2 front conv, oob_zero only)
bool mlt_chain_supported(int c, int h);
hipError_t mlt_probe_lds_oob(int *d_ok, hipStream_t st);  // *d_ok = 1 iff DS reads beyond the LDS allocation return zeros on this device
bool mlt_stage_supported(int c, int h);  // ... including the stage's stride-2 conv + shortcut (S2 variant)

struct HeadArgs {
  const float *gap[MLT_MAX_HEADS_K];  // GAP partial sums [n][slots][C] fp32 (written by the stage's last conv)
  int slots[MLT_MAX_HEADS_K];
  const float *w[MLT_MAX_HEADS_K];    // [classes][C+2] fp32
  const float *b[MLT_MAX_HEADS_K];
  int c[MLT_MAX_HEADS_K], hw[MLT_MAX_HEADS_K], classes[MLT_MAX_HEADS_K];
  int n_heads, decision_head;
  const int32_t *poc, *qp;
  float *logits;   // [n][sum classes] or NULL
  int32_t *split;  // [n]
  // Single-CU launches (n == 1, mlt_predict's captured graph): the guard selection rides on this kernel instead of a launch of its own
  // (guard_select_kernel: 5 us of a 160 us call).  g_count != NULL: g_count[0] = 1 and g_idx[0] = 0 when CU 0 is flagged by the
  // flat-content statistic (g_flat, thresholds as GuardSelectArgs) or by the decision-head margin (g_margin > 0), else g_count[0] = 0;
  // g_flat[0] is CLEARED afterwards (consume-and-clear: the next call's first kernel adds into it, no memset node in the graph).
  int32_t *g_count, *g_idx, *g_flat;
  int g_flat_thr, g_near_thr;
  float g_margin;
  // Round 6, the MAGNITUDE guard (mlt_api.cpp: SizeState.mag_thr): mag[n] <- max over all logits of sum_k |w_ck feat_k| -- the size of the
  // feature-driven part of the logits, which is what the fp16 pipeline's RELATIVE error acts on (poc, qp and the bias enter exactly).  NULL: not
  // wanted.  g_mag_thr > 0 (single-CU launches): CU 0 is also flagged when its magnitude exceeds the threshold (NaN included).
  float *mag;
  float g_mag_thr;
  // Round 6: the selection rides on this kernel for BATCHES too (g_next != NULL; guard_select_kernel -- a launch of its own, 23 us + a launch gap of a 4 ms step --
  // stays behind MLT_TUNING=1 MLT_GUARD_SELECT_KERNEL=1).  g_count[0] is then a running count, ZERO on entry: a workgroup whose CU is flagged appends it to g_idx
  // (the list is UNORDERED -- the exact re-run treats every CU independently, so the results do not depend on the order); workgroup 0 zeroes g_next[0], the counter
  // the slot's next launch will count on (the runtime alternates between two counters: the one zeroed here was read by the previous launch's count copy, which
  // precedes this kernel in stream order).  No atomics, no fences on the unflagged path.
  int32_t *g_next;
};

// ---- parity guard (fast arithmetic): device-side selection of the CUs that are re-evaluated with the exact arithmetic ----
struct FlatStatArgs {
  const int16_t *org, *pred;   // Pel planes
  long org_row_stride, org_cu_stride, pred_row_stride, pred_cu_stride;  // in elements
  int32_t *flat;               // [n] number of near-constant 4-pixel quads (MLT_FLAT_RANGE)
  int n, s_l;                  // CU size S = 1 << s_l
};
struct GuardSelectArgs {
  const int32_t *flat;         // [n] (FlatStatArgs.flat) or NULL
  const float *logits;         // [n][n_logits] or NULL (margin test off)
  int32_t *idx;                // [n] out: indices of the selected CUs, ascending
  int32_t *count;              // [1] out
  int n, n_logits, head_off, head_classes;
  int flat_thr;                // select when the count of EXACTLY flat quads (flat >> MLT_FLAT_EXACT_SHIFT) >= flat_thr (flat != NULL) ...
  int near_thr;                // ... or the count of near-flat quads (flat & 0xFFFF) >= near_thr
  float margin;                // select when top1 - top2 of the decision head < margin (logits != NULL, margin > 0)
  const float *mag;            // [n] (HeadArgs.mag) or NULL: select when mag > mag_thr (round 6: the magnitude guard)
  float mag_thr;
};
struct GuardGatherArgs {
  const int16_t *org, *pred;
  long org_row_stride, org_cu_stride, pred_row_stride, pred_cu_stride;
  const int32_t *poc, *qp, *idx;
  int16_t *g_org, *g_pred;     // dense [k][S][S]
  int32_t *g_poc, *g_qp;       // [k]
  int k, s_l;
};
struct GuardScatterArgs {
  const int32_t *idx, *g_split;
  const float *g_logits;
  int32_t *split;              // [n]
  float *logits;               // [n][n_logits] or NULL
  int k, n_logits;
};
hipError_t mlt_launch_flat_stat(const FlatStatArgs &a, bool aligned8, hipStream_t st);
hipError_t mlt_launch_guard_select(const GuardSelectArgs &a, hipStream_t st);
hipError_t mlt_launch_guard_gather(const GuardGatherArgs &a, hipStream_t st);
hipError_t mlt_launch_guard_scatter(const GuardScatterArgs &a, hipStream_t st);

enum { MLT_CONV_DEFAULT = 0, MLT_CONV_DMA = 1, MLT_CONV_LATENCY = 2, MLT_CONV_CENTRE = 3 };  // kernel variant of a layer shape
bool mlt_conv_has_centre_variant(int cin, int cout);  // 1x1 (centre-tap) instantiation for stride-1 layers on 1x1 maps
hipError_t mlt_launch_conv(int cin, int cout, int stride, int nsplit, int variant, const ConvArgs &a, int grid_x, int extra_lds, hipStream_t st);  // nsplit: 1 fast, 2 exact, 3 weights hi+lo only (exact tiling), 4 weights hi+lo on the FAST tiling (MLT_MODEL_W2)
bool mlt_conv_cfg(int cin, int cout, int stride, int exact, ConvCfg *out);
hipError_t mlt_launch_stem5(const Stem5Args &a, int nsplit, int grid_x, int lds, hipStream_t st);  // nsplit as mlt_launch_conv
hipError_t mlt_launch_block32(const Block32Args &a, bool w2, int grid_x, hipStream_t st);     // w2: hi+lo weights (8 x 32 tiles)
hipError_t mlt_launch_stem_block(const StemBlockArgs &a, bool w2, int grid_x, hipStream_t st);
hipError_t mlt_launch_layer0_stream(const Layer0Args &a, bool fuse5, bool mfma32, int grid_x, hipStream_t st);
hipError_t mlt_launch_layer1_stream(const Layer1Args &a, bool mfma32, int grid_x, hipStream_t st);
hipError_t mlt_launch_heads(const HeadArgs &a, int n, hipStream_t st);
