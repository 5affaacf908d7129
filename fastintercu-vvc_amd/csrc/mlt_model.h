// mlt_model.h -- host-side model description (folded + packed weights).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace mlt {

// fp16 weights in MFMA A-fragment order:
//   [cout tile (cout/ct)][cin chunk (cin/kc)][tap (taps, +1 if has_sc)][k-step (kc/16)][32-channel block (ct/32)][lane 64][8 halves]
// lane l = 32*hh + r holds W'[cout = tile*ct + blk*32 + r][cin = chunk*kc + 16*ks + 8*hh + j][tap], j = 0..7.
// With has_sc the block's 1x1 stride-2 projection shortcut is stored as one extra "tap" after the 9 conv taps.
struct PackedConv {
  int cin = 0, cout = 0, taps = 0, stride = 1;
  int kc = 0, ct = 0, mt = 0, gt = 0;  // from mlt_conv_cfg(): cin chunk, couts / pixels per tile, taps per step
  int gt_w2 = 0;                       // exact packing: taps per step when the layer runs in the hi+lo-weights tier (its own tiling where the cin chunk allows)
  int lat = 0;                         // 1: a small-batch variant exists (32 couts x 128 pixels per workgroup, same packed weights)
  int dma = 0, mt_dma = 0;             // LDS-DMA staging variant (0 none, 1 resident weights, 2 weight ring) and its pixels per workgroup
  bool has_sc = false;
  bool exact = false;          // w holds a hi plane followed by a lo plane (fp16 pair per weight)
  bool lo8 = false;            // w2 layers with 64-channel chunks (stride 1, >= 64 channels): the lo plane is FP8 (OCP e4m3), packed as the A operand of
                               // v_mfma_scale_f32_32x32x64_f8f6f4 -- [cout tile][chunk][tap][32-ch block][16-byte half][lane 64][16 bytes], byte j of lane 32h + r =
                               // lo[cout r][cin = 32 h + j of the chunk] * 2^lo8_exp -- half the bytes of an fp16 lo plane, and the lo
                               // product of a (tap, chunk) is ONE MFMA at twice the fp16 rate (a 2^-11-relative correction tolerates a 3-bit mantissa)
  int lo8_exp = 0;             // the MFMA's block scale undoes it (E8M0 byte 127 - lo8_exp)
  bool w2 = false;             // hi+lo-WEIGHTS tier on the FAST tiling (kc / ct / gt of the single-pass kernels): two planes like `exact`, read by the
                               // W2 forms of the fused kernels (chain_kernel, stem_block_kernel, block32_kernel) and by conv_mfma_kernel<NSPLIT = 3>
  bool xl = false;             // exact-lite (round 5, MLT_MODEL_XLITE; conv_mfma_kernel NSPLIT == 6): the exact packing with the lo plane replaced, byte for byte, by
                               // the FP8 A operand of the cross-term MFMA -- per (cout tile, chunk, tap, k-step slot g, 32-cout block) 1 KiB = [lane 64][16 bytes]:
                               // slot 0, lane 32 h + r = e4m3(Wl[cout r][cin 16 h + b] * 2^xl_ewl), slot 1 = e4m3(Wh[...] * 2^xl_ewh), b = 0..15 (the hardware's
                               // K block 0 is bytes 0-15 of both lane halves, block 1 bytes 16-31: scripts/probes/f8_mfma_scale_probe2.hip)
  int xl_ewl = 0, xl_ewh = 0;
  size_t plane_halves = 0;     // halves per plane
  float acc_scale = 1.f;       // stored weights = folded weights * 2^s; kernels multiply accumulators by 2^-s
                               // (keeps small weights and their lo parts out of fp16's subnormal range)
  std::vector<uint16_t> w;
  std::vector<float> bias;     // folded BN bias (zeros for the stem)
  std::vector<float> bias_sc;  // folded BN bias of the shortcut (has_sc)
  void *d_w = nullptr;         // device copies
  float *d_bias = nullptr, *d_bias_sc = nullptr;
};

struct Block {
  PackedConv conv1, conv2;  // conv1 carries the projection shortcut when the block has one (arch:44-50)
  PackedConv conv1_s2c;     // fast arithmetic, stages with a whole-stage kernel (chain_kernel S2): conv1 + shortcut packed again
                            // for 16-channel chunks and 128-cout tiles (empty otherwise)
};

struct Head {
  int classes = 0, c = 0;
  std::vector<float> w, b;  // w [classes][c+2], columns [features..., poc, qp] (arch:284)
  float *d_w = nullptr, *d_b = nullptr;
};

struct Model {
  int arch = 0, n_stages = 0, n_heads = 0, n_logits = 0;
  bool exact = false;
  bool w2 = false;          // fast tiling, two weight planes (MLT_MODEL_W2)
  bool xl = false;          // exact-lite (MLT_MODEL_XLITE): `exact` is set as well (two activation planes, the exact tiling); the 3x3 / 1x1 convs carry FP8 lo planes
  int planes[5] = {0, 0, 0, 0, 0};
  PackedConv stem;
  PackedConv stem_b;        // fast arithmetic: the composed first layer packed again for stem_block_kernel (mlt_model.cpp: pack_stem_b)
  Block blocks[5][2];
  Head heads[4];
  bool on_device = false;
  int rounding = 0;         // MLT_MODEL_FAST: the rounding realisation of the 3x3 layers (build_model)
};

enum { MLT_MODEL_FAST = 0, MLT_MODEL_EXACT = 1, MLT_MODEL_W2 = 2, MLT_MODEL_XLITE = 3 };  // tap-diffused single fp16 plane / (hi, lo) planes on the exact tiling / (hi, lo) planes on the fast tiling
// rounding (MLT_MODEL_FAST only, round 4): which realisation of the tap-diffused rounding the 3x3 layers get -- 0: raster tap order, error
// reset per (cout, cin) pair (rounds 1-3); 1: raster, error carried across cin; 2: reversed taps; 3: column-major taps; 4: spiral from the
// centre; 5: serpentine, carried across cin.  Every one keeps each weight within one ulp and cancels the rounding error along spatially
// adjacent taps; they differ by which weights take which error, i.e. they are different draws of the same error distribution, and the
// load-time calibration may pick the draw that suits a weight set (mlt_api.cpp).
enum { MLT_N_ROUNDINGS = 6 };
bool build_model(const void *blob, size_t bytes, int mode, int size, Model &m, std::string &err, int rounding = 0);  // size: CU size the model will serve
uint16_t f32_to_f16(float f);
float f16_to_f32(uint16_t h);

}  // namespace mlt
