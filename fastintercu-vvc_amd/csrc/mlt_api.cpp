// mlt_api.cpp -- C ABI (include/mltcnn.h) and host runtime of the MI355X MLT-CNN split predictor.
//
// Replaces the inline block EncCu.cpp:799-930 of the reference encoder: what was
//   xMalloc + copy loops (:810-830), cv::absdiff/convertTo/clip (:832-867), from_blob/cat/.to(kCUDA) (:869-887),
//   torch::jit::load PER CALL (:894-900), forward (:909), .cpu()/argmax (:920-921)
// becomes: one init (weights folded, packed, resident in HBM), then per call a strided H2D copy of the two
// Pel planes, a fixed chain of HIP kernel launches on one stream, and a few bytes D2H.
// There is NO CPU fallback: without a usable HIP device mlt_init fails with MLT_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <tuple>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mltcnn.h"
#include "mlt_kernels.h"
#include "mlt_model.h"
#include "mlt_tier_search.h"

namespace {

std::string g_init_error;
std::mutex g_mutex;

// Every environment switch of the library except MLT_CALIB_VERBOSE is a tuning / test knob (A/B runs, sweeps, fallback paths): they are
// honoured only when MLT_TUNING=1 is set as well, so that an encoder process cannot change kernel selection, chunking or wait modes by
// an accident of its environment.
const char *tuning_env(const char *name) {
  const char *e = std::getenv("MLT_TUNING");
  return (e && e[0] == '1') ? std::getenv(name) : nullptr;
}

// Every tuning switch of the dispatcher and of the load path, parsed ONCE on first use (round 6, VERDICT r5 item 7: round 5 read them through some thirty
// function-local statics scattered over the run_* functions).  All of them need MLT_TUNING=1 (tuning_env).
struct Tuning {
  const char *debug_dump_dir;      // MLT_DEBUG_DUMP_DIR: after every kernel, synchronise and write its output tensor there (bring-up only)
  int xl_kill, xl_kill_layer;      // MLT_XL_KILL / MLT_XL_KILL_LAYER: exact-lite bring-up, silence a K block (of one layer: cin * 1000 + cout)
  long lat_pixels;                 // MLT_LAT_PIXELS: launches of at most this many output pixels take the latency variants / per-conv forms (16384; 0 disables)
  bool no_exact_lat;               // MLT_NO_EXACT_LAT
  bool exact_patch_big;            // (MLT_EXACT_PATCH_64K clears it) the exact arithmetic's patch planes may use what the weight ring leaves of the LDS
  int wg_cap, wg_cap0, wg_cap1, wg_cap2;   // MLT_WG_CAP / _CAP0 / _CAP1 / _CAP2: persistent workgroups per launch (256 = one per CU)
  bool l0_mfma32, l1_mfma32;       // MLT_L0_MFMA32: layer0_stream_kernel on round 5's 32x32x16 MFMAs; layer1_stream_kernel runs them unless MLT_L1_MFMA16 (same bits either way)
  bool no_chain, no_chain_s2, no_c16, chain64, no_block_fusion, no_l0_s5;   // MLT_NO_CHAIN, _CHAIN_S2, _C16, _CHAIN64, _BLOCK_FUSION, _L0_S5: knock-outs of the fused forms
  int l0_stream_min, l1_stream_min;        // MLT_L0_STREAM_MIN / MLT_L1_STREAM_MIN (128; MLT_NO_L0_STREAM / MLT_NO_L1_STREAM: 0 = never)
  int guard_wait_mode;             // 0 sleep-then-poll, 1 MLT_GUARD_SPIN_WAIT, 2 MLT_GUARD_BLOCKING_WAIT
  bool no_small_mix, no_mag_guard, no_graph;   // MLT_NO_SMALL_MIX, MLT_NO_MAG_GUARD, MLT_NO_GRAPH (also set by MLT_DEBUG_DUMP_DIR)
};
const Tuning &tuning() {
  static const Tuning t = [] {
    auto num = [](const char *name, long dflt) { const char *e = tuning_env(name); return e ? std::atol(e) : dflt; };
    auto on = [](const char *name) { return tuning_env(name) != nullptr; };
    auto cap = [&](const char *name) { const long v = num(name, 0); return (int)(v > 0 ? v : 256); };
    Tuning u{};
    u.debug_dump_dir = tuning_env("MLT_DEBUG_DUMP_DIR");
    u.xl_kill = (int)num("MLT_XL_KILL", 0); u.xl_kill_layer = (int)num("MLT_XL_KILL_LAYER", -1);
    u.lat_pixels = num("MLT_LAT_PIXELS", 16384L);
    u.no_exact_lat = on("MLT_NO_EXACT_LAT"); u.exact_patch_big = !on("MLT_EXACT_PATCH_64K");
    u.wg_cap = cap("MLT_WG_CAP"); u.wg_cap0 = cap("MLT_WG_CAP0"); u.wg_cap1 = cap("MLT_WG_CAP1"); u.wg_cap2 = cap("MLT_WG_CAP2");
    u.l0_mfma32 = on("MLT_L0_MFMA32"); u.l1_mfma32 = !on("MLT_L1_MFMA16");
    u.no_block_fusion = on("MLT_NO_BLOCK_FUSION");
    u.no_chain = on("MLT_NO_CHAIN") || u.no_block_fusion; u.no_chain_s2 = on("MLT_NO_CHAIN_S2"); u.no_c16 = on("MLT_NO_C16"); u.chain64 = !on("MLT_NO_CHAIN64");
    u.no_l0_s5 = on("MLT_NO_L0_S5");
    u.l0_stream_min = on("MLT_NO_L0_STREAM") ? 0 : (int)num("MLT_L0_STREAM_MIN", 128);
    u.l1_stream_min = on("MLT_NO_L1_STREAM") ? 0 : (int)num("MLT_L1_STREAM_MIN", 128);
    u.guard_wait_mode = on("MLT_GUARD_SPIN_WAIT") ? 1 : on("MLT_GUARD_BLOCKING_WAIT") ? 2 : 0;
    u.no_small_mix = on("MLT_NO_SMALL_MIX"); u.no_mag_guard = on("MLT_NO_MAG_GUARD");
    u.no_graph = on("MLT_NO_GRAPH") || u.debug_dump_dir != nullptr;
    return u;
  }();
  return t;
}

int size_index(int size) {
  switch (size) {
    case 128: return 0;
    case 64: return 1;
    case 32: return 2;
    case 16: return 3;
    default: return -1;
  }
}
int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// One launch of a size's plan (built by plan_network, walked by run_network -- both further down): the kind says which buffers it reads and writes.
struct PlanStep {
  enum Kind : uint8_t {
    FLAT_STAT,      // the flat-content statistic as a launch of its own (the first kernel does not read the raw planes as aligned quads)
    LAYER0_STREAM,  // all of layer0 (front: + layer1.0.conv1 + shortcut) from the raw planes          -> out[0] | t = pool0, sc = pool1 (chunk-major)
    STEM_BLOCK,     // layer0.0 from the raw planes                                                      -> b0 = pool2
    STEM5,          // composed first layer: t, sc of layer0.0                                           -> pool0, pool1
    CONV_S2,        // stride-2 conv + shortcut that open stage s: stage input                           -> t = pool0, sc = pool1
    CHAIN,          // rest of stage s (inside_s2: all of it, from the stage input) in one launch        -> out[s] (+ GAP)
    LAYER1_STREAM,  // the three stride-1 convs of the 64-channel stage, streaming form                   -> out[1] (+ GAP)
    CONV_B0C2,      // block 0, conv2 + shortcut + ReLU: pool0, pool1                                     -> b0 = pool2
    BLOCK32,        // layer0.1 in one kernel: pool2                                                      -> out[0]
    CONV_B1C1,      // block 1, conv1: pool2                                                              -> pool3
    CONV_B1C2,      // block 1, conv2 + b0 + ReLU: pool3, pool2                                           -> out[s] (+ GAP)
    HEADS
  } kind;
  int8_t s = 0;             // stage
  bool front = false;       // LAYER0_STREAM: layer1.0.conv1 + shortcut ride along
  bool inside_s2 = false;   // CHAIN: the stage's stride-2 conv + shortcut run inside the launch
  bool x_c16 = false;       // CHAIN (inside_s2): the stage input is chunk-major
  bool sc_c16 = false;      // CONV_S2 writes / CHAIN, LAYER1_STREAM read the shortcut chunk-major
  bool y_c16 = false;       // the stage's output is written chunk-major (its consumer is a whole-stage launch)
};
using NetPlan = std::vector<PlanStep>;
using PlanKey = std::tuple<const void *, const void *, const void *, unsigned, unsigned, unsigned>;   // models (main, hi+lo weights, exact), their unit masks, batch class

struct SizeState {
  bool enabled = false, loaded = false;
  std::map<PlanKey, NetPlan> plans;   // launch plans of the loaded tier, by batch class (filled on first use; cleared by every (re)load)
  bool exact = false;          // arithmetic `model` runs (after calibration)
  bool w2 = false;             // middle tier: the main path runs `model_w2` (hi+lo weights on single fp16 activations, fused kernels); guards as for fast
  unsigned w2_mask = 0;        // ... in the stages whose bit is set (bit s = layer s; all four: the whole network); the other stages stay on the
                               // single-pass kernels of `model` ("mixed" tiers: the calibration picks the CHEAPEST set of stages that meets the contract)
  unsigned w2_units = 0;       // ... at the granularity the kernels allow: bit 2 s = the first launch unit of layer s (layer0.0 / the stride-2 conv + shortcut),
                               // bit 2 s + 1 = the second (layer0.1 / the three stride-1 convs); w2_mask = the stages with at least one unit set
  unsigned x_mask = 0;         // round 4, the tier below exact: the stages of this mask run the EXACT arithmetic (model_exact's per-conv kernels, a lo
                               // plane behind their activations), the other stages hi+lo weights (w2_mask = the complement) -- for weight sets whose
                               // ACTIVATION rounding, spread evenly over all 18 rounding sites, misses the contract by a few per cent
  unsigned x_units = 0;        // ... at launch-unit granularity (bit 2 s + u as in w2_units); x_mask = the stages with at least one exact unit
  bool lite = false;           // round 5: the calibration found no fp16 tier within the contract but the exact-lite arithmetic is: `model` is the exact-lite
                               // model (two activation planes, FP8 cross terms; mlt_model.h: xl), the flat guard is off (its error is 1/20 of the single
                               // pass's), the decision guard -- if configured -- still re-evaluates near-ties with `model_exact`
  bool cfg_flat_guard = false; // flat guard as configured (flags); flat_guard is what the loaded tier uses
  int flat_div = 8;            // the flat guard re-evaluates a CU when >= 1 / flat_div of its quads are EXACTLY flat.  8 for the fp16 tiers admitted by the plain rule; 16
                               // for a tier behind the magnitude guard (CalibSession::price_guarded) and for the exact-lite
                               // tier (round 6: its FP8 cross terms quantise a constant area's activations coherently -- 3-bit mantissas -- and a weight set with
                               // large logits turns a 10-12 % constant band into |dlogit| 1.0-1.5e-3: 4 of 331,776 probed logits of the second trained family,
                               // profiles/r06b_tail_probe_trained.txt; round 5 had switched the guard OFF for this tier on the strength of the seeded sets)
  float guard_margin = 3e-3f;  // decision guard's threshold for THIS size's tier: the context's (3 x tolerance, or as configured) -- except the exact-lite tier, whose
                               // largest calibration error is ~ 1/6 of the tolerance: 3 x 1.7 x calib_max there (1.7: the tail probes' largest error over the
                               // calibration set's), at least 1e-4, at most the context's
  bool want_exact = false;     // configured arithmetic (flags)
  bool flat_guard = false, margin_guard = false, calibrate = false;
  bool small_mix = false;      // 64 / 32 / 16 (round 4): configured exact, but the load-time calibration may keep the first stages -- the large maps, where the
                               // time is -- on the single-pass kernels (x_mask = the remaining stages); exact when no prefix meets the contract
  bool calibrated = false;
  // Round 6, the MAGNITUDE guard.  The fp16 pipeline's error is RELATIVE: it scales with the size of the feature-driven part of the logits,
  // M = max over logits of sum_k |w_ck gap_k| (heads_kernel).  A trained-like weight set amplifies what it was trained to see: on content far
  // outside its training range (a residual plane of hundreds of ten-bit steps: the synthetic calibration classes "uniform", "constant org /
  // pred", the bands) M is 20-80 x what texture or natural scenes give (tools/attribute_error.py, profiles/r06_attribution_*.txt) and so is
  // the absolute error -- the heavy tail that kept every fp16 tier of that family out.  mag_thr > 0: CUs whose M exceeds it
  // are re-evaluated with the exact arithmetic like the flat-content guard's; the threshold is where the tier's worst relative error
  // measured at load time (calib_rel, over EVERY calibration CU) reaches max_frac x tolerance.  0: off (every seeded weight set: their
  // error does not follow M, they are admitted -- or not -- by the plain rule as before).
  float mag_thr = 0.f, calib_rel = 0.f, mag_flag = 0.f;
  int mag_kind = 0;            // 0: no magnitude guard; 2: the tier was ADMITTED behind it (threshold from the admission rule, above); 1: RANGE guard -- a tier the plain rule
                               // admitted is still only validated on the magnitudes its calibration CUs had, and its error grows linearly with M: CUs beyond
                               // kMagRange x the largest calibrated magnitude are re-evaluated exactly (1.5 x the 0.65-of-tolerance the largest calibration error
                               // may reach = the tolerance).  Costs nothing on content inside the calibrated range; it is what stands between a tier and
                               // content it has never been priced on (scripts/r06_lite_range_probe.py: the exact-lite arithmetic forced onto activations
                               // beyond its e4m3 range is off by 2e-2)
  bool cfg_mag_guard = true;   // (MLT_FLAG_NO_MAGNITUDE_GUARD clears it)
  float calib_rms = 0.f, calib_max = 0.f;
  int calib_cus = 0, calib_caller_cus = 0;   // CUs the last calibration priced (after dropping those the flat guard re-evaluates anyway) / of them the caller's (mlt_calibrate)
  std::vector<char> blob;      // host copy of the MLTW blob the size was loaded from (mlt_calibrate re-packs from it)
  uint64_t reruns = 0;         // CUs re-evaluated by the guards
  // run_checked's sleep: measured duration (enqueue -> flagged-CU count on the host) of a guarded chunk, per power-of-two bucket of the chunk
  // size (a 4-CU tail costs ~50 us per CU, a 4096-CU chunk ~1.2: ONE per-CU figure for all sizes made a tail's measurement inflate the
  // next full chunk's sleep 20-fold); guard_n[b] = the chunk size guard_us[b] was measured on; 0: unknown -> poll
  double guard_us[16] = {0};
  int guard_n[16] = {0};
  int size = 0, head_index = 0;
  mlt::Model model;
  mlt::Model model_exact;      // fast sizes: exact-arithmetic copy the guards re-evaluate flagged CUs with
  mlt::Model model_w2;         // hi+lo-weights copy on the fast tiling (MLT_MODEL_W2); built only when the single-pass calibration fails
  mlt::Model model_xl;         // exact-lite copy (MLT_MODEL_XLITE): exists only while the calibration prices it; the tier's model moves into `model`
  bool guards() const { return !exact && (flat_guard || margin_guard || mag_thr > 0.f) && model_exact.on_device; }
};

// device-side guard state of one in-flight batch (mlt_kernels.hip: flat_stat / guard_select kernels)
struct GuardSlot {
  int32_t *d_flat = nullptr, *d_idx = nullptr, *d_count = nullptr;
  float *d_lg = nullptr;       // logits for the margin test when the caller wants none
  float *d_mag = nullptr;      // per-CU logit magnitude (HeadArgs.mag) for the magnitude guard
  int *phase = nullptr;        // batches with the selection inside the heads kernel: d_count[*phase] is the launch's running count (zero on entry), d_count[*phase ^ 1] the
                               // one it zeroes for the NEXT launch of the slot; toggled per launch (run_guarded_async).  NULL: no such pair (the slot serves one-CU launches)
  int32_t *h_count = nullptr;  // pinned
  bool single = false;         // mlt_predict's slot: one CU, d_flat zero on entry and cleared by the heads kernel (consume-and-clear), the
                               // selection rides on the heads kernel, and the caller's own result copy brings the count back
};
// guard selection fused into the heads kernel of a single-CU launch (HeadArgs.g_*)
struct GuardTail { int32_t *count, *idx, *flat; int flat_thr, near_thr; float margin, mag_thr; int32_t *next; };  // next != NULL: batches (HeadArgs.g_next)

// mlt_predict (one CU per call, the encoder's use): pinned host staging, one H2D, the kernel chain replayed from a
// hipGraph captured once per CU size, one D2H.
struct SingleCu {
  char *h_stage = nullptr, *d_stage = nullptr;  // [org plane][pred plane][poc, qp, split, pad, logits...]
  size_t plane = 0;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  uint64_t ws_gen_at_capture = 0;       // the graph bakes in workspace pointers: valid only for the workspace allocation it was captured on
  hipStream_t stream_at_capture = nullptr;
};

// mlt_submit / mlt_flush / mlt_wait: two generations of up to MLT_DEFER_CAP CUs per size (one accumulating, one in
// flight or finished).  Generation g uses buffer set g & 1; ticket = g * MLT_DEFER_CAP + slot.
struct Deferred {
  char *h_in = nullptr, *d_in = nullptr;    // [2 sets][org planes CAP][pred planes CAP][poc CAP][qp CAP]
  char *h_out = nullptr, *d_out = nullptr;  // [2 sets][split CAP][logits CAP * nl]
  size_t in_set = 0, out_set = 0, plane = 0;
  uint64_t gen = 0;                         // generation being filled
  int n = 0;                                // CUs submitted into it
  int n_launched[2] = {0, 0};               // CUs of the generation occupying each set (0: never launched)
  uint64_t gen_of_set[2] = {~0ull, ~0ull};
  hipEvent_t done[2] = {nullptr, nullptr};
  bool guard_pending[2] = {false, false};   // set's batch ran with guards and its flagged CUs have not been re-evaluated yet
  int phase[2] = {0, 0};                    // which of the set's two selection counters (count[0 .. 1]) its next launch counts on (GuardSlot.phase)
};

struct ProfAcc { uint32_t launches = 0; double flops = 0, bytes = 0; std::vector<std::pair<hipEvent_t, hipEvent_t>> ev; };

}  // namespace

struct mlt_ctx {
  // multi-device context (mlt_config.n_devices > 1): this object is the context of devices[0]; peers[i] is the full context of
  // devices[i + 1].  Host-pointer entry points shard over all of them (contiguous ranges, one host thread per peer), mlt_submit deals
  // CUs round-robin (the ticket's top byte is the device index); device-pointer calls, streams and profiles address ONE device
  // (mlt_device_ctx).  A peer never has peers of its own.
  std::vector<mlt_ctx *> peers;
  unsigned rr = 0;  // next device of mlt_submit
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  SizeState sz[4];
  float guard_margin = 3e-3f;  // decision guard: 3 x tolerance unless configured (mlt_init)
  bool guard_margin_configured = false;  // mlt_config.guard_margin > 0: used as is for every tier
  int max_batch = 4096, chunk = 4096;  // CUs per pass; MLT_CHUNK overrides (workspace ~1.5 MiB per CU at S = 128)
  char *ws = nullptr;
  size_t ws_bytes = 0;
  uint64_t ws_gen = 1;  // bumped by every (re)allocation / release of ws: a captured graph is replayed only on the generation it was captured on
                        // (an equal ADDRESS proves nothing once the workspace can shrink: a later, smaller allocation may land on it)
  char *zero_page = nullptr;  // 64 KiB of zeros: padding source of the LDS-DMA patch staging
  // mlt_predict_batch: second stream + events for the H2D / compute overlap, CUs per staged sub-chunk (MLT_STAGE_CHUNK)
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_h2d[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
  int stage_chunk = 512;  // measured on 4096 x 128x128 from pinned memory: 512 -> 526 k, 1024 -> 498 k, 2048 -> 426 k CU/s
  char *h_res = nullptr;  // pinned result staging (split + logits) for the two sets
  size_t h_res_bytes = 0;
  SingleCu single[4];
  Deferred deferred[4];
  // staging for the host-pointer entry points
  char *stage = nullptr;
  size_t stage_bytes = 0;
  // parity guards: selection buffers for two in-flight batches, gather staging for the flagged CUs
  float tolerance = 1e-3f;
  bool xlite = false;  // MLT_FLAG_EXACT_LITE: sizes configured exact run the exact-lite arithmetic (FP8 cross terms, mlt_model.h: xl)
  char *guard_dev = nullptr;
  size_t guard_slot_bytes = 0;
  int guard_cap_n = 0, guard_cap_nl = 0;
  int32_t *guard_host = nullptr;  // pinned: two counters
  int guard_phase[2] = {0, 0};    // which counter of a slot's pair the next batch launch counts on (GuardSlot.phase)
  hipEvent_t ev_guard = nullptr;  // "count of flagged CUs has landed" (device-pointer entry)
  char *gstage = nullptr;
  size_t gstage_bytes = 0;
  std::string err;
  bool profile = false;
  bool guard_select_kernel = false;  // MLT_GUARD_SELECT_KERNEL (read at mlt_init, like MLT_CHUNK): the guards' selection of a BATCH as a launch of its own (round 5's form,
                                     // ascending list) instead of a tail of the heads kernel -- kept for the A/B and the bit-identity test
  bool lds_oob_zero = false;  // DS reads beyond the LDS allocation return zeros on this device (probed at init): chain kernels without zero masks
  std::map<std::string, ProfAcc> prof;
  std::vector<std::string> prof_order;
  // Round 6: != NULL = PLAN mode -- the dispatcher (run_network and the run_* functions under it) runs its decisions as usual but
  // RECORDS every launch (name + variant) here instead of enqueuing it, touches no device and allocates nothing: what a batch of n CUs of a tier
  // would launch, observable and unit-tested on the CPU (mlt_plan_describe, tests/test_launch_plan_cpu.py).
  std::vector<std::string> *plan = nullptr;
  bool plan_detail = false;   // ... with the buffers of every launch (offsets into the workspace, which plan mode carves from address 0): the hand-offs between launches
};

namespace {

#define HIP_TRY(ctx, expr)                                                                             \
  do {                                                                                                 \
    hipError_t e__ = (expr);                                                                           \
    if (e__ != hipSuccess) {                                                                           \
      (ctx)->err = std::string(#expr) + ": " + hipGetErrorString(e__);                                 \
      return MLT_ERR_HIP;                                                                              \
    }                                                                                                  \
  } while (0)

// a kernel launch of the dispatcher: skipped in plan mode (the launch was recorded by Launch::prof_begin)
#define LAUNCH_TRY(ctx, expr) do { if (!(ctx)->plan) HIP_TRY(ctx, expr); } while (0)

int upload_model(mlt_ctx *ctx, mlt::Model &m) {
  auto up = [&](mlt::PackedConv &pc) -> int {
    if (pc.w.empty()) return MLT_OK;  // layer0.0.conv1: folded into the composed first layer
    HIP_TRY(ctx, hipMalloc(&pc.d_w, pc.w.size() * 2));
    HIP_TRY(ctx, hipMemcpy(pc.d_w, pc.w.data(), pc.w.size() * 2, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMalloc((void **)&pc.d_bias, pc.bias.size() * 4));
    HIP_TRY(ctx, hipMemcpy(pc.d_bias, pc.bias.data(), pc.bias.size() * 4, hipMemcpyHostToDevice));
    if (pc.has_sc) {
      HIP_TRY(ctx, hipMalloc((void **)&pc.d_bias_sc, pc.bias_sc.size() * 4));
      HIP_TRY(ctx, hipMemcpy(pc.d_bias_sc, pc.bias_sc.data(), pc.bias_sc.size() * 4, hipMemcpyHostToDevice));
    }
    return MLT_OK;
  };
  int rc;
  if ((rc = up(m.stem))) return rc;
  if ((rc = up(m.stem_b))) return rc;
  for (int s = 0; s < m.n_stages; ++s)
    for (int b = 0; b < 2; ++b) {
      if ((rc = up(m.blocks[s][b].conv1))) return rc;
      if ((rc = up(m.blocks[s][b].conv2))) return rc;
      if ((rc = up(m.blocks[s][b].conv1_s2c))) return rc;
    }
  for (int h = 0; h < m.n_heads; ++h) {
    mlt::Head &H = m.heads[h];
    HIP_TRY(ctx, hipMalloc((void **)&H.d_w, H.w.size() * 4));
    HIP_TRY(ctx, hipMemcpy(H.d_w, H.w.data(), H.w.size() * 4, hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMalloc((void **)&H.d_b, H.b.size() * 4));
    HIP_TRY(ctx, hipMemcpy(H.d_b, H.b.data(), H.b.size() * 4, hipMemcpyHostToDevice));
  }
  m.on_device = true;
  return MLT_OK;
}

void free_model(mlt::Model &m) {
  auto fr = [](mlt::PackedConv &pc) {
    if (pc.d_w) (void)hipFree(pc.d_w);
    if (pc.d_bias) (void)hipFree(pc.d_bias);
    if (pc.d_bias_sc) (void)hipFree(pc.d_bias_sc);
    pc.d_w = nullptr; pc.d_bias = pc.d_bias_sc = nullptr;
  };
  fr(m.stem);
  fr(m.stem_b);
  for (int s = 0; s < 5; ++s)
    for (int b = 0; b < 2; ++b) { fr(m.blocks[s][b].conv1); fr(m.blocks[s][b].conv2); fr(m.blocks[s][b].conv1_s2c); }
  for (int h = 0; h < 4; ++h) { if (m.heads[h].d_w) (void)hipFree(m.heads[h].d_w); if (m.heads[h].d_b) (void)hipFree(m.heads[h].d_b); m.heads[h].d_w = m.heads[h].d_b = nullptr; }
  m.on_device = false;
}

int gap_slots(int hw) { return hw >= 32 ? hw / 32 : 1; }

// activation workspace (bytes per CU) for size S: 4 scratch maps of stage-0 size, one output per stage,
// fp32 GAP partial sums per head.  The stem activation is never materialised (fused into layer0.0.conv1).
size_t ws_per_cu(const mlt::Model &m, int S, bool some_exact = false) {
  const int h0 = S / 2 > 0 ? S / 2 : 1;
  const int planes = (m.exact || some_exact) ? 2 : 1;  // exact mode keeps a lo plane behind every activation
  size_t b = 4 * ((size_t)h0 * h0 * 32 * 2 * planes + 256);
  int h = S;
  for (int s = 0; s < m.n_stages; ++s) {
    h = h / 2 > 0 ? h / 2 : 1;
    b += (size_t)h * h * m.planes[s] * 2 * planes + 256;
    if (s >= 1) b += (size_t)gap_slots(h * h) * m.planes[s] * 4 + 256;
  }
  return b + 4096;
}

int ensure_ws(mlt_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return MLT_OK;
  if (ctx->ws) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->ws); ctx->ws = nullptr; ctx->ws_bytes = 0; }
  ++ctx->ws_gen;
  HIP_TRY(ctx, hipMalloc((void **)&ctx->ws, bytes));
  ctx->ws_bytes = bytes;
  return MLT_OK;
}
void release_ws(mlt_ctx *ctx) {  // (the caller has synchronised the stream)
  if (ctx->ws) { (void)hipFree(ctx->ws); ctx->ws = nullptr; ctx->ws_bytes = 0; }
  ++ctx->ws_gen;
}

// MLT_DEBUG_DUMP_DIR=<dir>: after every kernel, synchronise and write the output tensor to <dir>/<seq>_<name>.bin
// (bring-up aid only; never set in production or in timed runs).
int debug_dump(mlt_ctx *ctx, const char *name, const void *dptr, size_t bytes) {
  const char *dir = tuning().debug_dump_dir;
  static int seq = 0;
  if (!dir) return MLT_OK;
  std::vector<char> host(bytes);
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  HIP_TRY(ctx, hipMemcpy(host.data(), dptr, bytes, hipMemcpyDeviceToHost));
  char path[1200];
  std::snprintf(path, sizeof path, "%s/%02d_%s.bin", dir, seq++, name);
  if (FILE *f = std::fopen(path, "wb")) { std::fwrite(host.data(), 1, bytes, f); std::fclose(f); }
  return MLT_OK;
}

struct Launch {
  mlt_ctx *ctx;
  int prof_begin(const std::string &name, double flops, double bytes, hipEvent_t &e0, hipEvent_t &e1, const char *variant = nullptr) {
    if (ctx->plan) { ctx->plan->push_back(variant && variant[0] ? name + " [" + variant + "]" : name); return MLT_OK; }
    if (!ctx->profile) return MLT_OK;
    auto it = ctx->prof.find(name);
    if (it == ctx->prof.end()) { ctx->prof_order.push_back(name); it = ctx->prof.emplace(name, ProfAcc()).first; }
    it->second.launches++; it->second.flops += flops; it->second.bytes += bytes;
    HIP_TRY(ctx, hipEventCreate(&e0));
    HIP_TRY(ctx, hipEventCreate(&e1));
    it->second.ev.emplace_back(e0, e1);
    HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
    return MLT_OK;
  }
  int prof_end(hipEvent_t e1) {
    if (!ctx->profile) return MLT_OK;
    HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
    return MLT_OK;
  }
};

// plan mode with detail: append the launch's buffers to its record ("{x=ws+0x..., y=...}"; NULL pointers are left out, workspace pointers print as offsets)
void plan_note(mlt_ctx *ctx, std::initializer_list<std::pair<const char *, const void *>> ptrs, std::initializer_list<std::pair<const char *, long>> vals = {}) {
  if (!ctx->plan || !ctx->plan_detail || ctx->plan->empty()) return;
  std::string t = " {";
  char b[64];
  bool first = true;
  for (const auto &q : ptrs) {
    if (!q.second) continue;
    std::snprintf(b, sizeof b, "%s%s=0x%llx", first ? "" : " ", q.first, (unsigned long long)(uintptr_t)q.second);
    t += b; first = false;
  }
  for (const auto &q : vals) {
    std::snprintf(b, sizeof b, "%s%s=%ld", first ? "" : " ", q.first, q.second);
    t += b; first = false;
  }
  ctx->plan->back() += t + "}";
}

struct ConvIO {
  const void *x = nullptr;    // input activation (ignored when raw planes are given)
  void *y = nullptr;          // main output (may be NULL when only the GAP sums are needed)
  void *y_sc = nullptr;       // shortcut output (conv carries a shortcut)
  const void *res = nullptr;  // residual added before the ReLU
  float *gap = nullptr;       // GAP partial sums
  bool relu = false;
  bool y_c16 = false;         // main output chunk-major (ConvArgs.y_c16)
  bool ysc_c16 = false;       // shortcut output chunk-major (ConvArgs.ysc_c16)
  size_t x_lo = 0, y_lo = 0, res_lo = 0, ysc_lo = 0;  // exact mode: byte offsets hi plane -> lo plane
};

int run_conv(mlt_ctx *ctx, const mlt::PackedConv &pc, int n, int hin, const ConvIO &io, int *hout_out) {
  const int hout = hin / pc.stride > 0 ? hin / pc.stride : 1;
  *hout_out = hout;
  ConvArgs a{};
  a.x = io.x; a.y = io.y; a.w = pc.d_w; a.bias = pc.d_bias; a.res = io.res; a.n = n; a.relu = io.relu ? 1 : 0;
  a.y_sc = io.y_sc; a.bias_sc = pc.d_bias_sc; a.acc_scale = pc.acc_scale; a.y_c16 = io.y_c16 ? 1 : 0; a.ysc_c16 = io.ysc_c16 ? 1 : 0;
  a.hin_l = ilog2(hin); a.hout_l = ilog2(hout);
  a.x_lo_off = io.x_lo; a.y_lo_off = io.y_lo; a.res_lo_off = io.res_lo; a.ysc_lo_off = io.ysc_lo; a.w_lo_off = pc.plane_halves * 2;
  a.lo8_scale = 0x01010101 * ((127 - pc.lo8_exp) & 0xFF);
  a.xl_sa0 = 0x01010101 * ((127 - pc.xl_ewl) & 0xFF); a.xl_sa1 = 0x01010101 * ((127 - pc.xl_ewh) & 0xFF); a.xl_sb1 = 0x01010101 * (127 - 12);  // (12 = MLT_XL_LO_EXP, mlt_kernels.hip)
  { const int kill = tuning().xl_kill, only = tuning().xl_kill_layer;  // bring-up: E8M0 byte 0 = 2^-127 silences a K block (of one layer: cin * 1000 + cout)
    if (only < 0 || only == pc.cin * 1000 + pc.cout) { if (kill & 1) a.xl_sa0 = 0; if (kill & 2) a.xl_sb1 = 0; } }
  // LDS-DMA staging variants (fast arithmetic): resident weights on maps >= 16 x 16, weight ring on maps >= 8 x 8
  // Small batches (the encoder's one-CU-per-call use): the throughput tiling would put a whole layer on 1-4 workgroups
  // that stream all its weights through their LDS one after the other.  The latency variants cut the couts into 32-channel
  // tiles (4x more workgroups, 4x fewer weight bytes each) on the same packed weights.
  const long lat_px = tuning().lat_pixels;  // output pixels of the launch; measured crossover 13-33 k per layer; 0 disables
  // hi+lo weights on the fast tiling (MLT_MODEL_W2): its stride-1 layers with >= 64 channels have ONE per-conv form, the 32-cout x 128-pixel
  // variant (large launches of those layers go through chain_kernel<..., W2>; this is the bit-identical small-launch / fallback form)
  // (its stride-2 layers share their tiling with the exact packing and run that tier's kernels at any launch size)
  const bool no_exact_lat = tuning().no_exact_lat;
  const bool lat = !(pc.exact && no_exact_lat) && pc.lat && hout >= 8 && (pc.w2 ? pc.stride == 1 : (long)n * hout * hout <= lat_px);
  const int dma = (pc.exact || pc.w2 || lat) ? 0 : (pc.dma == 1 && hout >= 16) ? 1 : (pc.dma == 2 && hout >= 8) ? 2 : 0;
  const int MT = lat ? 128 : dma == 2 ? pc.mt_dma : pc.mt;
  const int nsplit = pc.xl ? 6 : pc.exact ? 2 : pc.w2 ? 4 : 1;  // (mlt_launch_conv; 6 = exact-lite: the exact arithmetic's geometry, FP8 cross terms)
  const int act_planes = (nsplit == 2 || nsplit == 6) ? 2 : 1;
  int tw = hout < 32 ? hout : 32;
  int th = MT / tw < hout ? MT / tw : hout;
  int spw = MT / (tw * th);
  const int halo = pc.taps == 9 ? 3 : 1;
  const int PS = pc.kc * 2 + 16;
  const int ph = (th - 1) * pc.stride + halo, pw = (tw - 1) * pc.stride + halo;
  // row pitch (pixels): stride 2 keeps even / odd columns in two halves; 8-wide maps need RP = 2 (mod 4) so that the
  // two rows a ds_read_b128 lane group reads sit 8 (mod 16) pixels apart (mlt_kernels.hip, lane ranking)
  int half = pc.stride == 2 ? (pw + 1) / 2 : 0;
  if (pc.stride == 2 && tw == 8 && (2 * half) % 4 != 2) ++half;
  int rp = pc.stride == 2 ? 2 * half : pw;
  if (pc.stride == 1 && tw == 8) while (rp % 4 != 2) ++rp;
  // LDS budget of the patch planes: 64 KiB -- or, for the exact arithmetic (two planes), what the two-deep weight ring of the tiling leaves of the
  // 160 KiB (round 4: the 128 -> 256 stride-2 layer on 8 x 8 maps needs 96 KiB for TWO samples per tile; with one, half of the tile's waves idled)
  size_t patch_budget = 64 * 1024;
  if (nsplit == 2 || nsplit == 6) {
    const int tt = pc.taps + (pc.has_sc ? 1 : 0), nbuf = tt / pc.gt > 1 ? 2 : 1;
    const size_t ring = (size_t)nbuf * 2 * pc.gt * (pc.kc / 16) * (pc.ct / 32) * 1024;
    const bool big = tuning().exact_patch_big;
    if (big && ring + 64 * 1024 < 160 * 1024) patch_budget = 160 * 1024 - ring;
  }
  while (spw > 1 && (((size_t)spw * ph * rp * PS + 1023) / 1024 * 1024) * act_planes > patch_budget) spw /= 2;
  a.tw_l = ilog2(tw); a.th_l = ilog2(th); a.spw_l = ilog2(spw);
  a.ph = ph; a.pw = pw; a.rp = rp; a.half = half;
  auto magic = [](int d) { return d < 2 ? 0u : (uint32_t)((0x100000000ull + d - 1) / d); };  // 0 encodes d == 1 (1x1 convs on 1x1 maps)
  a.pw_magic = magic(pw); a.ph_magic = magic(ph);
  a.patch_bytes = (int)((((size_t)spw * ph * rp * PS) + 1023) / 1024 * 1024);
  if (dma) {  // two unpadded, swizzled patch buffers; the row pitch keeps the rules above
    a.patch_bytes = (int)((((size_t)spw * ph * rp * pc.kc * 2) + 1023) / 1024 * 1024);
    a.rp_magic = magic(rp);
    a.zero = ctx->zero_page;
  }
  const int extra_lds = 0;
  const int hw = hout * hout;
  a.gap = io.gap; a.gap_slots = gap_slots(hw); a.gap_l = hw >= 32 ? 5 : ilog2(hw);
  a.ntiles = ((n + spw - 1) / spw) * (hout / th) * (hout / tw);
  // persistent workgroups: at most MLT_WG_PER_CU (default 2) x 256 CUs per cout tile, each looping over tiles
  const int wg_cap = tuning().wg_cap;
  // only the weights-resident kernels (single weight step, single channel chunk) are persistent (mlt_kernels.hip PERSIST)
  const int gt = (nsplit == 4 && !lat) ? pc.gt_w2 : pc.gt;  // the hi+lo-weights tier has its own taps-per-step (mlt_conv_cfg)
  const bool persistent = (gt == pc.taps + (pc.has_sc ? 1 : 0) && pc.cin == pc.kc) || dma == 2;
  // ring-DMA: one 16-wave or two 8-wave workgroups per CU, counted over all cout tiles
  const int cap = dma == 2 ? wg_cap * ((pc.mt_dma >= 256 || pc.stride == 2) ? 1 : 2) / (pc.cout / pc.ct) : wg_cap;  // stride 2: LDS fits one
  const int grid_x = (persistent && a.ntiles > cap) ? cap : a.ntiles;
  char name[48];
  std::snprintf(name, sizeof name, "conv3x3_s%d_%dto%d_h%d%s", pc.stride, pc.cin, pc.cout, hout, pc.has_sc ? "+sc" : "");
  const double px = (double)n * hw;
  const double flops = 2.0 * px * pc.cout * pc.cin * (pc.taps + (pc.has_sc ? 1 : 0));
  const double in_bytes = (double)n * hin * hin * pc.cin * 2;
  const double bytes = in_bytes + px * pc.cout * 2 * ((io.y ? 1 : 0) + (io.y_sc ? 1 : 0) + (io.res ? 1 : 0)) + (double)pc.w.size() * 2;
  Launch L{ctx};
  hipEvent_t e0 = nullptr, e1 = nullptr;
  char variant[96];
  std::snprintf(variant, sizeof variant, "%s%s%s%s%s", nsplit == 1 ? "single pass" : nsplit == 2 ? "exact" : nsplit == 4 ? "hi+lo weights" : nsplit == 6 ? "exact-lite" : "?",
                pc.taps == 1 ? ", centre tap" : lat ? ", latency tiles" : dma == 1 ? ", resident weights + LDS-DMA patches" : dma == 2 ? ", weight ring + LDS-DMA" : "",
                io.y_c16 ? ", y chunk-major" : "", io.ysc_c16 ? ", sc chunk-major" : "", io.gap ? ", GAP" : "");
  int rc = L.prof_begin(name, flops, bytes, e0, e1, variant);
  if (rc) return rc;
  plan_note(ctx, {{"x", io.x}, {"y", io.y}, {"y_sc", io.y_sc}, {"res", io.res}, {"gap", io.gap}},
            {{"x_lo", (long)io.x_lo}, {"y_lo", (long)io.y_lo}, {"res_lo", (long)io.res_lo}, {"ysc_lo", (long)io.ysc_lo}, {"relu", io.relu}, {"grid", grid_x}});
  LAUNCH_TRY(ctx, mlt_launch_conv(pc.cin, pc.cout, pc.stride, nsplit, pc.taps == 1 ? MLT_CONV_CENTRE : lat ? MLT_CONV_LATENCY : dma ? MLT_CONV_DMA : MLT_CONV_DEFAULT, a, grid_x, extra_lds, ctx->stream));
  if ((rc = L.prof_end(e1))) return rc;
  if (io.y && (rc = debug_dump(ctx, name, io.y, (size_t)px * pc.cout * 2))) return rc;
  if (io.y_sc && (rc = debug_dump(ctx, (std::string(name) + "_sc").c_str(), io.y_sc, (size_t)px * pc.cout * 2))) return rc;
  return MLT_OK;
}

// First layer: raw Pel planes -> t = relu(bn1(conv1(stem x))) and sc = bn(shortcut(stem x)) in one composed kernel.
int run_stem5(mlt_ctx *ctx, const mlt::PackedConv &pc, int n, int S, const int16_t *d_org, long org_rs, long org_cs, const int16_t *d_pred,
              long pred_rs, long pred_cs, void *y, void *y_sc, size_t lo_off) {
  const int hout = S / 2;
  Stem5Args a{};
  a.org = d_org; a.pred = d_pred; a.org_row_stride = org_rs; a.org_cu_stride = org_cs; a.pred_row_stride = pred_rs; a.pred_cu_stride = pred_cs;
  a.w = pc.d_w; a.bias = pc.d_bias; a.bias_sc = pc.d_bias_sc; a.y = y; a.y_sc = y_sc;
  a.y_lo_off = lo_off; a.ysc_lo_off = lo_off; a.w_lo_off = pc.plane_halves * 2; a.acc_scale = pc.acc_scale;
  a.n = n; a.s_l = ilog2(S); a.hout_l = ilog2(hout);
  const int MT = 256;
  const int tw = hout < 32 ? hout : 32;
  const int th = MT / tw < hout ? MT / tw : hout;
  const int spw = MT / (tw * th);
  a.tw_l = ilog2(tw); a.th_l = ilog2(th); a.spw_l = ilog2(spw);
  a.rh = 2 * th + 3; a.rw = 2 * tw + 3; a.halfw = (a.rw + 1) / 2;
  a.rw_magic = (uint32_t)((0x100000000ull + a.rw - 1) / a.rw);
  a.rh_magic = (uint32_t)((0x100000000ull + a.rh - 1) / a.rh);
  const int lds = spw * a.rh * 2 * a.halfw * 4;
  const int grid_x = ((n + spw - 1) / spw) * (hout / th) * (hout / tw);
  char name[48];
  std::snprintf(name, sizeof name, "stem5x5_s2_2to32_h%d+sc", hout);
  const double px = (double)n * hout * hout;
  Launch L{ctx};
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = L.prof_begin(name, 2.0 * px * 32 * (50 + 18), (double)n * S * S * 4 + px * 32 * 2 * 2, e0, e1, pc.xl ? "exact-lite" : pc.exact ? "exact" : pc.w2 ? "hi+lo weights" : "single pass");
  if (rc) return rc;
  plan_note(ctx, {{"y", y}, {"y_sc", y_sc}}, {{"lo", (long)lo_off}, {"grid", grid_x}});
  LAUNCH_TRY(ctx, mlt_launch_stem5(a, pc.exact ? 2 : pc.w2 ? 3 : 1, grid_x, lds, ctx->stream));
  if ((rc = L.prof_end(e1))) return rc;
  if ((rc = debug_dump(ctx, name, y, (size_t)px * 32 * 2))) return rc;
  return debug_dump(ctx, (std::string(name) + "_sc").c_str(), y_sc, (size_t)px * 32 * 2);
}

// Whole layer0.0 (composed first layer + conv2 + shortcut + relu) from the raw planes in one kernel (fast, H >= 32).
int run_stem_block(mlt_ctx *ctx, const mlt::Model &m, int n, int S, const int16_t *d_org, long org_rs, long org_cs, const int16_t *d_pred,
                   long pred_rs, long pred_cs, void *y, int32_t *d_flat, bool flat_is_clear = false) {
  const int h = S / 2;
  const mlt::PackedConv &c2 = m.blocks[0][0].conv2;
  StemBlockArgs a{};
  a.org = d_org; a.pred = d_pred; a.org_row_stride = org_rs; a.org_cu_stride = org_cs; a.pred_row_stride = pred_rs; a.pred_cu_stride = pred_cs;
  a.w = m.stem_b.d_w; a.w2 = c2.d_w; a.bias = m.stem.d_bias; a.bias_sc = m.stem.d_bias_sc; a.bias2 = c2.d_bias; a.y = y;
  a.flat = d_flat;
  a.w_lo_off = m.stem_b.plane_halves * 2; a.w2_lo_off = c2.plane_halves * 2; a.scale2 = c2.acc_scale;
  if (d_flat && !flat_is_clear) LAUNCH_TRY(ctx, hipMemsetAsync(d_flat, 0, (size_t)n * 4, ctx->stream));  // (flat_is_clear: the consumer of the previous call left it zero)
  a.acc_scale = m.stem.acc_scale; a.n = n; a.hout_l = ilog2(h); a.ntiles = n * (h / 16) * (h / 32);
  const int wg_cap = tuning().wg_cap2;  // one (pipelined) workgroup per CU
  const int grid_x = a.ntiles > wg_cap ? wg_cap : a.ntiles;
  char name[48];
  std::snprintf(name, sizeof name, "stem+block_s2_2to32_h%d(layer0.0)", h);
  const double px = (double)n * h * h;
  Launch L{ctx};
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = L.prof_begin(name, 2.0 * px * 32 * (50 + 18 + 288), (double)n * S * S * 4 + px * 32 * 2, e0, e1, m.w2 ? "hi+lo weights" : "single pass");
  if (rc) return rc;
  plan_note(ctx, {{"y", y}, {"flat", d_flat}}, {{"flat_is_clear", flat_is_clear}, {"grid", grid_x}});
  LAUNCH_TRY(ctx, mlt_launch_stem_block(a, m.w2, grid_x, ctx->stream));
  if ((rc = L.prof_end(e1))) return rc;
  return debug_dump(ctx, name, y, (size_t)px * 32 * 2);
}

// All of layer0 of 128 x 128 CUs in one launch (layer0_stream_kernel): bit-identical to run_stem_block + run_block32, b0 never reaches HBM.
// c5 != nullptr: layer1.0.conv1 + shortcut ride along as a fifth stage (layer0_stream_kernel<true>): y is not written, t -> y_t (NHWC), sc -> y_sc (chunk-major)
int run_layer0_stream(mlt_ctx *ctx, const mlt::Model &m0, const mlt::Model &m1, int n, const int16_t *d_org, long org_rs, long org_cs, const int16_t *d_pred,
                      long pred_rs, long pred_cs, void *y, int32_t *d_flat, bool flat_is_clear, const mlt::PackedConv *c5 = nullptr, void *y_t = nullptr, void *y_sc = nullptr) {
  const mlt::PackedConv &c2 = m0.blocks[0][0].conv2;
  const mlt::Block &B1 = m1.blocks[0][1];
  Layer0Args a{};
  a.org = d_org; a.pred = d_pred; a.org_row_stride = org_rs; a.org_cu_stride = org_cs; a.pred_row_stride = pred_rs; a.pred_cu_stride = pred_cs;
  a.w = m0.stem_b.d_w; a.w2 = c2.d_w; a.w3 = B1.conv1.d_w; a.w4 = B1.conv2.d_w;
  a.bias = m0.stem.d_bias; a.bias_sc = m0.stem.d_bias_sc; a.bias2 = c2.d_bias; a.bias3 = B1.conv1.d_bias; a.bias4 = B1.conv2.d_bias;
  a.y = y; a.flat = d_flat; a.acc_scale = m0.stem.acc_scale; a.n = n;
  if (c5) { a.w5 = c5->d_w; a.bias5 = c5->d_bias; a.bias5_sc = c5->d_bias_sc; a.scale5 = c5->acc_scale; a.y_t = y_t; a.y_sc = y_sc; }
  if (d_flat && !flat_is_clear) LAUNCH_TRY(ctx, hipMemsetAsync(d_flat, 0, (size_t)n * 4, ctx->stream));
  const int wg_cap = tuning().wg_cap0;
  const int grid_x = n > wg_cap ? wg_cap : n;
  const double px = (double)n * 64 * 64;
  Launch L{ctx};
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = c5 ? L.prof_begin("layer0_stream_h64(stem+layer0+layer1.0.conv1+sc)", 2.0 * px * 32 * (50 + 18 + 3 * 288) + 2.0 * (px / 4) * 64 * (288 + 32),
                             (double)n * 128 * 128 * 4 + (px / 4) * 64 * 2 * 2, e0, e1)
              : L.prof_begin("layer0_stream_h64(stem+layer0.0+layer0.1)", 2.0 * px * 32 * (50 + 18 + 3 * 288), (double)n * 128 * 128 * 4 + px * 32 * 2, e0, e1);
  if (rc) return rc;
  const bool mfma32 = tuning().l0_mfma32;   // round 5's MFMA shape (A/B; the results are the same bits)
  plan_note(ctx, {{"y", c5 ? nullptr : y}, {"y_t", c5 ? y_t : nullptr}, {"y_sc", c5 ? y_sc : nullptr}, {"flat", d_flat}}, {{"flat_is_clear", flat_is_clear}, {"grid", grid_x}});
  LAUNCH_TRY(ctx, mlt_launch_layer0_stream(a, c5 != nullptr, mfma32, grid_x, ctx->stream));
  if ((rc = L.prof_end(e1))) return rc;
  if (c5) {
    if ((rc = debug_dump(ctx, "conv3x3_s2_32to64_h32+sc", y_t, (size_t)(px / 4) * 64 * 2))) return rc;
    return debug_dump(ctx, "conv3x3_s2_32to64_h32+sc_sc", y_sc, (size_t)(px / 4) * 64 * 2);
  }
  return debug_dump(ctx, "block_s1_32_h64(conv1+conv2)", y, (size_t)px * 32 * 2);  // (the dump carries the two-launch form's name: same tensor)
}

// Fused identity BasicBlock of the 32-channel stage (fast arithmetic, H >= 32): conv1 -> LDS -> conv2 + residual.
int run_block32(mlt_ctx *ctx, const mlt::Block &B, int n, int h, const void *x, void *y) {
  Block32Args a{};
  const bool w2 = B.conv1.w2;  // hi+lo weights: 8 x 32 tiles (four resident weight planes)
  a.x = x; a.y = y; a.w1 = B.conv1.d_w; a.w2 = B.conv2.d_w; a.bias1 = B.conv1.d_bias; a.bias2 = B.conv2.d_bias;
  a.w1_lo_off = B.conv1.plane_halves * 2; a.w2_lo_off = B.conv2.plane_halves * 2; a.scale1 = B.conv1.acc_scale; a.scale2 = B.conv2.acc_scale;
  a.n = n; a.h_l = ilog2(h); a.ntiles = n * (h / (w2 ? 8 : 16)) * (h / 32);
  const int wg_cap = tuning().wg_cap;
  const int grid_x = a.ntiles > wg_cap ? wg_cap : a.ntiles;
  char name[48];
  std::snprintf(name, sizeof name, "block_s1_32_h%d(conv1+conv2)", h);
  const double px = (double)n * h * h;
  Launch L{ctx};
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = L.prof_begin(name, 2.0 * px * 32 * 32 * 9 * 2, px * 32 * 2 * 2 + 2.0 * 18 * 1024, e0, e1, w2 ? "hi+lo weights" : "single pass");
  if (rc) return rc;
  plan_note(ctx, {{"x", x}, {"y", y}}, {{"grid", grid_x}});
  LAUNCH_TRY(ctx, mlt_launch_block32(a, w2, grid_x, ctx->stream));
  if ((rc = L.prof_end(e1))) return rc;
  return debug_dump(ctx, name, y, (size_t)px * 32 * 2);
}

// BasicBlock tail of a stage as ONE launch (chain_kernel): b0 = relu(bn2(conv2 t) + sc); t1 = relu(bn1(conv1 b0));
// out = relu(bn2(conv2 t1) + b0) (+ GAP).  Fast arithmetic, stages whose whole sample fits the LDS (128 channels @ 16 x 16).
// s2_in != NULL: the stage's stride-2 conv + shortcut run inside the same launch from the stage input s2_in ([n][2h][2h][c/2]);
// t / sc are then unused.
int run_chain3(mlt_ctx *ctx, const mlt::Block &B0, const mlt::Block &B1, int n, int h, const void *t, const void *sc, void *y, float *gap,
               const void *s2_in = nullptr, bool x_c16 = false, bool y_c16 = false, void *b0_hbm = nullptr, bool sc_c16 = false) {
  const int c = B0.conv2.cout;
  ChainArgs a{};
  a.x = s2_in ? s2_in : t; a.nconv = 3; a.y = y; a.gap = gap; a.n = n; a.zero = ctx->zero_page;
  a.x_c16 = x_c16 ? 1 : 0; a.y_c16 = y_c16 ? 1 : 0; a.res0_c16 = sc_c16 ? 1 : 0;
  const mlt::PackedConv *pcs[3] = {&B0.conv2, &B1.conv1, &B1.conv2};
  const bool w2 = B0.conv2.w2;  // hi+lo weights: chain_kernel<..., W2> (two planes per ring step)
  for (int k = 0; k < 3; ++k) {
    a.cv[k].w = pcs[k]->d_w; a.cv[k].bias = pcs[k]->d_bias; a.cv[k].acc_scale = pcs[k]->acc_scale; a.cv[k].relu = 1;
    a.cv[k].w_lo_off = pcs[k]->plane_halves * 2;
    a.cv[k].lo8_scale = 0x01010101 * ((127 - pcs[k]->lo8_exp) & 0xFF);  // E8M0 byte of the FP8 lo plane's scale (2^-lo8_exp), all four bytes
  }
  a.cv[0].res_mode = s2_in ? 2 : 1; a.cv[0].res = sc; a.cv[0].save = 1; a.cv[2].res_mode = 2;
  if (c == 64) {  // no room for b0 in registers: conv 0 writes it to HBM (b0_hbm), the last conv reads it back as its residual
    a.cv[0].save = 0; a.cv[0].y = b0_hbm; a.cv[2].res_mode = 1; a.cv[2].res = b0_hbm;
  }
  if (s2_in) {
    const mlt::PackedConv &p2 = B0.conv1_s2c;
    a.s2_w = p2.d_w; a.s2_bias = p2.d_bias; a.s2_bias_sc = p2.d_bias_sc; a.s2_scale = p2.acc_scale;
  }
  const int hw = h * h;
  a.gap_slots = gap_slots(hw); a.gap_l = hw >= 32 ? 5 : ilog2(hw);
  const int wg_cap = tuning().wg_cap;
  const int spw = c == 256 ? 2 : 1;            // samples per workgroup (64 KiB of activations)
  const int ntiles = (n + spw - 1) / spw;
  const int grid_x = ntiles > wg_cap ? wg_cap : ntiles;  // one workgroup per CU (its LDS is full), persistent over tiles
  char name[48];
  std::snprintf(name, sizeof name, s2_in ? "stage_%d_h%d(s2+sc,conv2,conv1,conv2)" : "chain3_s1_%d_h%d(conv2+conv1+conv2)", c, h);
  const double px = (double)n * hw;
  const double flops = 3.0 * 2.0 * px * c * c * 9 + (s2_in ? 2.0 * px * c * (c / 2) * 10 : 0.0);
  const double bytes = (s2_in ? px * 4 * (c / 2) * 2 : px * c * 2 * 2) + (y ? px * c * 2 : 0.0) + 3.0 * (double)B0.conv2.w.size() * 2 +
                       (s2_in ? (double)B0.conv1_s2c.w.size() * 2 : 0.0);
  Launch L{ctx};
  hipEvent_t e0 = nullptr, e1 = nullptr;
  char variant[96];
  std::snprintf(variant, sizeof variant, "%s%s%s%s%s", w2 ? "hi+lo weights" : "single pass", x_c16 ? ", x chunk-major" : "", sc_c16 ? ", sc chunk-major" : "", y_c16 ? ", y chunk-major" : "",
                gap ? ", GAP" : "");
  int rc = L.prof_begin(name, flops, bytes, e0, e1, variant);
  if (rc) return rc;
  plan_note(ctx, {{"x", a.x}, {"sc", s2_in ? nullptr : sc}, {"y", y}, {"gap", gap}, {"b0", c == 64 ? b0_hbm : nullptr}}, {{"grid", grid_x}});
  LAUNCH_TRY(ctx, mlt_launch_chain(c, h, s2_in != nullptr, ctx->lds_oob_zero, w2, a, grid_x, ctx->stream));
  if ((rc = L.prof_end(e1))) return rc;
  if (y && (rc = debug_dump(ctx, name, y, (size_t)px * c * 2))) return rc;
  return MLT_OK;
}

// The three stride-1 convs of layer1 (64 channels at 32 x 32) as ONE streaming launch (layer1_stream_kernel): bit-identical to run_chain3's
// 64-channel chain, weights resident in registers, b0 never in HBM.  t NHWC, sc chunk-major (what layer0_stream_kernel<true> / the stride-2 launch write).
int run_layer1_stream(mlt_ctx *ctx, const mlt::Block &B0, const mlt::Block &B1, int n, const void *t, const void *sc, void *y, float *gap, bool y_c16) {
  Layer1Args a{};
  a.t = t; a.sc = sc; a.y = y; a.y_c16 = y_c16 ? 1 : 0; a.gap = gap; a.gap_slots = gap_slots(32 * 32); a.n = n;
  const mlt::PackedConv *pcs[3] = {&B0.conv2, &B1.conv1, &B1.conv2};
  for (int k = 0; k < 3; ++k) { a.w[k] = pcs[k]->d_w; a.bias[k] = pcs[k]->d_bias; a.scale[k] = pcs[k]->acc_scale; }
  const int wg_cap = tuning().wg_cap1;
  const int grid_x = n > wg_cap ? wg_cap : n;
  const double px = (double)n * 32 * 32;
  Launch L{ctx};
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc = L.prof_begin("layer1_stream_h32(conv2+conv1+conv2)", 3.0 * 2.0 * px * 64 * 64 * 9, px * 64 * 2 * 3 + 3.0 * 72 * 1024, e0, e1, y_c16 ? "single pass, y chunk-major, GAP" : "single pass, GAP");
  if (rc) return rc;
  // round 6: the 16x16x32 MFMA form exists (same bits) but measures 2 % SLOWER here than round 5's 32x32x16 form (0.882 against 0.863 ms, same box, alternating:
  // profiles/r06e_mfma16_ab.txt) -- the bare loop's +11 % is a clock effect at 1.6 GHz, this kernel already holds ~2.1 GHz and pays for twice the MFMA issue
  // slots and 8-byte epilogue accesses: it stays on 32x32x16 (MLT_TUNING=1 MLT_L1_MFMA16=1 selects the other form)
  const bool mfma32 = tuning().l1_mfma32;
  plan_note(ctx, {{"t", t}, {"sc", sc}, {"y", y}, {"gap", gap}}, {{"grid", grid_x}});
  LAUNCH_TRY(ctx, mlt_launch_layer1_stream(a, mfma32, grid_x, ctx->stream));
  if ((rc = L.prof_end(e1))) return rc;
  return debug_dump(ctx, "chain3_s1_64_h32(conv2+conv1+conv2)", y, (size_t)px * 64 * 2);
}

// ---- launch plans (round 6) ----
// WHAT a batch launches is decided once per (size, tier, batch class, alignment) -- plan_network, a pure function of the models' packing, the tier's
// launch-unit masks and a handful of booleans derived from n and the planes' alignment -- and cached in the size's state; the hot path (run_network)
// binds the workspace and walks the list.  mlt_plan_describe shows the walk on the CPU (tests/test_launch_plan_cpu.py, scripts/plan_matrix.py).
struct NetCfg {   // the arithmetic of a pass: every launch unit takes its weights from one of up to three models
  int S = 0;
  mlt::Model *m = nullptr, *mback = nullptr, *mx = nullptr;   // main | hi+lo weights for the units of back_mask | exact for the units of x_units
  unsigned back_mask = 0, x_units = 0;
  mlt::Model &model_of(int s, int u) const { return ((x_units >> (2 * s + u)) & 1u) ? *mx : (mback && ((back_mask >> (2 * s + u)) & 1u)) ? *mback : *m; }
  int h_in(int s) const { int h = S; for (int i = 0; i < s; ++i) h = h / 2 > 0 ? h / 2 : 1; return h; }   // map side at the input of stage s
};
// the batch class: everything about a CALL that the choice of launches depends on
enum : unsigned { CLS_QUADS = 1u, CLS_FLAT = 2u, CLS_L0_STREAM = 4u, CLS_L1_STREAM = 8u, CLS_CHAIN0 = 16u /* << s: stage s is large enough for a whole-stage launch */ };
unsigned batch_class(const mlt_ctx *ctx, const NetCfg &c, int n, const int16_t *d_org, long org_rs, long org_cs, const int16_t *d_pred, long pred_rs, long pred_cs, bool has_flat) {
  (void)ctx;
  const Tuning &tn = tuning();
  unsigned cls = 0;
  // stem_block_kernel / layer0_stream_kernel fetch 4-pixel quads with 8-byte loads: planes 8-byte aligned, strides multiples of 4 elements
  if ((((uintptr_t)d_org | (uintptr_t)d_pred) & 7) == 0 && ((org_rs | org_cs | pred_rs | pred_cs) & 3) == 0) cls |= CLS_QUADS;
  if (has_flat) cls |= CLS_FLAT;
  // round 5: batches of 128 x 128 CUs run ALL of layer0 in one streaming launch; round 6: from 128 CUs on -- the measured crossover (docs/KERNEL_NOTES.md,
  // "Where the streaming launches start to pay": 128 CUs +3 %, 192 CUs +12 %); the same for layer1's three stride-1 convs
  if (tn.l0_stream_min > 0 && n >= tn.l0_stream_min) cls |= CLS_L0_STREAM;
  if (tn.l1_stream_min > 0 && n >= tn.l1_stream_min) cls |= CLS_L1_STREAM;
  for (int s = 1; s < c.m->n_stages; ++s) {   // small launches keep the per-conv latency variants: a chain runs its convs one after the other on n workgroups
    const int h = c.h_in(s), ho = h / 2 > 0 ? h / 2 : 1;
    if ((long)n * ho * ho > tn.lat_pixels) cls |= CLS_CHAIN0 << s;
  }
  return cls;
}


NetPlan plan_network(const mlt_ctx *ctx, const NetCfg &c, unsigned cls) {
  const Tuning &tn = tuning();
  const mlt::Model &m = *c.m;
  NetPlan P;
  auto add = [&](PlanStep::Kind k, int s) -> PlanStep & { PlanStep st{}; st.kind = k; st.s = (int8_t)s; P.push_back(st); return P.back(); };
  // Which stages run as chain / whole-stage launches (fast arithmetic, large launches).  Asked for stage s AND for stage s + 1: a stage whose successor
  // is a whole-stage kernel writes its output chunk-major (ConvArgs.y_c16).
  auto wants_chain = [&](int s) -> bool {
    if (s <= 0 || s >= m.n_stages || tn.no_chain || !(cls & (CLS_CHAIN0 << s))) return false;
    const mlt::Model &mm = c.model_of(s, 1);
    if (mm.exact || (mm.w2 && !ctx->lds_oob_zero)) return false;  // (the hi+lo-weights chains exist in the padding-from-beyond-the-LDS form only)
    const int h = c.h_in(s), ho = h / 2 > 0 ? h / 2 : 1;
    const mlt::PackedConv &c2 = mm.blocks[s][0].conv2;
    // The 64-channel chain (a 128 KiB sample per workgroup, 8 accumulators per wave; b0 through HBM): 1.19 ms against 3 x 0.40 ms
    // for the launch itself, but the step gains 4 % (less HBM traffic -> the power-limited chip clocks the other kernels higher).
    const bool packing_ok = (m.planes[s] == 64 ? (c2.ct == 64 && c2.gt == 9 && tn.chain64) : (c2.ct == 128 && c2.gt == 3)) && (!mm.w2 || c2.lo8 || m.planes[s] == 64);  // what chain_kernel<C> streams
    return mlt_chain_supported(m.planes[s], ho) && c2.taps == 9 && c2.kc == 64 && packing_ok;
  };
  // (the whole-stage form -- stride-2 conv + shortcut inside the launch -- exists for the single pass only: both units of the stage on `m`)
  auto wants_s2 = [&](int s) -> bool {
    if (!wants_chain(s) || tn.no_chain_s2 || c.model_of(s, 0).w2 || c.model_of(s, 1).w2) return false;
    const mlt::PackedConv &p2 = c.model_of(s, 0).blocks[s][0].conv1_s2c;
    return ctx->plan ? !p2.w.empty() : p2.d_w != nullptr;
  };
  bool cur_c16 = false;    // layout of the stage input
  bool l0_front = false;   // the layer0 streaming launch carried layer1.0.conv1 + shortcut
  for (int s = 0; s < m.n_stages; ++s) {
    const bool last = s == m.n_stages - 1;
    const mlt::Model &ms = c.model_of(s, 0);  // first launch unit: layer0.0 / the stride-2 conv + shortcut
    const mlt::Model &mt = c.model_of(s, 1);  // second unit: layer0.1, or conv2 of block 0 + block 1 of the later stages
    const int h = c.h_in(s), ho = h / 2 > 0 ? h / 2 : 1;
    const bool fused_b0 = s == 0 && !ms.exact && ho >= 32 && !tn.no_block_fusion && (cls & CLS_QUADS);
    if (s == 0 && (cls & CLS_FLAT) && !fused_b0) add(PlanStep::FLAT_STAT, 0);
    if (fused_b0 && ho == 64 && !ms.w2 && !mt.exact && !mt.w2 && (cls & CLS_L0_STREAM)) {
      // ... and with it the stride-2 conv + shortcut that open layer1, when the 64-channel chain follows (it wants sc chunk-major) and layer1's first
      // unit runs the single pass too: layer0's output then never reaches HBM
      const mlt::Model &m10 = c.model_of(1, 0);
      const mlt::PackedConv &c5 = m10.blocks[1][0].conv1;
      l0_front = !tn.no_l0_s5 && m.n_stages > 1 && !m10.exact && !m10.w2 && wants_chain(1) && !wants_s2(1) && m.planes[1] == 64 && !tn.no_c16 &&
                 c5.has_sc && c5.taps == 9 && c5.kc == 32 && c5.ct == 64 && c5.cin == 32 && c5.cout == 64 && c5.stride == 2;
      add(PlanStep::LAYER0_STREAM, 0).front = l0_front;
      continue;
    }
    if (fused_b0) add(PlanStep::STEM_BLOCK, 0);   // raw planes -> b0 in ONE kernel (t and sc never leave the chip)
    else {
      const bool chain = wants_chain(s), chain_s2 = wants_s2(s);   // chain_s2: the stride-2 conv + shortcut join the launch
      bool sc_c16 = false;
      if (s == 0) add(PlanStep::STEM5, 0);
      else if (chain_s2) {}
      else if (s == 1 && l0_front) sc_c16 = true;   // layer0_stream_kernel<true> wrote t (pool0) and sc (pool1, chunk-major) already
      else {
        // the 64-channel chain reads sc as a residual in accumulator order: chunk-major makes that one cache line per lane quad
        // (t -- the chain's input, fetched by LDS-DMA -- stays NHWC: chunk-major, the stride-2 kernel's stores gained what the chain's
        // DMA lost, 0.455 -> 0.438 ms against 1.06 -> 1.08 ms)
        sc_c16 = chain && m.planes[s] == 64 && !tn.no_c16 && !ms.exact;  // (the exact kernels write NHWC)
        add(PlanStep::CONV_S2, s).sc_c16 = sc_c16;
      }
      if (chain) {  // rest of the stage (or all of it) in one launch: activations stay in LDS, b0 in registers
        const bool out_c16 = !last && !tn.no_c16 && wants_s2(s + 1);
        // round 5: large batches run the 64-channel stage's three stride-1 convs as a streaming launch (same bits as the chain)
        const mlt::PackedConv &q2 = mt.blocks[s][0].conv2;
        const bool stream = s == 1 && !chain_s2 && m.planes[s] == 64 && ho == 32 && sc_c16 && !mt.w2 && !mt.exact && (cls & CLS_L1_STREAM) &&
                            q2.taps == 9 && q2.kc == 64 && q2.ct == 64 && !last;
        PlanStep &st = add(stream ? PlanStep::LAYER1_STREAM : PlanStep::CHAIN, s);
        st.inside_s2 = chain_s2; st.x_c16 = cur_c16; st.sc_c16 = sc_c16; st.y_c16 = out_c16;
        cur_c16 = out_c16;
        continue;
      }
      add(PlanStep::CONV_B0C2, s);   // b0 = relu(bn2(conv2 t) + sc)
    }
    // block 1 (identity shortcut)
    if (s == 0 && !mt.exact && ho >= 32 && !tn.no_block_fusion) { add(PlanStep::BLOCK32, 0); continue; }  // 32-channel identity block in ONE kernel
    add(PlanStep::CONV_B1C1, s);
    PlanStep &st = add(PlanStep::CONV_B1C2, s);
    st.y_c16 = !last && !tn.no_c16 && wants_s2(s + 1);
    cur_c16 = st.y_c16;
  }
  add(PlanStep::HEADS, m.n_stages - 1);
  return P;
}

// the network in the size's main arithmetic: `model` (fast, or exact when that is the configured / calibrated arithmetic), or -- middle
// tier -- `model_w2` (hi+lo WEIGHTS on the fast tiling: the W2 forms of the fused kernels)
// d_flat != NULL: also produce the flat-content guard's per-CU statistic (fused into the first kernel where that kernel reads
// the raw planes as aligned quads, else by flat_stat_kernel)
// mback != NULL (hi+lo-weights tiers): the launch UNITS whose bit is set in back_mask (bit 2 s: layer0.0 / the stride-2 conv + shortcut of
// layer s; bit 2 s + 1: layer0.1 / the three stride-1 convs of layer s) run `mback` (the hi+lo-weights model; same single fp16 activation
// planes, so the two models' units compose freely), the others `m` (single-pass kernels).
// mx != NULL (round 4): the units of x_units run `mx` (the EXACT arithmetic's per-conv kernels; a lo plane behind every tensor such a unit
// writes; a unit that reads a single-plane producer's output takes its lo part as zero).
int run_network(mlt_ctx *ctx, SizeState &st, mlt::Model &m, int n, const int16_t *d_org, long org_rs, long org_cs, const int16_t *d_pred,
                long pred_rs, long pred_cs, const int32_t *d_poc, const int32_t *d_qp, int32_t *d_split, float *d_logits, int32_t *d_flat = nullptr,
                mlt::Model *mback = nullptr, unsigned back_mask = 0, const GuardTail *tail = nullptr, bool flat_is_clear = false,
                mlt::Model *mx = nullptr, unsigned x_units = 0, float *d_mag = nullptr) {
  const int S = st.size;
  if (!mx) x_units = 0;
  if (!mback) back_mask = 0;
  NetCfg c;
  c.S = S; c.m = &m; c.mback = mback; c.mx = mx; c.back_mask = back_mask; c.x_units = x_units;
  const unsigned cls = batch_class(ctx, c, n, d_org, org_rs, org_cs, d_pred, pred_rs, pred_cs, d_flat != nullptr);
  // the plan: cached per (models, unit masks, batch class) once the size is loaded -- while a load is in flight (the calibration prices candidate tiers and
  // rebuilds models in place) it is built per call
  NetPlan fresh;
  const NetPlan *plan;
  if (st.loaded && !ctx->plan) {
    const PlanKey key{&m, mback, mx, back_mask, x_units, cls};
    auto it = st.plans.find(key);
    if (it == st.plans.end()) it = st.plans.emplace(key, plan_network(ctx, c, cls)).first;
    plan = &it->second;
  } else {
    fresh = plan_network(ctx, c, cls);
    plan = &fresh;
  }
  int rc = ctx->plan ? MLT_OK : ensure_ws(ctx, ws_per_cu(m, S, x_units != 0) * (size_t)n);   // (plan mode: the workspace is carved from a fake base and never touched)
  if (rc) return rc;
  // carve the workspace
  char *p = ctx->ws;
  auto carve = [&](size_t bytes) { char *r = p; p += (bytes + 255) / 256 * 256; return (void *)r; };
  const int h0 = S / 2 > 0 ? S / 2 : 1;
  // a lo plane behind every activation (x_units: room for one behind every buffer, used by the units of the mask only)
  const int nplanes = (m.exact || x_units) ? 2 : 1;
  void *pool[4];
  for (int i = 0; i < 4; ++i) pool[i] = carve((size_t)n * h0 * h0 * 32 * 2 * nplanes);
  void *outs[5];
  float *gaps[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  {
    int h = S;
    for (int s = 0; s < m.n_stages; ++s) {
      h = h / 2 > 0 ? h / 2 : 1;
      outs[s] = carve((size_t)n * h * h * m.planes[s] * 2 * nplanes);
      if (s >= 1) gaps[s] = (float *)carve((size_t)n * gap_slots(h * h) * m.planes[s] * 4);
    }
  }
  Launch L{ctx};
  const bool quad_ok = (cls & CLS_QUADS) != 0;
  for (const PlanStep &ps : *plan) {
    const int s = ps.s;
    const bool last = s == m.n_stages - 1;
    mlt::Model &ms = c.model_of(s, 0), &mt = c.model_of(s, 1);
    const int h = c.h_in(s), ho = h / 2 > 0 ? h / 2 : 1;
    const void *cur = s > 0 ? outs[s - 1] : nullptr;   // the stage input
    // (exact units, round 4: a unit in the exact arithmetic keeps a lo plane behind the tensors it writes and expects one behind those it
    // reads -- a plane of zeros when the producer is a single-plane unit; single-plane units read the hi planes and ignore the offsets)
    const bool ex0 = ms.exact, ex1 = mt.exact;
    const size_t lo_in = ex0 ? (size_t)n * h * h * (s == 0 ? 32 : m.planes[s - 1]) * 2 : 0;  // plane bytes of the stage input
    const size_t lo_st = (ex0 || ex1) ? (size_t)n * ho * ho * m.planes[s] * 2 : 0;             // plane bytes inside the stage
    // an exact unit behind a single-plane unit: its input has no lo part (the producer wrote fp16 values): lo offset 0 = "no lo plane, read
    // zeros" (ConvArgs.x_lo_off / res_lo_off)
    const bool in_has_lo = s > 0 && c.model_of(s - 1, 1).exact;
    const bool b0_has_lo = s == 0 ? ex0 : ex1;  // (b0 = pool2 is written by unit 0 of layer0, by unit 1 of the later stages)
    int hh = 0;
    ConvIO io;
    switch (ps.kind) {
    case PlanStep::FLAT_STAT: {
      FlatStatArgs fa{};
      fa.org = d_org; fa.pred = d_pred; fa.org_row_stride = org_rs; fa.org_cu_stride = org_cs; fa.pred_row_stride = pred_rs;
      fa.pred_cu_stride = pred_cs; fa.flat = d_flat; fa.n = n; fa.s_l = ilog2(S);
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if ((rc = L.prof_begin("guard_flat_stat", 0.0, (double)n * S * S * 4, e0, e1))) return rc;
      plan_note(ctx, {{"flat", d_flat}}, {{"quads", quad_ok}});
      LAUNCH_TRY(ctx, mlt_launch_flat_stat(fa, quad_ok, ctx->stream));
      rc = L.prof_end(e1);
      break;
    }
    case PlanStep::LAYER0_STREAM:
      rc = run_layer0_stream(ctx, ms, mt, n, d_org, org_rs, org_cs, d_pred, pred_rs, pred_cs, outs[0], d_flat, flat_is_clear,
                             ps.front ? &c.model_of(1, 0).blocks[1][0].conv1 : nullptr, pool[0], pool[1]);
      break;
    case PlanStep::STEM_BLOCK:
      rc = run_stem_block(ctx, ms, n, S, d_org, org_rs, org_cs, d_pred, pred_rs, pred_cs, pool[2], d_flat, flat_is_clear);
      break;
    case PlanStep::STEM5:
      // block 0 (stride 2): ONE kernel gives t = relu(bn1(conv1 x)) and sc = bn(conv1x1 x) (arch:44-55); for s == 0 the same kernel also
      // computes x = stem(raw planes) on the fly (arch:277-278, EncCu.cpp:810-877)
      rc = run_stem5(ctx, ms.stem, n, S, d_org, org_rs, org_cs, d_pred, pred_rs, pred_cs, pool[0], pool[1], lo_st);
      break;
    case PlanStep::CONV_S2:
      io.x = cur; io.y = pool[0]; io.y_sc = pool[1]; io.relu = true;
      io.x_lo = in_has_lo ? lo_in : 0; io.y_lo = lo_st; io.ysc_lo = lo_st;
      io.ysc_c16 = ps.sc_c16;
      rc = run_conv(ctx, ms.blocks[s][0].conv1, n, h, io, &hh);
      break;
    case PlanStep::CHAIN:
      rc = run_chain3(ctx, mt.blocks[s][0], mt.blocks[s][1], n, ho, pool[0], pool[1], last ? nullptr : outs[s], gaps[s], ps.inside_s2 ? cur : nullptr,
                      ps.x_c16, ps.y_c16, pool[2], ps.sc_c16);
      break;
    case PlanStep::LAYER1_STREAM:
      rc = run_layer1_stream(ctx, mt.blocks[s][0], mt.blocks[s][1], n, pool[0], pool[1], outs[s], gaps[s], ps.y_c16);
      break;
    case PlanStep::CONV_B0C2:
      io.x = pool[0]; io.y = pool[2]; io.res = pool[1]; io.relu = true;  // b0 = relu(bn2(conv2 t) + sc)
      io.x_lo = io.y_lo = io.res_lo = lo_st;
      if (s > 0 && ex1 && !ex0) io.x_lo = io.res_lo = 0;  // t and sc came from a single-plane unit
      rc = run_conv(ctx, (s == 0 ? ms : mt).blocks[s][0].conv2, n, ho, io, &hh);
      break;
    case PlanStep::BLOCK32:
      rc = run_block32(ctx, mt.blocks[0][1], n, ho, pool[2], outs[0]);
      break;
    case PlanStep::CONV_B1C1:
      io.x = pool[2]; io.y = pool[3]; io.relu = true;
      io.x_lo = io.y_lo = lo_st;
      if (!b0_has_lo) io.x_lo = 0;
      rc = run_conv(ctx, mt.blocks[s][1].conv1, n, ho, io, &hh);
      break;
    case PlanStep::CONV_B1C2:
      io.x = pool[3]; io.y = last ? nullptr : outs[s]; io.res = pool[2]; io.relu = true; io.gap = gaps[s];
      io.x_lo = io.y_lo = io.res_lo = lo_st;
      if (!b0_has_lo) io.res_lo = 0;
      io.y_c16 = ps.y_c16;
      rc = run_conv(ctx, mt.blocks[s][1].conv2, n, ho, io, &hh);
      break;
    case PlanStep::HEADS: {
      HeadArgs ha{};
      for (int t = 1; t < m.n_stages; ++t) {   // head t - 1 pools stage t's output
        const int hd = t - 1, hs = c.h_in(t + 1);
        ha.gap[hd] = gaps[t]; ha.slots[hd] = gap_slots(hs * hs); ha.w[hd] = m.heads[hd].d_w; ha.b[hd] = m.heads[hd].d_b;
        ha.c[hd] = m.planes[t]; ha.hw[hd] = hs * hs; ha.classes[hd] = m.heads[hd].classes;
      }
      ha.n_heads = m.n_heads; ha.decision_head = st.head_index; ha.poc = d_poc; ha.qp = d_qp; ha.logits = d_logits; ha.split = d_split;
      ha.mag = d_mag;
      if (tail && (n == 1 || tail->next)) {
        ha.g_next = tail->next;   // (NULL: mlt_predict's slot -- one CU, the count is set outright)
        ha.g_count = tail->count; ha.g_idx = tail->idx; ha.g_flat = tail->flat; ha.g_flat_thr = tail->flat_thr; ha.g_near_thr = tail->near_thr; ha.g_margin = tail->margin;
        ha.g_mag_thr = tail->mag_thr;
      }
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if ((rc = L.prof_begin("heads", 0.0, 0.0, e0, e1))) return rc;
      plan_note(ctx, {{"gap0", ha.gap[0]}, {"gap1", ha.gap[1]}, {"gap2", ha.gap[2]}, {"gap3", ha.gap[3]}},
                {{"slots0", ha.slots[0]}, {"slots1", ha.slots[1]}, {"slots2", ha.slots[2]}, {"slots3", ha.slots[3]}, {"c0", ha.c[0]}, {"c1", ha.c[1]}, {"c2", ha.c[2]}, {"c3", ha.c[3]},
                 {"hw0", ha.hw[0]}, {"hw1", ha.hw[1]}, {"hw2", ha.hw[2]}, {"hw3", ha.hw[3]}, {"heads", ha.n_heads}});
      LAUNCH_TRY(ctx, mlt_launch_heads(ha, n, ctx->stream));
      rc = L.prof_end(e1);
      break;
    }
    }
    if (rc) return rc;
  }
  return MLT_OK;
}

int check_size(mlt_ctx *ctx, int size, SizeState **out) {
  const int si = size_index(size);
  if (si < 0) { ctx->err = "unsupported CU size"; return MLT_ERR_ARG; }
  SizeState &st = ctx->sz[si];
  if (!st.enabled || !st.loaded) { ctx->err = "CU size not enabled or weights not loaded"; return MLT_ERR_SIZE_DISABLED; }
  *out = &st;
  return MLT_OK;
}

int ensure_stage(mlt_ctx *ctx, size_t bytes) {
  if (bytes <= ctx->stage_bytes) return MLT_OK;
  if (ctx->stage) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->stage); ctx->stage = nullptr; ctx->stage_bytes = 0; }
  HIP_TRY(ctx, hipMalloc((void **)&ctx->stage, bytes));
  ctx->stage_bytes = bytes;
  return MLT_OK;
}

// ---- parity guards (include/mltcnn.h: flat guard, decision guard) --------------------------------------------------
// Per batch: [flat_stat_kernel] -> fast network -> guard_select_kernel (ascending list of flagged CUs + count) -> 4-byte
// D2H of the count.  Once the host knows the count k it enqueues, for k > 0: gather of the flagged CUs' planes -> exact
// network on k CUs -> scatter of their split modes / logits over the fast results.

int guard_slot(mlt_ctx *ctx, int which, int n, int nl, GuardSlot *g) {
  if (n > ctx->guard_cap_n || nl > ctx->guard_cap_nl || !ctx->guard_dev) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->guard_dev) (void)hipFree(ctx->guard_dev);
    ctx->guard_dev = nullptr;
    const int cn = n > ctx->guard_cap_n ? n : ctx->guard_cap_n, cl = nl > ctx->guard_cap_nl ? nl : ctx->guard_cap_nl;
    const size_t ints = ((size_t)cn * 4 + 255) / 256 * 256;
    ctx->guard_slot_bytes = 3 * ints + 256 + ((size_t)cn * cl * 4 + 255) / 256 * 256;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->guard_dev, 2 * ctx->guard_slot_bytes));
    HIP_TRY(ctx, hipMemset(ctx->guard_dev, 0, 2 * ctx->guard_slot_bytes));   // (the selection's ticket words must start at zero)
    ctx->guard_cap_n = cn; ctx->guard_cap_nl = cl;
  }
  if (!ctx->guard_host) HIP_TRY(ctx, hipHostMalloc((void **)&ctx->guard_host, 64, hipHostMallocDefault));
  const size_t ints = ((size_t)ctx->guard_cap_n * 4 + 255) / 256 * 256;
  char *base = ctx->guard_dev + (size_t)which * ctx->guard_slot_bytes;
  g->d_flat = (int32_t *)base; g->d_idx = (int32_t *)(base + ints); g->d_count = (int32_t *)(base + 2 * ints);
  g->phase = &ctx->guard_phase[which];   // (the pair d_count[0 .. 1] sits inside the 256 bytes reserved for the count)
  g->d_lg = (float *)(base + 2 * ints + 256);
  g->d_mag = (float *)(base + 2 * ints + 256 + ((size_t)ctx->guard_cap_n * ctx->guard_cap_nl * 4 + 255) / 256 * 256);
  g->h_count = ctx->guard_host + which;
  return MLT_OK;
}

struct Planes {  // the two Pel planes of a batch in device memory (element strides)
  const int16_t *org, *pred;
  long org_rs, org_cs, pred_rs, pred_cs;
  bool aligned8() const { return (((uintptr_t)org | (uintptr_t)pred) & 7) == 0 && ((org_rs | org_cs | pred_rs | pred_cs) & 3) == 0; }
};

int run_main(mlt_ctx *ctx, SizeState &st, int n, const int16_t *d_org, long org_rs, long org_cs, const int16_t *d_pred, long pred_rs, long pred_cs,
             const int32_t *d_poc, const int32_t *d_qp, int32_t *d_split, float *d_logits, int32_t *d_flat = nullptr, const GuardTail *tail = nullptr,
             bool flat_is_clear = false, float *d_mag = nullptr) {
  // (hi+lo-weights tiers: the two-plane model in the stages of w2_mask, single pass in the others)
  return run_network(ctx, st, st.model, n, d_org, org_rs, org_cs, d_pred, pred_rs, pred_cs, d_poc, d_qp, d_split, d_logits, d_flat,
                     st.w2 ? &st.model_w2 : nullptr, st.w2 ? st.w2_units : 0u, tail, flat_is_clear, st.x_units ? &st.model_exact : nullptr, st.x_units, d_mag);
}

// fast network + guard selection for n CUs, everything asynchronous on ctx->stream; the count lands in g.h_count
// (pinned) -- valid after the stream has been synchronised.  d_logits may be NULL.
int run_guarded_async(mlt_ctx *ctx, SizeState &st, int n, const Planes &pl, const int32_t *d_poc, const int32_t *d_qp, int32_t *d_split,
                      float *d_logits, const GuardSlot &g) {
  const int S = st.size, nl = st.model.n_logits;
  Launch L{ctx};
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int rc;
  float *lg = d_logits ? d_logits : (st.margin_guard ? g.d_lg : nullptr);
  if (g.single && n == 1) {
    // one CU (mlt_predict's captured graph): the selection is a tail of the heads kernel -- no guard_select launch, no memset of the
    // statistic (the tail clears it for the next call; it is only consumed when the first kernel is the one that produces it: aligned planes,
    // S >= 64 -- else flat_stat_kernel overwrites it), no separate copy of the count (the caller's result copy carries it)
    const GuardTail tail{g.d_count, g.d_idx, st.flat_guard ? g.d_flat : nullptr, (S * S / 4) / st.flat_div, (S * S / 4) / 2, st.margin_guard ? st.guard_margin : 0.f, st.mag_thr, nullptr};
    return run_main(ctx, st, 1, pl.org, pl.org_rs, pl.org_cs, pl.pred, pl.pred_rs, pl.pred_cs, d_poc, d_qp, d_split, lg, st.flat_guard ? g.d_flat : nullptr, &tail, true);
  }
  if (!ctx->guard_select_kernel && g.phase) {
    // round 6: the selection is a tail of the heads kernel for batches as well -- an unordered list of the flagged CUs through an atomic append on a counter that is
    // zero on entry; the launch zeroes the slot's OTHER counter for the next one (whose predecessor's count has left the device by then: same stream).  One launch and
    // one launch gap less per step.  (A first version counted finished workgroups to let the last one publish and re-arm a single counter: 4096 same-address atomics
    // and release fences made the heads launch 0.123 ms instead of 0.030 -- more than the launch it saved.)
    int32_t *cnt = g.d_count + *g.phase, *next = g.d_count + (*g.phase ^ 1);
    const GuardTail tail{cnt, g.d_idx, st.flat_guard ? g.d_flat : nullptr, (S * S / 4) / st.flat_div, (S * S / 4) / 2, st.margin_guard ? st.guard_margin : 0.f, st.mag_thr, next};
    if ((rc = run_main(ctx, st, n, pl.org, pl.org_rs, pl.org_cs, pl.pred, pl.pred_rs, pl.pred_cs, d_poc, d_qp, d_split, lg,
                       st.flat_guard ? g.d_flat : nullptr, &tail))) {
      // a pass that failed (e.g. no memory for the workspace of an oversized batch -- the caller may come back with a smaller one) may or may not have run its heads
      // kernel: both counters back to zero, the phase stays -- whichever counter the next launch counts on is zero on entry
      (void)hipMemsetAsync(g.d_count, 0, 8, ctx->stream);
      return rc;
    }
    *g.phase ^= 1;
    HIP_TRY(ctx, hipMemcpyAsync(g.h_count, cnt, 4, hipMemcpyDeviceToHost, ctx->stream));
    return MLT_OK;
  }
  float *mg = st.mag_thr > 0.f ? g.d_mag : nullptr;
  if ((rc = run_main(ctx, st, n, pl.org, pl.org_rs, pl.org_cs, pl.pred, pl.pred_rs, pl.pred_cs, d_poc, d_qp, d_split, lg,
                     st.flat_guard ? g.d_flat : nullptr, nullptr, false, mg))) return rc;
  GuardSelectArgs sa{};
  sa.mag = mg; sa.mag_thr = st.mag_thr;
  sa.flat = st.flat_guard ? g.d_flat : nullptr;
  sa.logits = st.margin_guard ? lg : nullptr;
  sa.idx = g.d_idx; sa.count = g.d_count; sa.n = n; sa.n_logits = nl;
  int off = 0;
  for (int h = 0; h < st.head_index; ++h) off += st.model.heads[h].classes;
  sa.head_off = off; sa.head_classes = st.model.heads[st.head_index].classes;
  sa.flat_thr = (S * S / 4) / st.flat_div;  // >= 1/8 (exact-lite tier: 1/16) of the quads exactly flat (constant / exactly linear in both planes)
  sa.near_thr = (S * S / 4) / 2;  // or >= 1/2 of them near-flat (mlt_kernels.h: MLT_FLAT_RANGE)
  sa.margin = st.margin_guard ? st.guard_margin : 0.f;
  if ((rc = L.prof_begin("guard_select", 0.0, 0.0, e0, e1))) return rc;
  LAUNCH_TRY(ctx, mlt_launch_guard_select(sa, ctx->stream));
  if ((rc = L.prof_end(e1))) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(g.h_count, g.d_count, 4, hipMemcpyDeviceToHost, ctx->stream));
  return MLT_OK;
}

// k > 0 flagged CUs (g.d_idx) of a batch whose fast results are in d_split / d_logits: exact re-evaluation, asynchronous.
int guard_fixup_async(mlt_ctx *ctx, SizeState &st, int k, const Planes &pl, const int32_t *d_poc, const int32_t *d_qp, int32_t *d_split,
                      float *d_logits, const GuardSlot &g) {
  const int S = st.size, nl = st.model.n_logits;
  const size_t cs = (size_t)S * S;
  const size_t plane = (cs * 2 * k + 255) / 256 * 256, small = ((size_t)k * 4 + 255) / 256 * 256, lgb = ((size_t)k * nl * 4 + 255) / 256 * 256;
  const size_t need = 2 * plane + 3 * small + lgb;
  if (need > ctx->gstage_bytes) {
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->gstage) (void)hipFree(ctx->gstage);
    ctx->gstage = nullptr; ctx->gstage_bytes = 0;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->gstage, need));
    ctx->gstage_bytes = need;
  }
  GuardGatherArgs ga{};
  ga.org = pl.org; ga.pred = pl.pred; ga.org_row_stride = pl.org_rs; ga.org_cu_stride = pl.org_cs; ga.pred_row_stride = pl.pred_rs; ga.pred_cu_stride = pl.pred_cs;
  ga.poc = d_poc; ga.qp = d_qp; ga.idx = g.d_idx; ga.k = k; ga.s_l = ilog2(S);
  ga.g_org = (int16_t *)ctx->gstage; ga.g_pred = (int16_t *)(ctx->gstage + plane);
  ga.g_poc = (int32_t *)(ctx->gstage + 2 * plane); ga.g_qp = (int32_t *)(ctx->gstage + 2 * plane + small);
  int32_t *g_split = (int32_t *)(ctx->gstage + 2 * plane + 2 * small);
  float *g_lg = (float *)(ctx->gstage + 2 * plane + 3 * small);
  HIP_TRY(ctx, mlt_launch_guard_gather(ga, ctx->stream));
  int rc = run_network(ctx, st, st.model_exact, k, ga.g_org, S, (long)cs, ga.g_pred, S, (long)cs, ga.g_poc, ga.g_qp, g_split, g_lg);
  if (rc) return rc;
  GuardScatterArgs sc{};
  sc.idx = g.d_idx; sc.g_split = g_split; sc.g_logits = g_lg; sc.split = d_split; sc.logits = d_logits; sc.k = k; sc.n_logits = nl;
  HIP_TRY(ctx, mlt_launch_guard_scatter(sc, ctx->stream));
  st.reruns += (uint64_t)k;
  return MLT_OK;
}

// network for n CUs with whatever guards the size has; synchronises once when guards are on (see mlt_predict_batch_device).
int run_checked(mlt_ctx *ctx, SizeState &st, int n, const Planes &pl, const int32_t *d_poc, const int32_t *d_qp, int32_t *d_split, float *d_logits) {
  if (!st.guards())
    return run_main(ctx, st, n, pl.org, pl.org_rs, pl.org_cs, pl.pred, pl.pred_rs, pl.pred_cs, d_poc, d_qp, d_split, d_logits);
  GuardSlot g;
  int rc = guard_slot(ctx, 0, n, st.model.n_logits, &g);
  if (rc) return rc;
  if ((rc = run_guarded_async(ctx, st, n, pl, d_poc, d_qp, d_split, d_logits, g))) return rc;
  // Wait for the 4-byte count.  Default: SLEEP for most of the time the batch is expected to take (a running estimate per CU of this
  // size, learnt from the previous calls), then poll the event for the rest: the host core is idle for all but the last ~0.2 ms of a
  // 5 ms batch -- in the encoder host cores are the scarce resource -- and the caller's next batch is still enqueued the moment this one
  // is through (a plain blocking wait wakes up on an interrupt and left the GPU idle for ~0.14 ms per 4096-CU step, 2.8 %).
  // MLT_GUARD_SPIN_WAIT=1: poll from the start (one busy core); MLT_GUARD_BLOCKING_WAIT=1: hipEventSynchronize on a blocking-sync event.
  const int wait_mode = tuning().guard_wait_mode;
  if (!ctx->ev_guard) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_guard, hipEventDisableTiming | hipEventBlockingSync));
  HIP_TRY(ctx, hipEventRecord(ctx->ev_guard, ctx->stream));
  if (wait_mode == 2) HIP_TRY(ctx, hipEventSynchronize(ctx->ev_guard));
  else {
    const auto t0 = std::chrono::steady_clock::now();
    // Only chunks of >= 512 CUs are worth sleeping for (shorter ones are through in well under a millisecond: poll), and only on an
    // estimate measured on a chunk of the same power-of-two bucket, scaled by the ratio of the sizes.
    int b = 0;
    while (b < 15 && (2 << b) <= n) ++b;
    const double expect_us = (n >= 512 && st.guard_n[b] > 0) ? st.guard_us[b] * (double)n / (double)st.guard_n[b] : 0.0;
    const bool slept = wait_mode == 0 && expect_us > 400.0;
    if (slept) std::this_thread::sleep_for(std::chrono::microseconds((long)(expect_us - 250.0)));
    hipError_t e = hipEventQuery(ctx->ev_guard);
    const bool overslept = slept && e != hipErrorNotReady;  // through already on waking up: the true duration is unknown, only "shorter"
    while (e == hipErrorNotReady) {
#if defined(__x86_64__)
      for (int i = 0; i < 32; ++i) __builtin_ia32_pause();
#endif
      e = hipEventQuery(ctx->ev_guard);
    }
    HIP_TRY(ctx, e);
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    // The estimate must never stay above the truth (oversleeping costs GPU time, polling only host time): a wake-up that found the batch
    // finished FORGETS the bucket's estimate -- the next chunk of that size is polled from the start and measured exactly -- and a
    // measured duration (we polled, so `us` is exact) replaces the estimate at once when it is shorter, half-way when it is longer.
    if (overslept) { st.guard_n[b] = 0; st.guard_us[b] = 0.0; }
    else if (n >= 512) {
      const double scaled = st.guard_n[b] > 0 ? st.guard_us[b] * (double)n / (double)st.guard_n[b] : 0.0;
      st.guard_us[b] = (st.guard_n[b] == 0 || us < scaled) ? us : 0.5 * (scaled + us);
      st.guard_n[b] = n;
    }
  }
  const int k = *g.h_count;
  if (k < 0 || k > n) { ctx->err = "guard: bad flagged-CU count"; return MLT_ERR_HIP; }
  return k ? guard_fixup_async(ctx, st, k, pl, d_poc, d_qp, d_split, d_logits, g) : MLT_OK;
}

// ---- load-time calibration of the fast arithmetic against the exact one (include/mltcnn.h: mlt_load_weights) ----
// Calibration set (round 3): NOT only the bench's texture distribution.  Seeded CUs in five content classes -- the classes the
// flat-content guard does NOT re-evaluate exactly, because admission must be decided on what the fast arithmetic will really see:
//   0 texture (blocky base + texture +-48, pred = org + noise +-40)      1 i.i.d. uniform org and pred (large residuals)
//   2 constant org / textured pred    3 textured org / constant pred   4 texture with a constant band over 10-12 % of the
//   quads, just under the guard's 1/8 for exactly flat quads   5 (round 4) texture with a NEAR-flat band (+-1 LSB dither or amplitude-4
//   texture on constants, alternating) over 40-48 % of the rows, just under the guard's 1/2 for near-flat quads
// (content the guard catches -- constant, dithered, low-contrast, ramps -- is evaluated with the exact arithmetic anyway).
// Round 4: 560 CUs (160 + 5 x 80; round 3: 96) = 5040 logits of the 128 model, so that the LARGEST error seen is a statistic with some power:
// a Gaussian sample of that size peaks at 3.9 sigma, the round-3 tail probe found weight sets whose worst error sits at 6.5 x their rms.
constexpr int kCalibClasses = 6;          // synthetic content classes; class kCalibClasses = the caller's own CUs (mlt_calibrate)
constexpr int kCalibCount[kCalibClasses] = {160, 80, 80, 80, 80, 80};
constexpr int kCalibN = 560;
constexpr int kCalibCallerMax = 4096;     // caller-supplied CUs per mlt_calibrate call (64 KiB of planes each at S = 128)

struct CalibInputs { std::vector<int16_t> org, pred; std::vector<int32_t> poc, qp; std::vector<int> cls; };

// (generated once per process and CU size: 31 MB of planes for S = 128, ~0.1 s of host time)
const CalibInputs &calibration_set(int S) {
  static std::mutex mu;
  static std::map<int, CalibInputs> cache;
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(S);
  if (it != cache.end()) return it->second;
  CalibInputs &ci = cache[S];
  const size_t cs = (size_t)S * S;
  ci.org.assign(cs * kCalibN, 0); ci.pred.assign(cs * kCalibN, 0); ci.poc.assign(kCalibN, 0); ci.qp.assign(kCalibN, 0); ci.cls.assign(kCalibN, 0);
  uint64_t z = 0x9E3779B97F4A7C15ull;  // splitmix64
  auto next = [&]() { z += 0x9E3779B97F4A7C15ull; uint64_t x = z; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); };
  auto clip = [](int v) { return v < 0 ? 0 : v > 1023 ? 1023 : v; };
  const int nb = S / 16 > 0 ? S / 16 : 1, bs = S / nb;
  int i = 0;
  for (int c = 0; c < kCalibClasses; ++c)
    for (int k = 0; k < kCalibCount[c]; ++k, ++i) {
      ci.cls[i] = c;
      int16_t *o = &ci.org[(size_t)i * cs], *q = &ci.pred[(size_t)i * cs];
      std::vector<int> base((size_t)nb * nb);
      for (int &b : base) b = 64 + (int)(next() % 896);
      const int co = (int)(next() % 1024), cp = (int)(next() % 1024);
      int band_h = (S * (10 + k % 3)) / 100;
      if (band_h * 8 >= S) band_h = S / 8 - 1;
      if (band_h < 1) band_h = 1;
      if (c == 5) band_h = (S * (40 + 4 * (k % 3))) / 100;
      const int band_y = (int)(next() % (uint64_t)(S - band_h + 1));
      const int amp = (k & 1) ? 4 : 1;  // class 5: low contrast / dither
      for (int y = 0; y < S; ++y)
        for (int x = 0; x < S; ++x) {
          int vo, vp;
          if (c == 1) { vo = (int)(next() % 1024); vp = (int)(next() % 1024); }
          else {
            vo = clip(base[(size_t)(y / bs) * nb + x / bs] + (int)(next() % 97) - 48);
            vp = clip(vo + (int)(next() % 81) - 40);
            if (c == 2) vo = co;
            if (c == 3) vp = cp;
            if (c == 4 && y >= band_y && y < band_y + band_h) { vo = co; vp = cp; }
            if (c == 5 && y >= band_y && y < band_y + band_h) {
              vo = clip(8 + co % 1008 + (int)(next() % (uint64_t)(2 * amp + 1)) - amp);
              vp = clip(8 + cp % 1008 + (int)(next() % (uint64_t)(2 * amp + 1)) - amp);
            }
          }
          o[(size_t)y * S + x] = (int16_t)vo;
          q[(size_t)y * S + x] = (int16_t)vp;
        }
      ci.poc[i] = (int32_t)(next() % 601);
      ci.qp[i] = 17 + (int32_t)(next() % 31);
    }
  return ci;
}

// Round 6: the IN-DISTRIBUTION set behind the magnitude guard.  A configuration admitted behind that guard runs only CUs of ordinary logit
// magnitude in the non-exact arithmetic -- for a trained-like weight set that leaves ~200 of the 560 synthetic CUs (the texture class and parts of
// the band classes), too few for the statistical admission rule, and none of them has the statistics of natural scenes, the content on which
// such a set's WEIGHT rounding error is largest (smooth activations: the error of a weight is the same at every pixel and survives the pooling;
// tools/attribute_error.py).  So the guarded figures are taken over the synthetic CUs below the threshold PLUS this set: 160 further CUs of the
// texture class (class 0) and 160 "1/f scenes" (class kClassScenes): a random-phase field of 48 plane waves with log-uniform spatial frequency
// (equal power per octave = the 1/f^2 power law of natural images: fastintercu-vvc_amd/synth.py natural_patches, without the FFT), contrast
// log-uniform 6 ... 160 ten-bit steps around a mean of 120 ... 900, +-1 step of sensor noise; prediction = the scene displaced by a motion vector
// in [-2, 2]^2, smoothed by [1 2 1]^2 / 16 with probability 1/2, + noise of amplitude 0 ... 6.  Only priced for configurations the plain rule
// rejects; generated once per process (~0.2 s).
constexpr float kMagRange = 1.5f;   // range guard: a plain-admitted tier is trusted up to this multiple of the largest logit magnitude of its calibration CUs
constexpr int kClassScenes = kCalibClasses + 1;   // content class ids: 0 .. 5 synthetic, kCalibClasses = the caller's, kClassScenes = the 1/f scenes
constexpr int kCalibExtraTexture = 160, kCalibExtraScenes = 160, kCalibExtraN = kCalibExtraTexture + kCalibExtraScenes;

const CalibInputs &calibration_extra_set(int S) {
  static std::mutex mu;
  static std::map<int, CalibInputs> cache;
  std::lock_guard<std::mutex> lock(mu);
  auto it = cache.find(S);
  if (it != cache.end()) return it->second;
  CalibInputs &ci = cache[S];
  const size_t cs = (size_t)S * S;
  ci.org.assign(cs * kCalibExtraN, 0); ci.pred.assign(cs * kCalibExtraN, 0); ci.poc.assign(kCalibExtraN, 0); ci.qp.assign(kCalibExtraN, 0); ci.cls.assign(kCalibExtraN, 0);
  uint64_t z = 0xD1B54A32D192ED03ull;  // splitmix64, another stream than calibration_set's
  auto next = [&]() { z += 0x9E3779B97F4A7C15ull; uint64_t x = z; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); };
  auto unif = [&]() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); };
  auto clip = [](int v) { return v < 0 ? 0 : v > 1023 ? 1023 : v; };
  const int nb = S / 16 > 0 ? S / 16 : 1, bs = S / nb;
  for (int i = 0; i < kCalibExtraTexture; ++i) {   // the texture class of calibration_set (class 0)
    int16_t *o = &ci.org[(size_t)i * cs], *q = &ci.pred[(size_t)i * cs];
    std::vector<int> base((size_t)nb * nb);
    for (int &b : base) b = 64 + (int)(next() % 896);
    for (int y = 0; y < S; ++y)
      for (int x = 0; x < S; ++x) {
        const int vo = clip(base[(size_t)(y / bs) * nb + x / bs] + (int)(next() % 97) - 48);
        o[(size_t)y * S + x] = (int16_t)vo;
        q[(size_t)y * S + x] = (int16_t)clip(vo + (int)(next() % 81) - 40);
      }
    ci.poc[i] = (int32_t)(next() % 601);
    ci.qp[i] = 17 + (int32_t)(next() % 31);
  }
  const int m = S + 8, K = 48;
  std::vector<float> field((size_t)m * m);
  std::vector<int> scene((size_t)m * m), ref((size_t)m * m);
  for (int i = kCalibExtraTexture; i < kCalibExtraN; ++i) {
    ci.cls[i] = kClassScenes;
    std::fill(field.begin(), field.end(), 0.f);
    for (int k = 0; k < K; ++k) {
      const double f = std::exp(std::log(1.0 / m) + unif() * (std::log(0.5) - std::log(1.0 / m)));   // cycles per pixel, log-uniform in [1 / m, 1 / 2]
      const double th = unif() * 6.283185307179586, ph = unif() * 6.283185307179586;
      const double wx = 6.283185307179586 * f * std::cos(th), wy = 6.283185307179586 * f * std::sin(th);
      const double cb = std::cos(wx), sb = std::sin(wx);
      for (int y = 0; y < m; ++y) {   // cos(ph + wy y + wx x) along x by rotation
        double c = std::cos(ph + wy * y), sn = std::sin(ph + wy * y);
        float *row = &field[(size_t)y * m];
        for (int x = 0; x < m; ++x) { row[x] += (float)c; const double c2 = c * cb - sn * sb; sn = sn * cb + c * sb; c = c2; }
      }
    }
    double mean = 0.0, var = 0.0;
    for (float v : field) mean += v;
    mean /= (double)field.size();
    for (float v : field) var += (v - mean) * (v - mean);
    const double sd_f = std::sqrt(var / (double)field.size()) + 1e-12;
    const double u0 = unif(), u1 = unif(), u2 = unif(), u3 = unif();
    const double sd = 6.0 * std::exp(u0 * std::log(160.0 / 6.0)), mu_s = 120.0 + 780.0 * u1;
    for (size_t j = 0; j < field.size(); ++j) scene[j] = clip((int)std::lrint(mu_s + sd * (field[j] - mean) / sd_f) + (int)(next() % 3) - 1);
    ref = scene;
    if (u3 < 0.5)
      for (int y = 1; y < m - 1; ++y)
        for (int x = 1; x < m - 1; ++x) {
          const int *r0 = &scene[(size_t)(y - 1) * m + x], *r1 = r0 + m, *r2 = r1 + m;
          ref[(size_t)y * m + x] = (r0[-1] + 2 * r0[0] + r0[1] + 2 * r1[-1] + 4 * r1[0] + 2 * r1[1] + r2[-1] + 2 * r2[0] + r2[1] + 8) / 16;
        }
    const int my = (int)(next() % 5) - 2, mx = (int)(next() % 5) - 2, a = (int)(u2 * 7.0);
    int16_t *o = &ci.org[(size_t)i * cs], *q = &ci.pred[(size_t)i * cs];
    for (int y = 0; y < S; ++y)
      for (int x = 0; x < S; ++x) {
        o[(size_t)y * S + x] = (int16_t)scene[(size_t)(y + 4) * m + x + 4];
        q[(size_t)y * S + x] = (int16_t)clip(ref[(size_t)(y + 4 + my) * m + x + 4 + mx] + (a ? (int)(next() % (uint64_t)(2 * a + 1)) - a : 0));
      }
    ci.poc[i] = (int32_t)(next() % 601);
    ci.qp[i] = 17 + (int32_t)(next() % 31);
  }
  return ci;
}

// The caller's own content for the calibration (mlt_calibrate): n dense CUs in HOST memory, appended to the synthetic set or replacing it.
struct CalibExtra { const int16_t *org, *pred; const int32_t *poc, *qp; int n; bool replace; };

// One calibration session: the calibration CUs resident on the device (the synthetic set, the caller's CUs, or both), their exact logits
// (computed ONCE, 96 CUs at a time: the exact workspace is 5.6 MiB per 128x128 CU), and price(w2 units, exact units) = the set through
// `model` with hi+lo weights / the exact arithmetic in those launch units (0: single pass everywhere), leaving in st.calib_rms the WORST
// pooled rms |dlogit| over {each content class, each head}, in st.calib_max the overall maximum and in tail_ratio max / (rms pooled over
// everything).  Caller CUs the flat-content guard would re-evaluate exactly anyway (same statistic, same thresholds) do not count: the
// admission is about what the non-exact arithmetic will really see.
struct CalibSession {
  mlt_ctx *ctx; SizeState &st;
  const CalibExtra *extra;
  // one resident set of CUs: the synthetic calibration set (+ the caller's), or the in-distribution set behind the magnitude guard
  struct Set {
    int n = 0;
    char *d = nullptr;
    int16_t *d_org = nullptr, *d_pred = nullptr;
    int32_t *d_poc = nullptr, *d_qp = nullptr, *d_split = nullptr;
    float *d_lg = nullptr, *d_mag = nullptr;
    std::vector<int> cls;
    std::vector<char> use;
    std::vector<float> le, lf, mag;   // exact logits, the candidate's logits, logit magnitude (HeadArgs.mag of the exact pass)
  };
  Set main, xtra;
  int n = 0, n_syn = 0, n_used = 0, n_caller_used = 0;
  float tail_ratio = 0.f;
  bool want_mag = false;   // the size may run behind the magnitude guard: the exact pass also delivers the magnitudes
  static constexpr int kSub = 96;
  CalibSession(mlt_ctx *c, SizeState &s, const CalibExtra *e = nullptr) : ctx(c), st(s), extra(e) {}
  ~CalibSession() {
    if (main.d) (void)hipFree(main.d);
    if (xtra.d) (void)hipFree(xtra.d);
    // the workspace grew to 96 exact CUs (540 MiB at S = 128): release it, the first real call sizes it for its own batch (a max_batch = 1
    // encoder context would otherwise carry it for life); captured graphs of every size that baked the old workspace in are dropped
    // with it (a later allocation may return the same address with fewer bytes behind it)
    (void)hipStreamSynchronize(ctx->stream);
    release_ws(ctx);
  }
  int alloc(Set &t, int count) {
    const int S = st.size, nl = st.model.n_logits;
    const size_t cs = (size_t)S * S, plane = cs * 2 * (size_t)count;
    t.n = count;
    HIP_TRY(ctx, hipMalloc((void **)&t.d, 2 * plane + 4 * (size_t)count * 4 + (size_t)count * nl * 4));
    t.d_org = (int16_t *)t.d; t.d_pred = (int16_t *)(t.d + plane);
    t.d_poc = (int32_t *)(t.d + 2 * plane); t.d_qp = t.d_poc + count; t.d_split = t.d_qp + count;
    t.d_mag = (float *)(t.d_split + count);
    t.d_lg = t.d_mag + count;
    return MLT_OK;
  }
  // which CUs of t[first ..) the flat-content guard re-evaluates exactly at run time (flat_stat_kernel's statistic, guard_select_kernel's
  // thresholds): those never see the arithmetic being priced
  int drop_flat(Set &t, int first, int div = 8, std::vector<char> *mask = nullptr) {
    const int S = st.size, cnt = t.n - first;
    const size_t cs = (size_t)S * S;
    if (!st.cfg_flat_guard || cnt <= 0) return MLT_OK;
    FlatStatArgs fa{};
    fa.org = t.d_org + cs * first; fa.pred = t.d_pred + cs * first; fa.org_row_stride = S; fa.org_cu_stride = (long)cs; fa.pred_row_stride = S;
    fa.pred_cu_stride = (long)cs; fa.flat = t.d_split; fa.n = cnt; fa.s_l = ilog2(S);
    HIP_TRY(ctx, mlt_launch_flat_stat(fa, true, ctx->stream));
    std::vector<int32_t> fl((size_t)cnt);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(fl.data(), t.d_split, (size_t)cnt * 4, hipMemcpyDeviceToHost));
    const int flat_thr = (S * S / 4) / div, near_thr = (S * S / 4) / 2;
    std::vector<char> &u = mask ? *mask : t.use;
    for (int i = 0; i < cnt; ++i)
      if ((fl[(size_t)i] >> MLT_FLAT_EXACT_SHIFT) >= flat_thr || (fl[(size_t)i] & 0xFFFF) >= near_thr) u[(size_t)(first + i)] = 0;
    return MLT_OK;
  }
  int run(Set &t, std::vector<float> &out, bool exact, unsigned mask, unsigned xmask = 0, mlt::Model *whole = nullptr, bool with_mag = false) {
    const int S = st.size, nl = st.model.n_logits;
    const long cs = (long)S * S;
    const bool prof = ctx->profile;
    ctx->profile = false;
    int rc = MLT_OK;
    for (int i0 = 0; i0 < t.n && rc == MLT_OK; i0 += kSub) {
      const int c = t.n - i0 < kSub ? t.n - i0 : kSub;
      float *mg = with_mag ? t.d_mag + i0 : nullptr;
      rc = whole ? run_network(ctx, st, *whole, c, t.d_org + (size_t)i0 * cs, S, cs, t.d_pred + (size_t)i0 * cs, S, cs, t.d_poc + i0, t.d_qp + i0, t.d_split, t.d_lg + (size_t)i0 * nl)
         : exact ? run_network(ctx, st, st.model_exact, c, t.d_org + (size_t)i0 * cs, S, cs, t.d_pred + (size_t)i0 * cs, S, cs, t.d_poc + i0, t.d_qp + i0, t.d_split, t.d_lg + (size_t)i0 * nl,
                               nullptr, nullptr, 0, nullptr, false, nullptr, 0, mg)
                 : run_network(ctx, st, st.model, c, t.d_org + (size_t)i0 * cs, S, cs, t.d_pred + (size_t)i0 * cs, S, cs, t.d_poc + i0, t.d_qp + i0, t.d_split, t.d_lg + (size_t)i0 * nl,
                               nullptr, mask ? &st.model_w2 : nullptr, mask, nullptr, false, xmask ? &st.model_exact : nullptr, xmask);
    }
    ctx->profile = prof;
    if (rc) return rc;
    out.resize((size_t)t.n * nl);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(out.data(), t.d_lg, out.size() * 4, hipMemcpyDeviceToHost));
    if (with_mag) {
      t.mag.resize((size_t)t.n);
      HIP_TRY(ctx, hipMemcpy(t.mag.data(), t.d_mag, (size_t)t.n * 4, hipMemcpyDeviceToHost));
    }
    return MLT_OK;
  }
  // MLT_CALIB_REPLACE needs enough of the caller's CUs to carry the statistical admission rule on their own (it assumes thousands of logits:
  // the synthetic set has 560 CUs): when fewer than kCalibMinReplace of them are left after dropping those the flat-content guard re-evaluates
  // exactly anyway -- all of them flat, or a tiny n -- the synthetic set is kept and the caller's CUs are APPENDED to it instead (visible to the
  // caller as mlt_arith_info.calib_cus > calib_caller_cus).  No tier is ever admitted on an empty or near-empty set.
  static constexpr int kCalibMinReplace = 256;
  int begin() {
    int rc = stage_set(extra && extra->replace);
    if (rc == MLT_OK && extra && extra->replace && n_used < kCalibMinReplace) {
      (void)hipFree(main.d);
      main = Set();
      rc = stage_set(false);
    }
    return rc ? rc : run(main, main.le, true, 0, 0, nullptr, want_mag);
  }
  int stage_set(bool replace) {
    const int S = st.size;
    const size_t cs = (size_t)S * S;
    const CalibInputs *syn = replace ? nullptr : &calibration_set(S);
    n_syn = syn ? kCalibN : 0;
    const int n_ex = extra ? extra->n : 0;
    n = n_syn + n_ex;
    int rc = alloc(main, n);
    if (rc) return rc;
    main.cls.assign((size_t)n, kCalibClasses);
    main.use.assign((size_t)n, 1);
    if (syn) std::copy(syn->cls.begin(), syn->cls.end(), main.cls.begin());
    if (syn) {
      HIP_TRY(ctx, hipMemcpy(main.d_org, syn->org.data(), cs * 2 * kCalibN, hipMemcpyHostToDevice));
      HIP_TRY(ctx, hipMemcpy(main.d_pred, syn->pred.data(), cs * 2 * kCalibN, hipMemcpyHostToDevice));
      HIP_TRY(ctx, hipMemcpy(main.d_poc, syn->poc.data(), (size_t)kCalibN * 4, hipMemcpyHostToDevice));
      HIP_TRY(ctx, hipMemcpy(main.d_qp, syn->qp.data(), (size_t)kCalibN * 4, hipMemcpyHostToDevice));
    }
    if (n_ex) {
      HIP_TRY(ctx, hipMemcpy(main.d_org + cs * n_syn, extra->org, cs * 2 * (size_t)n_ex, hipMemcpyHostToDevice));
      HIP_TRY(ctx, hipMemcpy(main.d_pred + cs * n_syn, extra->pred, cs * 2 * (size_t)n_ex, hipMemcpyHostToDevice));
      HIP_TRY(ctx, hipMemcpy(main.d_poc + n_syn, extra->poc, (size_t)n_ex * 4, hipMemcpyHostToDevice));
      HIP_TRY(ctx, hipMemcpy(main.d_qp + n_syn, extra->qp, (size_t)n_ex * 4, hipMemcpyHostToDevice));
      if ((rc = drop_flat(main, n_syn))) return rc;
    }
    n_used = 0; n_caller_used = 0;
    for (int i = 0; i < n; ++i) { n_used += main.use[(size_t)i]; if (i >= n_syn) n_caller_used += main.use[(size_t)i]; }
    return MLT_OK;
  }
  struct TierPrice_ { float rms = 0.f, max = 0.f, tail = 0.f; double cls_rms[kCalibClasses + 2] = {0}, head_rms[4] = {0}; };
  // pooled figures of (candidate - exact) over the CUs of `sets` that count and whose magnitude is <= thr (thr <= 0: all of them)
  void pool(const Set *const *sets, int n_sets, float thr, TierPrice_ &out, int *n_kept = nullptr, const std::vector<char> *const *masks = nullptr) {
    const int nl = st.model.n_logits;
    double mx = 0.0, s2_all = 0.0;
    double s2_cls[kCalibClasses + 2] = {0}, s2_head[4] = {0};
    size_t n_cls[kCalibClasses + 2] = {0}, n_head[4] = {0}, n_all = 0;
    int kept = 0;
    for (int t = 0; t < n_sets; ++t) {
      const Set &T = *sets[t];
      const std::vector<char> &use = masks ? *masks[t] : T.use;
      for (int i = 0; i < T.n; ++i) {
        if (!use[(size_t)i]) continue;
        if (thr > 0.f && !(T.mag[(size_t)i] <= thr)) continue;
        ++kept;
        int lo = 0;
        for (int h = 0; h < st.model.n_heads; ++h) {
          for (int k = 0; k < st.model.heads[h].classes; ++k) {
            const size_t j = (size_t)i * nl + lo + k;
            const double e = std::fabs((double)T.lf[j] - (double)T.le[j]);
            if (!(e <= mx)) mx = e;  // NaN -> mx = NaN -> fails the admission test
            s2_cls[T.cls[(size_t)i]] += e * e; ++n_cls[T.cls[(size_t)i]];
            s2_head[h] += e * e; ++n_head[h];
            s2_all += e * e; ++n_all;
          }
          lo += st.model.heads[h].classes;
        }
      }
    }
    double worst = 0.0;
    for (int c = 0; c < kCalibClasses + 2; ++c) if (n_cls[c]) { const double r = std::sqrt(s2_cls[c] / (double)n_cls[c]); if (!(r <= worst)) worst = r; }
    for (int h = 0; h < st.model.n_heads; ++h) if (n_head[h]) { const double r = std::sqrt(s2_head[h] / (double)n_head[h]); if (!(r <= worst)) worst = r; }
    const double rms_all = n_all ? std::sqrt(s2_all / (double)n_all) : 0.0;
    out.rms = (float)worst; out.max = (float)mx; out.tail = (float)(rms_all > 0.0 ? mx / rms_all : 0.0);
    for (int c = 0; c < kCalibClasses + 2; ++c) out.cls_rms[c] = std::sqrt(s2_cls[c] / (double)(n_cls[c] ? n_cls[c] : 1));
    for (int h = 0; h < 4; ++h) out.head_rms[h] = std::sqrt(s2_head[h] / (double)(n_head[h] ? n_head[h] : 1));
    if (n_kept) *n_kept = kept;
  }
  int price(unsigned mask, unsigned xmask = 0, mlt::Model *whole = nullptr) {
    int rc = run(main, main.lf, false, mask, xmask, whole);
    if (rc) return rc;
    TierPrice_ P;
    const Set *sets[1] = {&main};
    std::vector<char> keep_use;
    if (whole && st.cfg_flat_guard) {  // the exact-lite tier runs behind the flat guard at 1 / 16 (SizeState.flat_div): the CUs THAT guard re-evaluates do not count
      keep_use = main.use;
      if ((rc = drop_flat(main, 0, 16))) return rc;
    }
    pool(sets, 1, 0.f, P);
    if (!keep_use.empty()) main.use = keep_use;
    tail_ratio = P.tail;
    if (std::getenv("MLT_CALIB_VERBOSE")) {  // diagnostics: which content class / head decides the admission
      std::fprintf(stderr, "mltcnn calibration (size %d, %d CUs of which %d the caller's, hi+lo weights in units 0x%x, exact in units 0x%x): rms per class", st.size, n_used, n_caller_used, mask, xmask);
      for (int c = 0; c <= kCalibClasses; ++c) std::fprintf(stderr, " %.3e", P.cls_rms[c]);
      std::fprintf(stderr, " | per head");
      for (int h = 0; h < st.model.n_heads; ++h) std::fprintf(stderr, " %.3e", P.head_rms[h]);
      std::fprintf(stderr, " | max %.3e = %.1f x rms\n", (double)P.max, (double)tail_ratio);
    }
    st.calibrated = true;
    st.calib_rms = P.rms;
    st.calib_max = P.max;
    return MLT_OK;
  }
  // The configuration price() has just measured, behind the MAGNITUDE guard.  Threshold: the LARGEST magnitude T on a quarter-octave grid (from the
  // largest magnitude in the sets downwards) such that the CUs with M <= T -- main set + the in-distribution set, which is staged, and its exact
  // logits computed, on first use -- meet the REFINED admission rule (mlt_tier_search.h: k x rms <= 0.95 x and max <= 0.6 x tolerance: the rule for
  // choices made on the calibration data itself) with at least kGuardMinKept CUs left and at most flag_max of the in-distribution CUs above T.
  // The rule is the plain one applied to the population that will really run the tier -- the logic of the flat-content guard ("content the guard
  // catches does not count") with the threshold found instead of fixed.  (A first form derived T from the worst RELATIVE error over all CUs,
  // T = 0.65 x tolerance / max(e / M): it charged ordinary content for the relative error of the constant-band classes -- 4 x the others' -- and
  // flagged 9 % of it where the kept CUs' largest error was a quarter of the limit: profiles/r06b_calib_trained1.txt.)
  static constexpr int kGuardMinKept = 256;
  int price_guarded(unsigned mask, unsigned xmask, const mlt::TierRules &R, mlt::TierPrice &out) {
    out.g_valid = false;
    if (!want_mag || main.mag.size() != (size_t)main.n) return MLT_OK;
    int rc;
    if (!xtra.d) {
      const CalibInputs &ci = calibration_extra_set(st.size);
      const size_t cs = (size_t)st.size * st.size;
      if ((rc = alloc(xtra, kCalibExtraN))) return rc;
      xtra.cls = ci.cls;
      xtra.use.assign((size_t)kCalibExtraN, 1);
      HIP_TRY(ctx, hipMemcpy(xtra.d_org, ci.org.data(), cs * 2 * kCalibExtraN, hipMemcpyHostToDevice));
      HIP_TRY(ctx, hipMemcpy(xtra.d_pred, ci.pred.data(), cs * 2 * kCalibExtraN, hipMemcpyHostToDevice));
      HIP_TRY(ctx, hipMemcpy(xtra.d_poc, ci.poc.data(), (size_t)kCalibExtraN * 4, hipMemcpyHostToDevice));
      HIP_TRY(ctx, hipMemcpy(xtra.d_qp, ci.qp.data(), (size_t)kCalibExtraN * 4, hipMemcpyHostToDevice));
      if ((rc = drop_flat(xtra, 0))) return rc;
      if ((rc = run(xtra, xtra.le, true, 0, 0, nullptr, true))) return rc;
    }
    if ((rc = run(xtra, xtra.lf, false, mask, xmask))) return rc;
    const Set *sets[2] = {&main, &xtra};
    // A tier behind the magnitude guard also runs the FLAT guard at 1 / 16 of the quads exactly flat instead of 1 / 8 (SizeState.flat_div): the weight
    // sets that need this guard are the ones whose errors grow with what they amplify, and a 10-12 % constant band -- just under 1 / 8 -- was the one class
    // whose deep tail left the contract behind the guard (5 of 331,776 probed logits of the first trained family at 1.0-1.35e-3, all in that class:
    // profiles/r06d_tail_probe_trained.txt; 80 such CUs in the calibration set do not see a 1-in-7000 event).  The CUs THAT guard takes do not count here.
    if (use16[0].empty()) {
      use16[0] = main.use; use16[1] = xtra.use;
      if ((rc = drop_flat(main, 0, 16, &use16[0]))) return rc;
      if ((rc = drop_flat(xtra, 0, 16, &use16[1]))) return rc;
    }
    const std::vector<char> *masks[2] = {&use16[0], &use16[1]};
    float m_hi = 0.f, m_lo = INFINITY;
    int in_dist = 0;
    for (int t = 0; t < 2; ++t) {
      const Set *T = sets[t];
      for (int i = 0; i < T->n; ++i) {
        if (!use16[t][(size_t)i]) continue;
        const float m = T->mag[(size_t)i];
        if (!(m > 0.f) || !std::isfinite(m)) return MLT_OK;   // (a NaN / zero magnitude: no guarded variant)
        if (m > m_hi) m_hi = m;
        if (m < m_lo) m_lo = m;
        const int c = T->cls[(size_t)i];
        if (c == 0 || c == kCalibClasses || c == kClassScenes) ++in_dist;
      }
    }
    if (!(m_hi > 0.f) || in_dist == 0) return MLT_OK;
    const bool verbose = std::getenv("MLT_CALIB_VERBOSE") != nullptr;
    for (float thr = m_hi * 0.840896415f; thr >= m_lo; thr *= 0.840896415f) {   // 2^(-1/4) per step; T = m_hi would be the plain rule again
      TierPrice_ P;
      int kept = 0;
      pool(sets, 2, thr, P, &kept, masks);
      if (kept < kGuardMinKept) break;
      mlt::TierPrice tp;
      tp.rms = P.rms; tp.max = P.max; tp.tail = P.tail;
      if (!R.within_refined(tp)) continue;
      // the guard's price on ordinary content: the in-distribution CUs (texture, 1/f scenes, the caller's own) it sends to the exact re-run
      int flagged = 0;
      for (int t = 0; t < 2; ++t) {
        const Set *T = sets[t];
        for (int i = 0; i < T->n; ++i) {
          const int c = T->cls[(size_t)i];
          if (use16[t][(size_t)i] && (c == 0 || c == kCalibClasses || c == kClassScenes) && !(T->mag[(size_t)i] <= thr)) ++flagged;
        }
      }
      out.g_valid = true;
      out.g_rms = P.rms; out.g_max = P.max; out.g_tail = P.tail; out.g_thr = thr; out.g_flag = (float)flagged / (float)in_dist;
      if (verbose) {
        std::fprintf(stderr, "mltcnn calibration, behind the magnitude guard (threshold %.3f of %.3f .. %.3f: %d CUs at or below it, %d of %d in-distribution CUs above): rms per class",
                     (double)thr, (double)m_lo, (double)m_hi, kept, flagged, in_dist);
        for (int c = 0; c < kCalibClasses + 2; ++c) std::fprintf(stderr, " %.3e", P.cls_rms[c]);
        std::fprintf(stderr, " | per head");
        for (int h = 0; h < st.model.n_heads; ++h) std::fprintf(stderr, " %.3e", P.head_rms[h]);
        std::fprintf(stderr, " | max %.3e = %.1f x rms\n", (double)P.max, (double)P.tail);
      }
      return MLT_OK;
    }
    if (verbose) std::fprintf(stderr, "mltcnn calibration, behind the magnitude guard: no threshold in %.3f .. %.3f meets the refined rule with >= %d CUs\n", (double)m_lo, (double)m_hi, kGuardMinKept);
    return MLT_OK;
  }
  std::vector<char> use16[2];   // main / xtra: the CUs that count behind the flat guard at 1 / 16
};

void drop_graphs(mlt_ctx *ctx, int si) {  // a captured kernel chain bakes in weight / workspace pointers
  SingleCu &sg = ctx->single[si];
  if (sg.exec) (void)hipGraphExecDestroy(sg.exec);
  if (sg.graph) (void)hipGraphDestroy(sg.graph);
  sg.exec = nullptr; sg.graph = nullptr;
}

// The pricer of the tier search (mlt_tier_search.h) on the device: a configuration = launch units in hi+lo weights / in the exact arithmetic +
// the realisation of the single-pass weights' rounding.  Models are built and uploaded lazily: another realisation replaces st.model
// (~50 ms each), the hi+lo-weights copy appears with the first candidate that needs it.
struct DevicePricer : mlt::TierPricer {
  mlt_ctx *ctx; SizeState &st; CalibSession &cal;
  const void *blob; size_t bytes; int size;
  int cur_rounding = 0;
  const mlt::TierRules *rules = nullptr;   // != NULL: configurations the plain rule rejects are also priced behind the magnitude guard
  DevicePricer(mlt_ctx *c, SizeState &s, CalibSession &cs, const void *b, size_t n, int sz) : ctx(c), st(s), cal(cs), blob(b), bytes(n), size(sz) {}
  int price(unsigned w2_units, unsigned x_units, int rounding, mlt::TierPrice &out) override {
    std::string err;
    int rc;
    if (rounding != cur_rounding) {
      mlt::Model mv;
      if (!mlt::build_model(blob, bytes, mlt::MLT_MODEL_FAST, size, mv, err, rounding)) { ctx->err = "weights (rounding " + std::to_string(rounding) + "): " + err; return MLT_ERR_WEIGHTS; }
      if ((rc = upload_model(ctx, mv))) { free_model(mv); return rc; }
      std::swap(st.model, mv);
      free_model(mv);
      cur_rounding = rounding;
    }
    if (w2_units && !st.model_w2.on_device) {
      mlt::Model mw;
      if (!mlt::build_model(blob, bytes, mlt::MLT_MODEL_W2, size, mw, err)) { ctx->err = "weights (hi+lo copy): " + err; return MLT_ERR_WEIGHTS; }
      st.model_w2 = std::move(mw);
      if ((rc = upload_model(ctx, st.model_w2))) return rc;
    }
    if ((rc = cal.price(w2_units, x_units))) return rc;
    out.rms = st.calib_rms; out.max = st.calib_max; out.tail = cal.tail_ratio;
    // (the refined rule is the stricter of the two: whatever the search is about to test, a configuration that fails it gets its guarded figures)
    if (rules && cal.want_mag && !rules->within_refined(out) && (rc = cal.price_guarded(w2_units, x_units, *rules, out))) return rc;
    return MLT_OK;
  }
  int price_lite(mlt::TierPrice &out) override {
    std::string err;
    int rc;
    if (!st.model_xl.on_device) {
      mlt::Model mx;
      if (!mlt::build_model(blob, bytes, mlt::MLT_MODEL_XLITE, size, mx, err)) { ctx->err = "weights (exact-lite copy): " + err; return MLT_ERR_WEIGHTS; }
      st.model_xl = std::move(mx);
      if ((rc = upload_model(ctx, st.model_xl))) return rc;
    }
    if ((rc = cal.price(0, 0, &st.model_xl))) return rc;
    out.rms = st.calib_rms; out.max = st.calib_max; out.tail = cal.tail_ratio;
    return MLT_OK;
  }
};

int env_int(const char *name) {  // tuning switch holding a number (MLT_TUNING=1 only); -1: not set
  const char *e = tuning_env(name);
  return e ? (int)std::strtol(e, nullptr, 0) : -1;
}

void unload_size(mlt_ctx *ctx, int si) {
  SizeState &st = ctx->sz[si];
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  drop_graphs(ctx, si);
  free_model(st.model); free_model(st.model_exact); free_model(st.model_w2); free_model(st.model_xl);
  st.model = mlt::Model(); st.model_exact = mlt::Model(); st.model_w2 = mlt::Model(); st.model_xl = mlt::Model();
  st.loaded = false;
  st.plans.clear();
}

// Load (or re-calibrate: `extra` = the caller's CUs) ONE device's copy of a size.  blob / bytes stay valid for the call.
int load_one(mlt_ctx *ctx, int size, const void *blob, size_t bytes, const CalibExtra *extra) {
  const int si = size_index(size);
  if (si < 0) { ctx->err = "unsupported CU size"; return MLT_ERR_ARG; }
  SizeState &st = ctx->sz[si];
  if (!st.enabled) { ctx->err = "CU size not enabled in size_mask"; return MLT_ERR_SIZE_DISABLED; }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return MLT_ERR_NO_DEVICE; }
  std::string err;
  mlt::Model m;
  const bool no_small_mix = tuning().no_small_mix;
  const bool small_mix = st.small_mix && !no_small_mix;   // (then: fast copy = `model`, exact copy = `model_exact`, the calibration picks the stages)
  if (!mlt::build_model(blob, bytes, (st.want_exact && !small_mix) ? (ctx->xlite ? mlt::MLT_MODEL_XLITE : mlt::MLT_MODEL_EXACT) : mlt::MLT_MODEL_FAST, size, m, err)) { ctx->err = "weights: " + err; return MLT_ERR_WEIGHTS; }
  if (m.arch != (size == 128 ? 0 : 1)) { ctx->err = "weights: blob arch does not match CU size"; return MLT_ERR_WEIGHTS; }
  if (st.head_index < 0 || st.head_index >= m.n_heads) { ctx->err = "head_index out of range"; return MLT_ERR_ARG; }
  // a reload replaces device buffers that captured graphs and in-flight work point to
  (void)hipStreamSynchronize(ctx->stream);
  drop_graphs(ctx, si);
  // (models own device buffers: whatever the state held -- loaded or left over from a failed load -- is released first, and every
  // error path below releases what it uploaded, so a failed reload leaves the size cleanly unloaded instead of leaking)
  free_model(st.model); free_model(st.model_exact); free_model(st.model_w2); free_model(st.model_xl);
  st.loaded = false;
  st.plans.clear();
  st.exact = st.want_exact && !small_mix;
  st.lite = false; st.flat_guard = st.cfg_flat_guard; st.flat_div = 8; st.guard_margin = ctx->guard_margin;
  st.w2 = false; st.w2_mask = 0; st.w2_units = 0; st.x_mask = 0; st.x_units = 0;
  st.mag_thr = 0.f; st.calib_rel = 0.f; st.mag_flag = 0.f; st.mag_kind = 0;
  st.calibrated = false; st.calib_rms = st.calib_max = 0.f;
  st.calib_cus = st.calib_caller_cus = 0;
  st.model = std::move(m);
  st.model_exact = mlt::Model();
  st.model_w2 = mlt::Model();
  st.model_xl = mlt::Model();
  auto fail = [&](int rc) {
    free_model(st.model); free_model(st.model_exact); free_model(st.model_w2); free_model(st.model_xl);
    st.model = mlt::Model(); st.model_exact = mlt::Model(); st.model_w2 = mlt::Model(); st.model_xl = mlt::Model();
    return rc;
  };
  int rc = upload_model(ctx, st.model);
  if (rc) return fail(rc);
  if (!st.exact && (st.flat_guard || st.margin_guard || st.calibrate || small_mix)) {
    mlt::Model me;
    if (!mlt::build_model(blob, bytes, mlt::MLT_MODEL_EXACT, size, me, err)) { ctx->err = "weights (exact copy): " + err; return fail(MLT_ERR_WEIGHTS); }
    st.model_exact = std::move(me);
    if ((rc = upload_model(ctx, st.model_exact))) return fail(rc);
    if (small_mix || st.calibrate) {
      // The search itself lives in mlt_tier_search.h (no HIP in it; unit-tested on the CPU with a stub pricer):
      //  128: single pass -> other realisations of the weights' rounding -> hi+lo weights in a subset of stages (cheapest first) -> some stages
      //       exact -> refinements at launch-unit granularity -> exact;
      //  64 / 32 / 16 (maps of 1 .. 32 pixels: their time is in the FIRST stages, their error in the LAST ones): the longest single-pass
      //       prefix, layer0 with hi+lo weights, half of layer0 -> exact.  Largest error held to 0.5 x tolerance (their tails are heavier:
      //       profiles/r04s_tail_probe_{64,32}.txt measured 1.5 .. 1.85 x the calibration set's largest error).
      CalibSession cal(ctx, st, extra);
      // (the magnitude guard serves the 128 model's tiers; the small models' search is over exact prefixes and is left as it was)
      const bool no_mag = tuning().no_mag_guard;
      cal.want_mag = !small_mix && st.cfg_mag_guard && !no_mag;
      if ((rc = cal.begin())) return fail(rc);
      DevicePricer pricer(ctx, st, cal, blob, bytes, size);
      mlt::TierRules rules;
      rules.tolerance = ctx->tolerance;
      rules.max_frac = small_mix ? 0.5f : 0.65f;
      if (cal.want_mag) pricer.rules = &rules;
      mlt::TierForce force;
      force.no_mag_guard = !cal.want_mag;
      force.rounding = env_int("MLT_ROUNDING"); force.w2_mask = env_int("MLT_W2_MASK"); force.x_mask = env_int("MLT_X_MASK");
      force.w2_units = env_int("MLT_W2_UNITS"); force.small_prefix = env_int("MLT_SMALL_PREFIX");
      force.no_roundings = tuning_env("MLT_NO_ROUNDINGS") != nullptr; force.no_w2 = tuning_env("MLT_NO_W2") != nullptr;
      force.no_xmix = tuning_env("MLT_NO_XMIX") != nullptr; force.no_w2_units = tuning_env("MLT_NO_W2_UNITS") != nullptr;
      force.no_x_units = tuning_env("MLT_NO_X_UNITS") != nullptr; force.no_lite = tuning_env("MLT_NO_LITE") != nullptr;
      mlt::TierChoice ch;
      rc = small_mix ? mlt::search_tier_small(pricer, rules, force, st.model.n_stages, ch) : mlt::search_tier_128(pricer, rules, force, mlt::MLT_N_ROUNDINGS, ch);
      if (rc) return fail(rc);
      st.calib_cus = cal.n_used; st.calib_caller_cus = cal.n_caller_used;
      st.calib_rms = ch.price.rms; st.calib_max = ch.price.max;
      if (ch.lite) {  // the exact-lite arithmetic everywhere: its model becomes `model`; the exact copy stays for the decision guard's near-ties
        free_model(st.model_w2); st.model_w2 = mlt::Model();
        free_model(st.model);
        st.model = std::move(st.model_xl);
        st.model_xl = mlt::Model();
        st.lite = true;
        st.flat_guard = st.cfg_flat_guard;   // (round 5 switched it off here; round 6: on, at 1 / 16 of the quads exactly flat)
        st.flat_div = 16;
        if (!ctx->guard_margin_configured) st.guard_margin = std::min(ctx->guard_margin, std::max(1e-4f, 3.f * 1.7f * st.calib_max));
        if (!st.margin_guard && !st.flat_guard) { free_model(st.model_exact); st.model_exact = mlt::Model(); }
      } else if (ch.exact) {  // run it exact
        free_model(st.model_w2); st.model_w2 = mlt::Model();
        free_model(st.model_xl); st.model_xl = mlt::Model();
        free_model(st.model);
        st.model = std::move(st.model_exact);
        st.model_exact = mlt::Model();
        st.exact = true;
      } else {
        free_model(st.model_xl); st.model_xl = mlt::Model();
        st.w2 = ch.w2;
        st.w2_units = ch.w2_units; st.x_units = ch.x_units;
        st.w2_mask = mlt::stages_of_units(ch.w2_units); st.x_mask = mlt::stages_of_units(ch.x_units);
        if (!st.w2) { free_model(st.model_w2); st.model_w2 = mlt::Model(); }
        st.mag_thr = ch.mag_thr; st.mag_flag = ch.mag_flag;   // > 0: the tier was admitted behind the magnitude guard
        if (st.mag_thr > 0.f) { st.flat_div = 16; st.mag_kind = 2; }   // ... and then runs the flat guard at 1 / 16 (CalibSession::price_guarded)
        st.calib_rel = ch.mag_thr > 0.f ? ch.price.max / ch.mag_thr : 0.f;
      }
      // the RANGE guard of every non-exact tier the plain rule admitted (SizeState.mag_kind == 1)
      if (!st.exact && st.mag_kind == 0 && cal.want_mag && cal.main.mag.size() == (size_t)cal.main.n) {
        float m_hi = 0.f;
        for (int i = 0; i < cal.main.n; ++i)
          if (cal.main.use[(size_t)i] && cal.main.mag[(size_t)i] > m_hi) m_hi = cal.main.mag[(size_t)i];
        if (m_hi > 0.f && std::isfinite(m_hi)) { st.mag_thr = kMagRange * m_hi; st.mag_kind = 1; }
      }
    }
  }
  st.loaded = true;
  return MLT_OK;
}

// every device of a context: load / re-calibrate, then make sure they all landed on the SAME arithmetic (the header promises results
// bit-identical to a one-device context); on any failure the size is unloaded everywhere (no mixed weight sets)
int load_all(mlt_ctx *ctx, int size, const void *blob, size_t bytes, const CalibExtra *extra) {
  const int si = size_index(size);
  if (si < 0) { ctx->err = "unsupported CU size"; return MLT_ERR_ARG; }
  int rc = load_one(ctx, size, blob, bytes, extra);
  for (size_t i = 0; i < ctx->peers.size() && rc == MLT_OK; ++i) {
    mlt_ctx *p = ctx->peers[i];
    rc = load_one(p, size, blob, bytes, extra);
    if (rc) ctx->err = "device " + std::to_string(p->device) + ": " + p->err;
    else {
      const SizeState &a = ctx->sz[si], &b = p->sz[si];
      if (a.exact != b.exact || a.lite != b.lite || a.w2 != b.w2 || a.w2_units != b.w2_units || a.x_units != b.x_units || a.model.rounding != b.model.rounding || a.mag_thr != b.mag_thr || a.mag_kind != b.mag_kind) {
        ctx->err = "device " + std::to_string(p->device) + " calibrated to a different arithmetic than device " + std::to_string(ctx->device);
        rc = MLT_ERR_WEIGHTS;
      }
    }
  }
  if (rc) {
    unload_size(ctx, si);
    for (mlt_ctx *p : ctx->peers) unload_size(p, si);
    ctx->sz[si].blob.clear();
  }
  return rc;
}

}  // namespace

extern "C" {
#pragma GCC visibility push(default)

int mlt_abi_version(void) { return MLT_ABI_VERSION; }

// Signature of the sources this binary was built from (fastintercu-vvc_amd/build.py passes -DMLT_SOURCE_SIG; the marker is also what build.py's
// stale() greps the file for): the host side refuses a library that does not match the csrc/ beside it.
#ifndef MLT_SOURCE_SIG
#define MLT_SOURCE_SIG "unsigned-build!!"
#endif
static const char g_source_sig[] = "MLTCNN_SOURCE_SIG=" MLT_SOURCE_SIG;
const char *mlt_build_signature(void) { return g_source_sig + sizeof("MLTCNN_SOURCE_SIG=") - 1; }

int mlt_num_logits(int size) { return size == 128 ? 9 : (size == 64 || size == 32 || size == 16) ? 15 : 0; }

const char *mlt_last_error(const mlt_ctx *ctx) { return ctx ? ctx->err.c_str() : g_init_error.c_str(); }

int mlt_load_weights(mlt_ctx *ctx, int size, const void *blob, size_t bytes) {
  if (!ctx || !blob) return MLT_ERR_ARG;
  const int si = size_index(size);
  if (si < 0) { ctx->err = "unsupported CU size"; return MLT_ERR_ARG; }
  // the library keeps the blob (5.6 - 6.3 MB): mlt_calibrate re-packs from it
  std::vector<char> keep((const char *)blob, (const char *)blob + bytes);
  const int rc = load_all(ctx, size, keep.data(), keep.size(), nullptr);
  if (rc == MLT_OK) ctx->sz[si].blob = std::move(keep);
  return rc;
}

int mlt_calibrate(mlt_ctx *ctx, int size, const int16_t *org, const int16_t *pred, const int32_t *poc, const int32_t *qp, int n, int mode) {
  if (!ctx) return MLT_ERR_ARG;
  if (!org || !pred || !poc || !qp || n <= 0 || n > kCalibCallerMax || (mode != MLT_CALIB_APPEND && mode != MLT_CALIB_REPLACE)) { ctx->err = "mlt_calibrate: bad argument"; return MLT_ERR_ARG; }
  SizeState *st;
  int rc = check_size(ctx, size, &st);
  if (rc) return rc;
  if (st->blob.empty()) { ctx->err = "mlt_calibrate: no weight blob kept for this size"; return MLT_ERR_WEIGHTS; }
  if (!st->calibrate && !st->small_mix) return MLT_OK;  // configured exact / calibration switched off: nothing to decide
  const CalibExtra ex{org, pred, poc, qp, n, mode == MLT_CALIB_REPLACE};
  std::vector<char> keep = std::move(st->blob);  // (load_all clears the kept blob on failure)
  rc = load_all(ctx, size, keep.data(), keep.size(), &ex);
  if (rc == MLT_OK) ctx->sz[size_index(size)].blob = std::move(keep);
  return rc;
}

// Host-only hook (not part of include/mltcnn.h; no HIP call, no device): the LAUNCH PLAN of one batch -- what run_network would enqueue for n CUs of `size`
// with hi+lo weights in the launch units of w2_units and the exact arithmetic in those of x_units (tier: 0 the fp16 tiers as given by the two masks, 1 exact
// everywhere, 5 exact-lite everywhere), planes 8-byte aligned or not.  The models are built from the blob on the host exactly as mlt_load_weights builds them;
// the dispatcher then runs in plan mode (mlt_ctx::plan).  One launch per line: "name [variant, layouts]".  Returns the number of launches, or -1.
int mlt_plan_describe(const void *blob, size_t bytes, int size, int n, int tier, unsigned w2_units, unsigned x_units, int aligned, char *out, size_t cap) {
  const int si = size_index(size);
  if (!blob || !out || cap == 0 || si < 0 || n <= 0) return -1;
  mlt_ctx ctx;
  std::vector<std::string> plan;
  ctx.plan = &plan;
  ctx.plan_detail = (aligned & 2) != 0;   // (bit 1 of `aligned`: every record also lists the launch's buffers -- the hand-offs between launches)
  ctx.lds_oob_zero = true;    // (what every gfx950 device reports: mlt_probe_lds_oob)
  SizeState &st = ctx.sz[si];
  st.size = size; st.enabled = st.loaded = true;
  st.head_index = size == 128 ? 2 : 0;
  std::string err;
  const bool whole_exact = tier == 1 || tier == 5;
  if (!mlt::build_model(blob, bytes, tier == 5 ? mlt::MLT_MODEL_XLITE : tier == 1 ? mlt::MLT_MODEL_EXACT : mlt::MLT_MODEL_FAST, size, st.model, err)) return -1;
  if (!whole_exact && w2_units && !mlt::build_model(blob, bytes, mlt::MLT_MODEL_W2, size, st.model_w2, err)) return -1;
  if (!whole_exact && x_units && !mlt::build_model(blob, bytes, mlt::MLT_MODEL_EXACT, size, st.model_exact, err)) return -1;
  st.exact = whole_exact;
  st.w2 = !whole_exact && w2_units != 0; st.w2_units = st.w2 ? w2_units : 0; st.x_units = whole_exact ? 0 : x_units;
  const long cs = (long)size * size;
  const int16_t *planes = (const int16_t *)(uintptr_t)((aligned & 1) ? 0x1000 : 0x1002);   // never dereferenced: only the alignment is looked at
  int32_t *const flat_fake = (int32_t *)(uintptr_t)0x2000;   // (never dereferenced in plan mode; a constant so that the detailed records are reproducible)
  ctx.ws = (char *)(uintptr_t)0x100000000ull;   // (never touched in plan mode: a base that tells a workspace offset from a NULL pointer in the detailed records)
  const int rc = run_main(&ctx, st, n, planes, size, cs, planes, size, cs, nullptr, nullptr, nullptr, nullptr, whole_exact ? nullptr : flat_fake);
  ctx.ws = nullptr;
  if (rc) return -1;
  size_t pos = 0;
  for (const std::string &l : plan) {
    if (pos + l.size() + 2 > cap) return -1;
    std::memcpy(out + pos, l.data(), l.size());
    pos += l.size();
    out[pos++] = '\n';
  }
  out[pos] = 0;
  return (int)plan.size();
}

// Host-only hook (not part of include/mltcnn.h; no HIP call): the synthetic calibration set of a CU size, so that the numerics tools
// (tools/attribute_error.py, scripts/emul_fast.py) and the CPU tests see exactly the CUs the load-time calibration prices.  Buffers: dense
// [560][size][size] int16 org / pred, int32 poc / qp / content class; any of them may be NULL.  Returns the number of CUs (560) or -1.
int mlt_calibration_set_copy(int size, int16_t *org, int16_t *pred, int32_t *poc, int32_t *qp, int32_t *cls) {
  if (size_index(size) < 0) return -1;
  const CalibInputs &ci = calibration_set(size);
  const size_t cs = (size_t)size * size * kCalibN;
  if (org) std::memcpy(org, ci.org.data(), cs * 2);
  if (pred) std::memcpy(pred, ci.pred.data(), cs * 2);
  if (poc) std::memcpy(poc, ci.poc.data(), (size_t)kCalibN * 4);
  if (qp) std::memcpy(qp, ci.qp.data(), (size_t)kCalibN * 4);
  if (cls) for (int i = 0; i < kCalibN; ++i) cls[i] = ci.cls[(size_t)i];
  return kCalibN;
}

// CPU test hook of the tier search (mlt_tier_search.h; not part of include/mltcnn.h): no HIP call on this path
int mlt_tier_search_run(int kind, int n, float tolerance, float max_frac, const int *force,
                        int (*price_cb)(void *user, unsigned w2_units, unsigned x_units, int rounding, float *out3), void *user, int *result, float *figures) {
  if (!price_cb || !result || !figures || n <= 0) return MLT_ERR_ARG;
  struct CbPricer : mlt::TierPricer {
    int (*cb)(void *, unsigned, unsigned, int, float *); void *user;
    int price(unsigned w2u, unsigned xu, int r, mlt::TierPrice &out) override {
      float o[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      const int rc = cb(user, w2u, xu, r, o);
      out.rms = o[0]; out.max = o[1]; out.tail = o[2];
      out.g_valid = o[3] != 0.f; out.g_rms = o[4]; out.g_max = o[5]; out.g_tail = o[6]; out.g_thr = o[7]; out.g_flag = o[8];
      return rc;
    }
    int price_lite(mlt::TierPrice &out) override { return price(~0u, ~0u, 0, out); }
  } pricer;
  pricer.cb = price_cb; pricer.user = user;
  mlt::TierRules rules;
  if (tolerance > 0.f) rules.tolerance = tolerance;
  if (max_frac > 0.f) rules.max_frac = max_frac;
  mlt::TierForce f;
  if (force) {
    f.rounding = force[0]; f.w2_mask = force[1]; f.x_mask = force[2]; f.w2_units = force[3]; f.small_prefix = force[4];
    f.no_roundings = force[5] != 0; f.no_w2 = force[6] != 0; f.no_xmix = force[7] != 0; f.no_w2_units = force[8] != 0; f.no_x_units = force[9] != 0; f.no_lite = force[10] != 0; f.no_mag_guard = force[11] != 0;
  }
  mlt::TierChoice ch;
  const int rc = kind == 0 ? mlt::search_tier_128(pricer, rules, f, n, ch) : mlt::search_tier_small(pricer, rules, f, n, ch);
  result[0] = ch.exact ? 1 : 0; result[1] = ch.w2 ? 1 : 0; result[2] = (int)ch.w2_units; result[3] = (int)ch.x_units; result[4] = ch.rounding; result[5] = ch.priced;
  result[6] = ch.lite ? 1 : 0; result[7] = ch.mag_thr > 0.f ? 1 : 0;
  figures[0] = ch.price.rms; figures[1] = ch.price.max; figures[2] = ch.price.tail; figures[3] = ch.mag_thr;
  return rc;
}

int mlt_arithmetic(mlt_ctx *ctx, int size, mlt_arith_info *out) {
  if (!ctx || !out) return MLT_ERR_ARG;
  // the caller says how large ITS struct is; only that much is written (a later, longer mlt_arith_info cannot overrun an older caller)
  if (out->struct_size < offsetof(mlt_arith_info, mag_guard_thr)) { ctx->err = "mlt_arith_info.struct_size does not cover the ABI-4 fields"; return MLT_ERR_ARG; }
  SizeState *st;
  int rc = check_size(ctx, size, &st);
  if (rc) return rc;
  out->exact = st->exact ? 1 : st->lite ? 5 : st->x_units ? 4 : st->w2 ? (st->w2_units != 0xFFu ? 3 : 2) : 0;
  out->w2_stages = st->w2 ? (int32_t)st->w2_mask : 0;
  out->x_stages = st->exact ? 0 : (int32_t)st->x_mask;
  out->w2_units = st->w2 ? (int32_t)st->w2_units : 0;
  out->x_units = st->exact ? 0 : (int32_t)st->x_units;
  out->rounding = st->model.rounding;
  out->guard_margin = (!st->exact && st->margin_guard) ? st->guard_margin : 0.f;
  out->calibrated = st->calibrated ? 1 : 0;
  out->calib_rms = st->calib_rms; out->calib_max = st->calib_max;
  out->flat_guard = (!st->exact && st->flat_guard) ? 1 : 0;
  out->decision_guard = (!st->exact && st->margin_guard) ? 1 : 0;
  out->guard_reruns = st->reruns;
  out->calib_cus = st->calib_cus; out->calib_caller_cus = st->calib_caller_cus;
  if (out->struct_size >= sizeof(mlt_arith_info)) {  // round 6 fields: written only into a struct that has them
    out->mag_guard_thr = st->exact ? 0.f : st->mag_thr;
    out->mag_guard_flagged = st->exact ? 0.f : st->mag_flag;
    out->mag_guard_kind = st->exact ? 0 : st->mag_kind;
  }
  return MLT_OK;
}

#pragma GCC visibility pop
}  // extern "C"
namespace {
// one context on one device (cfg->device is ignored: `device` decides)
int init_one(const mlt_config *cfg, int device, mlt_ctx **out) {
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
    g_init_error = "no usable HIP device (this library has no CPU fallback)";
    return MLT_ERR_NO_DEVICE;
  }
  if (hipSetDevice(device) != hipSuccess) { g_init_error = "hipSetDevice failed"; return MLT_ERR_NO_DEVICE; }
  mlt_ctx *ctx = new (std::nothrow) mlt_ctx();
  if (!ctx) return MLT_ERR_NOMEM;
  ctx->device = device;
  ctx->max_batch = cfg->max_batch > 0 ? cfg->max_batch : 4096;
  if (cfg->tolerance > 0.f) ctx->tolerance = cfg->tolerance;
  ctx->xlite = (cfg->flags & MLT_FLAG_EXACT_LITE) != 0;
  // decision guard: two logits that are each within `tolerance` of the reference change their difference by at most 2 x tolerance.  The
  // admission of the fast arithmetic is calibrated, not proven (the tail probes put single logits at up to ~1.1 x tolerance), so the
  // default threshold is 3 x tolerance: everything below it is re-evaluated exactly (the extra re-runs are a fraction of a per cent)
  ctx->guard_margin = cfg->guard_margin > 0.f ? cfg->guard_margin : 3.f * ctx->tolerance;
  ctx->guard_margin_configured = cfg->guard_margin > 0.f;
  if (const char *e = tuning_env("MLT_CHUNK")) { int v = std::atoi(e); if (v > 0) ctx->chunk = v; }
  ctx->guard_select_kernel = tuning_env("MLT_GUARD_SELECT_KERNEL") != nullptr;
  if (const char *e = tuning_env("MLT_STAGE_CHUNK")) { int v = std::atoi(e); if (v > 0) ctx->stage_chunk = v; }
  if (ctx->stage_chunk > ctx->chunk) ctx->stage_chunk = ctx->chunk;
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) { g_init_error = "hipStreamCreate failed"; delete ctx; return MLT_ERR_HIP; }
  if (hipMalloc((void **)&ctx->zero_page, 65536) != hipSuccess || hipMemset(ctx->zero_page, 0, 65536) != hipSuccess) {
    g_init_error = "zero page allocation failed";
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return MLT_ERR_NOMEM;
  }
  ctx->own_stream = true;
  {
    // the chain kernels take conv padding from DS reads beyond the LDS allocation (zeros on gfx950) instead of zero masks: probe, once
    // per context, that this device behaves so -- (ab)using the zero page as the 4-byte result slot, restored afterwards; a device that
    // does not (or MLT_NO_LDS_OOB=1) gets the masked form of the same kernels
    int ok = 0;
    const bool ran = tuning_env("MLT_NO_LDS_OOB") == nullptr && mlt_probe_lds_oob((int *)ctx->zero_page, ctx->stream) == hipSuccess &&
                     hipMemcpyAsync(&ok, ctx->zero_page, sizeof ok, hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                     hipStreamSynchronize(ctx->stream) == hipSuccess;
    if (hipMemsetAsync(ctx->zero_page, 0, sizeof ok, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) {
      g_init_error = "zero page reset failed";
      (void)hipFree(ctx->zero_page);
      (void)hipStreamDestroy(ctx->stream);
      delete ctx;
      return MLT_ERR_HIP;
    }
    ctx->lds_oob_zero = ran && ok == 1;
  }
  const uint32_t mask = cfg->size_mask ? cfg->size_mask : MLT_SIZE_128;  // reference gate: 128 only (EncCu.cpp:754)
  static const int sizes[4] = {128, 64, 32, 16};
  for (int i = 0; i < 4; ++i) {
    SizeState &st = ctx->sz[i];
    st.size = sizes[i];
    st.enabled = (mask >> i) & 1u;
    // precision: 128 -> fast (single fp16 pass) unless MLT_FLAG_EXACT_128; 64/32/16 -> exact (fp16 hi+lo pairs, 3 passes)
    // unless MLT_FLAG_FAST_SMALL.  See DESIGN.md "Numerics".
    st.want_exact = st.exact = sizes[i] == 128 ? (cfg->flags & MLT_FLAG_EXACT_128) != 0 : (cfg->flags & MLT_FLAG_FAST_SMALL) == 0;
    st.margin_guard = (cfg->flags & MLT_FLAG_NO_DECISION_GUARD) == 0;  // ABI 4: on by default (MLT_FLAG_DECISION_GUARD is accepted and has no effect)
    st.flat_guard = st.cfg_flat_guard = (cfg->flags & MLT_FLAG_NO_FLAT_GUARD) == 0;
    st.cfg_mag_guard = (cfg->flags & MLT_FLAG_NO_MAGNITUDE_GUARD) == 0;
    // the calibration decides "fast or exact" for the 128 model; MLT_FLAG_FAST_SMALL is an explicit request for fast
    st.calibrate = sizes[i] == 128 && (cfg->flags & MLT_FLAG_NO_CALIBRATION) == 0;
    // Round 6: only the 64 x 64 model.  A non-exact tier brings the guards with it, and a re-run of even ONE CU costs ~0.3 ms whatever the model (the exact chain's ~25
    // dependent launches, each streaming a layer's weight planes through a few workgroups): a tier must save more than that per batch to be worth having.
    //   16 x 16: layer0 works on 8 x 8 maps -- the calibrated prefix tier saved 9 % of the exact step and its one re-run per 4096-CU batch (the decision guard's usual
    //            catch) cost 0.34 ms of 0.43: 5.30 M CU/s against 8.66 M exact (profiles/r06m_small_exact_vs_prefix.txt);
    //   32 x 32: the exact-lite tier the search ended on is 0.3 - 2 % faster than exact on content that flags nothing and 29 % slower with 5 % flat CUs in the batch
    //            (2.51 M against 3.53 M, profiles/r06n_small_exact_vs_calibrated.txt);
    //   64 x 64: layer0.0 on the fused single-pass kernel saves 16 % (1.24 M against 1.05 M exact, also on natural scenes); break-even at ~4 % flagged CUs: it stays.
    // Exact also means 2e-5 instead of 1e-4 .. 4e-4 from the oracle and no data-dependent latency.
    st.small_mix = sizes[i] == 64 && st.want_exact && (cfg->flags & MLT_FLAG_NO_CALIBRATION) == 0;
    st.head_index = cfg->head_index[i] >= 0 ? cfg->head_index[i] : (sizes[i] == 128 ? 2 : 0);  // EncCu.cpp:913-919
    if (st.enabled && cfg->weights_dir) {
      char path[1024];
      std::snprintf(path, sizeof path, "%s/MLTORPQ_splitMode_%d.mltw", cfg->weights_dir, sizes[i]);  // cf. EncCu.cpp:899
      FILE *f = std::fopen(path, "rb");
      if (!f) { g_init_error = std::string("cannot open ") + path; mlt_shutdown(ctx); return MLT_ERR_WEIGHTS; }
      std::fseek(f, 0, SEEK_END);
      long len = std::ftell(f);
      std::fseek(f, 0, SEEK_SET);
      std::vector<char> buf(len > 0 ? len : 0);
      const size_t got = len > 0 ? std::fread(buf.data(), 1, len, f) : 0;
      std::fclose(f);
      int rc = got == (size_t)len ? mlt_load_weights(ctx, sizes[i], buf.data(), buf.size()) : MLT_ERR_WEIGHTS;
      if (rc) { g_init_error = ctx->err.empty() ? std::string("short read: ") + path : ctx->err; mlt_shutdown(ctx); return rc; }
    }
  }
  *out = ctx;
  return MLT_OK;
}

// contiguous shard of n items for device g of G (fastintercu-vvc_amd/shard.py: shard_bounds)
inline int shard_lo(int n, int g, int G) { return (int)(((long long)n * g) / G); }
inline mlt_ctx *device_of(mlt_ctx *ctx, int i) { return i == 0 ? ctx : ctx->peers[(size_t)i - 1]; }
}  // namespace

extern "C" {
#pragma GCC visibility push(default)

int mlt_init(const mlt_config *cfg, mlt_ctx **out) {
  std::lock_guard<std::mutex> lock(g_mutex);
  // ABI 4 is a hard break: older, shorter structs (the 56-byte ABI-2 mlt_config) are rejected -- such a binary would also pass a
  // 32-byte mlt_arith_info to mlt_arithmetic
  if (!cfg || !out || cfg->struct_size != sizeof(mlt_config)) { g_init_error = "bad mlt_config (struct_size does not match ABI 4)"; return MLT_ERR_ARG; }
  *out = nullptr;
  const int nd = cfg->n_devices;
  if (nd < 0 || nd > MLT_MAX_DEVICES) { g_init_error = "bad mlt_config.n_devices"; return MLT_ERR_ARG; }
  mlt_ctx *ctx = nullptr;
  int rc = init_one(cfg, nd > 0 ? cfg->devices[0] : cfg->device, &ctx);
  if (rc) return rc;
  for (int i = 1; i < nd; ++i) {  // one full context per further device; weights_dir is read (and calibrated) by each
    mlt_ctx *peer = nullptr;
    if ((rc = init_one(cfg, cfg->devices[i], &peer))) { mlt_shutdown(ctx); return rc; }
    ctx->peers.push_back(peer);
  }
  *out = ctx;
  return MLT_OK;
}

int mlt_num_devices(const mlt_ctx *ctx) { return ctx ? 1 + (int)ctx->peers.size() : 0; }

mlt_ctx *mlt_device_ctx(mlt_ctx *ctx, int index) {
  if (!ctx || index < 0 || index > (int)ctx->peers.size()) return nullptr;
  return device_of(ctx, index);
}

void mlt_shutdown(mlt_ctx *ctx) {
  if (!ctx) return;
  for (mlt_ctx *p : ctx->peers) mlt_shutdown(p);
  ctx->peers.clear();
  (void)hipSetDevice(ctx->device);
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  for (auto &kv : ctx->prof)
    for (auto &ev : kv.second.ev) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  for (int i = 0; i < 4; ++i) { free_model(ctx->sz[i].model); free_model(ctx->sz[i].model_exact); free_model(ctx->sz[i].model_w2); free_model(ctx->sz[i].model_xl); }
  for (SingleCu &sg : ctx->single) {
    if (sg.exec) (void)hipGraphExecDestroy(sg.exec);
    if (sg.graph) (void)hipGraphDestroy(sg.graph);
    if (sg.h_stage) (void)hipHostFree(sg.h_stage);
    if (sg.d_stage) (void)hipFree(sg.d_stage);
  }
  for (Deferred &df : ctx->deferred) {
    if (df.h_in) (void)hipHostFree(df.h_in);
    if (df.h_out) (void)hipHostFree(df.h_out);
    if (df.d_in) (void)hipFree(df.d_in);
    if (df.d_out) (void)hipFree(df.d_out);
    for (int b = 0; b < 2; ++b) if (df.done[b]) (void)hipEventDestroy(df.done[b]);
  }
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->zero_page) (void)hipFree(ctx->zero_page);
  if (ctx->h_res) (void)hipHostFree(ctx->h_res);
  if (ctx->copy_stream) (void)hipStreamDestroy(ctx->copy_stream);
  for (int b = 0; b < 2; ++b) { if (ctx->ev_h2d[b]) (void)hipEventDestroy(ctx->ev_h2d[b]); if (ctx->ev_done[b]) (void)hipEventDestroy(ctx->ev_done[b]); }
  if (ctx->stage) (void)hipFree(ctx->stage);
  if (ctx->guard_dev) (void)hipFree(ctx->guard_dev);
  if (ctx->gstage) (void)hipFree(ctx->gstage);
  if (ctx->guard_host) (void)hipHostFree(ctx->guard_host);
  if (ctx->ev_guard) (void)hipEventDestroy(ctx->ev_guard);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  delete ctx;
}

int mlt_set_stream(mlt_ctx *ctx, void *hip_stream) {
  if (!ctx) return MLT_ERR_ARG;
  // (a peer context handed out by mlt_device_ctx lives on another GPU than the current one: the replacement stream must be created there)
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return MLT_ERR_NO_DEVICE; }
  if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
  if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
  ctx->stream = nullptr;
  ctx->own_stream = false;
  if (hip_stream) {
    ctx->stream = (hipStream_t)hip_stream;
  } else {  // NULL: back to a stream owned by the context
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;
  }
  return MLT_OK;
}

int mlt_synchronize(mlt_ctx *ctx) {
  if (!ctx) return MLT_ERR_ARG;
  for (mlt_ctx *p : ctx->peers) {
    const int rc = mlt_synchronize(p);
    if (rc) { ctx->err = p->err; return rc; }
  }
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return MLT_ERR_NO_DEVICE; }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MLT_OK;
}

int mlt_predict_batch_device(mlt_ctx *ctx, int n, int size, const void *d_org, const void *d_pred, const void *d_poc, const void *d_qp,
                             void *d_split_mode, void *d_logits) {
  if (!ctx) return MLT_ERR_ARG;
  if (n < 0 || !d_split_mode || (n > 0 && (!d_org || !d_pred || !d_poc || !d_qp))) { ctx->err = "bad argument"; return MLT_ERR_ARG; }
  SizeState *st;
  int rc = check_size(ctx, size, &st);
  if (rc) return rc;
  if (n == 0) return MLT_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return MLT_ERR_NO_DEVICE; }
  const int nl = st->model.n_logits;
  const long cs = (long)size * size;
  for (int i0 = 0; i0 < n; i0 += ctx->chunk) {
    const int c = n - i0 < ctx->chunk ? n - i0 : ctx->chunk;
    const Planes pl{(const int16_t *)d_org + (size_t)i0 * cs, (const int16_t *)d_pred + (size_t)i0 * cs, size, cs, size, cs};
    rc = run_checked(ctx, *st, c, pl, (const int32_t *)d_poc + i0, (const int32_t *)d_qp + i0, (int32_t *)d_split_mode + i0,
                     d_logits ? (float *)d_logits + (size_t)i0 * nl : nullptr);
    if (rc) return rc;
  }
  return MLT_OK;
}

static int predict_batch_single(mlt_ctx *ctx, int n, int size, const int16_t *org, const int16_t *pred, const int32_t *poc, const int32_t *qp,
                      int32_t *split_mode, float *logits) {
  if (!ctx) return MLT_ERR_ARG;
  if (n < 0 || !split_mode || (n > 0 && (!org || !pred || !poc || !qp))) { ctx->err = "bad argument"; return MLT_ERR_ARG; }
  SizeState *st;
  int rc = check_size(ctx, size, &st);
  if (rc) return rc;
  if (n == 0) return MLT_OK;
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return MLT_ERR_NO_DEVICE; }
  const int nl = st->model.n_logits;
  const size_t cs = (size_t)size * size;
  // Host batches are pipelined in sub-chunks through TWO staging sets: the H2D copy of sub-chunk k+1 (copy stream) runs
  // under the kernels of sub-chunk k (compute stream).  256 MiB of planes per 4096 CUs take about as long over PCIe as the
  // network does, so the overlap is worth ~1.5x end to end when the caller's buffers are pinned (mlt_alloc_pinned).
  const int cap = n < ctx->stage_chunk ? n : ctx->stage_chunk;
  const size_t plane = (cs * 2 * cap + 255) / 256 * 256;
  const size_t small = ((size_t)cap * 4 + 255) / 256 * 256;
  const size_t lgb = ((size_t)cap * nl * 4 + 255) / 256 * 256;
  const size_t setbytes = 2 * plane + 3 * small + lgb;
  const int nset = n > cap ? 2 : 1;
  if ((rc = ensure_stage(ctx, nset * setbytes))) return rc;
  if (nset == 2 && !ctx->copy_stream) {
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
      HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_h2d[b], hipEventDisableTiming));
      if (!ctx->ev_done[b]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_done[b], hipEventDisableTiming));
    }
  }
  struct Set { int16_t *d_org, *d_pred; int32_t *d_poc, *d_qp, *d_split; float *d_lg; };
  auto set_of = [&](int b) {
    char *base = ctx->stage + (size_t)b * setbytes;
    return Set{(int16_t *)base, (int16_t *)(base + plane), (int32_t *)(base + 2 * plane), (int32_t *)(base + 2 * plane + small),
               (int32_t *)(base + 2 * plane + 2 * small), (float *)(base + 2 * plane + 3 * small)};
  };
  const bool guards = st->guards();
  GuardSlot gs[2];
  if (guards)
    for (int b = 0; b < nset; ++b)
      if ((rc = guard_slot(ctx, b, cap, nl, &gs[b]))) return rc;
  // Results come back through pinned buffers owned by the context: a D2H into the caller's (usually pageable) arrays
  // would block the host until the kernels are done and serialise the next sub-chunk's H2D behind them.
  const size_t hres_set = (size_t)cap * 4 + (size_t)cap * nl * 4;
  if (ctx->h_res_bytes < 2 * hres_set) {
    if (ctx->h_res) (void)hipHostFree(ctx->h_res);
    ctx->h_res = nullptr; ctx->h_res_bytes = 0;
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_res, 2 * hres_set, hipHostMallocDefault));
    ctx->h_res_bytes = 2 * hres_set;
  }
  int pend_i0[2] = {-1, -1}, pend_c[2] = {0, 0};
  auto fetch = [&](int b, int c) -> int {  // results of set b -> pinned (asynchronous)
    const Set S = set_of(b);
    char *hb = ctx->h_res + (size_t)b * hres_set;
    HIP_TRY(ctx, hipMemcpyAsync(hb, S.d_split, (size_t)c * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (logits) HIP_TRY(ctx, hipMemcpyAsync(hb + (size_t)cap * 4, S.d_lg, (size_t)c * nl * 4, hipMemcpyDeviceToHost, ctx->stream));
    return MLT_OK;
  };
  auto flush = [&](int b) -> int {  // sub-chunk in set b has completed: guard fix-up if needed, then hand its results to the caller
    if (pend_i0[b] < 0) return MLT_OK;
    HIP_TRY(ctx, hipEventSynchronize(ctx->ev_done[b]));
    if (guards) {
      const int k = *gs[b].h_count;
      if (k < 0 || k > pend_c[b]) { ctx->err = "guard: bad flagged-CU count"; return MLT_ERR_HIP; }
      if (k > 0) {  // the set's inputs are still in place (the next H2D into it is issued after this flush)
        const Set S = set_of(b);
        const Planes pl{S.d_org, S.d_pred, size, (long)cs, size, (long)cs};
        int r = guard_fixup_async(ctx, *st, k, pl, S.d_poc, S.d_qp, S.d_split, logits ? S.d_lg : nullptr, gs[b]);
        if (r) return r;
        if ((r = fetch(b, pend_c[b]))) return r;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      }
    }
    const char *hb = ctx->h_res + (size_t)b * hres_set;
    std::memcpy(split_mode + pend_i0[b], hb, (size_t)pend_c[b] * 4);
    if (logits) std::memcpy(logits + (size_t)pend_i0[b] * nl, hb + (size_t)cap * 4, (size_t)pend_c[b] * nl * 4);
    pend_i0[b] = -1;
    return MLT_OK;
  };
  if (!ctx->ev_done[0])
    for (int b = 0; b < 2; ++b) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_done[b], hipEventDisableTiming));
  int k = 0;
  for (int i0 = 0; i0 < n; i0 += cap, ++k) {
    const int c = n - i0 < cap ? n - i0 : cap;
    const int b = nset == 2 ? (k & 1) : 0;
    const Set S = set_of(b);
    hipStream_t cps = nset == 2 ? ctx->copy_stream : ctx->stream;
    if ((rc = flush(b))) return rc;  // set b: sub-chunk k-2 fully drained (host-side wait)
    HIP_TRY(ctx, hipMemcpyAsync(S.d_org, org + (size_t)i0 * cs, cs * 2 * c, hipMemcpyHostToDevice, cps));
    HIP_TRY(ctx, hipMemcpyAsync(S.d_pred, pred + (size_t)i0 * cs, cs * 2 * c, hipMemcpyHostToDevice, cps));
    HIP_TRY(ctx, hipMemcpyAsync(S.d_poc, poc + i0, (size_t)c * 4, hipMemcpyHostToDevice, cps));
    HIP_TRY(ctx, hipMemcpyAsync(S.d_qp, qp + i0, (size_t)c * 4, hipMemcpyHostToDevice, cps));
    if (nset == 2) {
      HIP_TRY(ctx, hipEventRecord(ctx->ev_h2d[b], cps));
      HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_h2d[b], 0));
    }
    const Planes pl{S.d_org, S.d_pred, size, (long)cs, size, (long)cs};
    if (guards) rc = run_guarded_async(ctx, *st, c, pl, S.d_poc, S.d_qp, S.d_split, logits ? S.d_lg : nullptr, gs[b]);
    else rc = run_main(ctx, *st, c, S.d_org, size, (long)cs, S.d_pred, size, (long)cs, S.d_poc, S.d_qp, S.d_split, logits ? S.d_lg : nullptr);
    if (rc) return rc;
    if ((rc = fetch(b, c))) return rc;
    HIP_TRY(ctx, hipEventRecord(ctx->ev_done[b], ctx->stream));
    pend_i0[b] = i0; pend_c[b] = c;
  }
  if ((rc = flush(k & 1))) return rc;        // older of the two pending sub-chunks first
  if ((rc = flush((k & 1) ^ 1))) return rc;
  if (nset == 2) HIP_TRY(ctx, hipStreamSynchronize(ctx->copy_stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MLT_OK;
}

int mlt_predict_batch(mlt_ctx *ctx, int n, int size, const int16_t *org, const int16_t *pred, const int32_t *poc, const int32_t *qp,
                      int32_t *split_mode, float *logits) {
  if (!ctx) return MLT_ERR_ARG;
  if (n < 0 || !split_mode || (n > 0 && (!org || !pred || !poc || !qp))) { ctx->err = "bad argument"; return MLT_ERR_ARG; }
  SizeState *st;
  int rc = check_size(ctx, size, &st);
  if (rc) return rc;
  if (n == 0) return MLT_OK;
  if (!ctx->peers.empty() && n > 1) {
    // multi-device context: contiguous shards (device g gets CUs [n g / G, n (g + 1) / G)), one host thread per further device, no
    // exchange between them; every device runs the single-device path on its shard, so the results are those of one device
    const int G = 1 + (int)ctx->peers.size(), nlg = st->model.n_logits;
    const size_t csz = (size_t)size * size;
    std::vector<mlt_ctx *> devs;
    for (int g = 0; g < G; ++g) devs.push_back(device_of(ctx, g));
    std::vector<int> rcs((size_t)G, MLT_OK);
    auto run = [&](int g) {
      const int lo = shard_lo(n, g, G), hi = shard_lo(n, g + 1, G);
      if (hi > lo) rcs[(size_t)g] = predict_batch_single(devs[(size_t)g], hi - lo, size, org + (size_t)lo * csz, pred + (size_t)lo * csz, poc + lo, qp + lo,
                                                         split_mode + lo, logits ? logits + (size_t)lo * nlg : nullptr);
    };
    std::vector<std::thread> th;
    for (int g = 1; g < G; ++g) th.emplace_back(run, g);
    run(0);
    for (std::thread &t : th) t.join();
    for (int g = 0; g < G; ++g)
      if (rcs[(size_t)g]) { if (g) ctx->err = "device " + std::to_string(devs[(size_t)g]->device) + ": " + devs[(size_t)g]->err; return rcs[(size_t)g]; }
    return MLT_OK;
  }
  return predict_batch_single(ctx, n, size, org, pred, poc, qp, split_mode, logits);
}

int mlt_predict(mlt_ctx *ctx, const int16_t *org, int org_stride, const int16_t *pred, int pred_stride, int size, int32_t poc, int32_t qp,
                int32_t *split_mode, float *logits_opt) {
  if (!ctx) return MLT_ERR_ARG;
  if (!org || !pred || !split_mode || org_stride < size || pred_stride < size) { ctx->err = "bad argument"; return MLT_ERR_ARG; }
  SizeState *st;
  int rc = check_size(ctx, size, &st);
  if (rc) return rc;
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return MLT_ERR_NO_DEVICE; }
  const int nl = st->model.n_logits;
  const size_t cs = (size_t)size * size;
  SingleCu &sg = ctx->single[size_index(size)];
  if (!sg.h_stage) {
    sg.plane = (cs * 2 + 255) / 256 * 256;
    HIP_TRY(ctx, hipHostMalloc((void **)&sg.h_stage, 2 * sg.plane + 256, hipHostMallocDefault));
    HIP_TRY(ctx, hipMalloc((void **)&sg.d_stage, 2 * sg.plane + 256));
    HIP_TRY(ctx, hipMemset(sg.d_stage + 2 * sg.plane, 0, 256));  // the flat-guard statistic of the single-CU path starts at zero (consume-and-clear)
  }
  // the gather of EncCu.cpp:810-830: rows of `size` Pels out of a `stride`-Pel pitch -> dense planes (pinned)
  for (int y = 0; y < size; ++y) {
    std::memcpy(sg.h_stage + (size_t)y * size * 2, org + (size_t)y * org_stride, (size_t)size * 2);
    std::memcpy(sg.h_stage + sg.plane + (size_t)y * size * 2, pred + (size_t)y * pred_stride, (size_t)size * 2);
  }
  int32_t *h_sc = (int32_t *)(sg.h_stage + 2 * sg.plane);  // [poc, qp, split, pad, logits...]
  h_sc[0] = poc; h_sc[1] = qp;
  HIP_TRY(ctx, hipMemcpyAsync(sg.d_stage, sg.h_stage, 2 * sg.plane + 8, hipMemcpyHostToDevice, ctx->stream));
  int16_t *d_org = (int16_t *)sg.d_stage, *d_pred = (int16_t *)(sg.d_stage + sg.plane);
  int32_t *d_sc = (int32_t *)(sg.d_stage + 2 * sg.plane);  // [poc, qp, split, flagged count, logits (<= 16) ..., flat, idx]
  const Planes pl{d_org, d_pred, size, (long)cs, size, (long)cs};
  const bool guards = st->guards();
  GuardSlot g;
  g.d_count = d_sc + 3; g.d_flat = d_sc + 20; g.d_idx = d_sc + 21; g.d_lg = (float *)(d_sc + 4); g.d_mag = (float *)(d_sc + 22); g.h_count = h_sc + 3;
  g.single = true;  // (d_flat was zeroed with the staging buffer and is cleared by every call's heads kernel)
  auto chain = [&]() -> int {  // the kernel chain of one CU (captured into a hipGraph below)
    if (!guards) return run_main(ctx, *st, 1, d_org, size, (long)cs, d_pred, size, (long)cs, d_sc, d_sc + 1, d_sc + 2, (float *)(d_sc + 4));
    int r = run_guarded_async(ctx, *st, 1, pl, d_sc, d_sc + 1, d_sc + 2, (float *)(d_sc + 4), g);  // its 4-byte count D2H lands in h_sc[3]
    return r;
  };
  const bool no_graph = tuning().no_graph;
  bool replayed = false;
  if (!no_graph && !ctx->profile && ctx->own_stream) {
    if (sg.exec && (sg.ws_gen_at_capture != ctx->ws_gen || sg.stream_at_capture != ctx->stream)) {  // workspace re-allocated since: re-capture
      (void)hipGraphExecDestroy(sg.exec); (void)hipGraphDestroy(sg.graph);
      sg.exec = nullptr; sg.graph = nullptr;
    }
    if (!sg.exec) {
      // first call: run eagerly once (allocates the workspace, configures every kernel), then capture the same chain
      if ((rc = chain())) return rc;
      HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
      if (hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        rc = chain();
        hipGraph_t gr = nullptr;
        const hipError_t ce = hipStreamEndCapture(ctx->stream, &gr);
        if (rc == MLT_OK && ce == hipSuccess && gr && hipGraphInstantiate(&sg.exec, gr, nullptr, nullptr, 0) == hipSuccess) {
          sg.graph = gr; sg.ws_gen_at_capture = ctx->ws_gen; sg.stream_at_capture = ctx->stream;
        } else {
          if (gr) (void)hipGraphDestroy(gr);
          sg.exec = nullptr;
          (void)hipGetLastError();
        }
      }
      replayed = true;  // the eager run above already produced this call's result
    } else {
      HIP_TRY(ctx, hipGraphLaunch(sg.exec, ctx->stream));
      replayed = true;
    }
  }
  if (!replayed && (rc = chain())) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(h_sc + 2, d_sc + 2, (size_t)(2 + nl) * 4, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  if (guards && h_sc[3] != 0) {  // flagged (flat content / near-tie on the decision head): re-evaluate with the exact arithmetic
    if ((rc = guard_fixup_async(ctx, *st, 1, pl, d_sc, d_sc + 1, d_sc + 2, (float *)(d_sc + 4), g))) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(h_sc + 2, d_sc + 2, (size_t)(2 + nl) * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  }
  *split_mode = h_sc[2];
  if (logits_opt) std::memcpy(logits_opt, h_sc + 4, (size_t)nl * 4);
  return MLT_OK;
}

// ---- deferred single-CU prediction ----
namespace {
// device / pinned layout of one output set: split[CAP] | logits[CAP * nl] | flagged count (16 ints) | flat[CAP] | idx[CAP]
struct DeferredOut { int32_t *split; float *lg; int32_t *count, *flat, *idx; float *mag; };
DeferredOut deferred_out(char *base, int nl) {
  DeferredOut o;
  o.split = (int32_t *)base; o.lg = (float *)(base + (size_t)MLT_DEFER_CAP * 4);
  o.count = (int32_t *)(base + (size_t)MLT_DEFER_CAP * 4 * (1 + nl));
  o.flat = o.count + 16; o.idx = o.flat + MLT_DEFER_CAP; o.mag = (float *)(o.idx + MLT_DEFER_CAP);
  return o;
}
size_t deferred_fetch_bytes(int nl) { return (size_t)MLT_DEFER_CAP * 4 * (1 + nl) + 64; }

int deferred_launch(mlt_ctx *ctx, SizeState *st, Deferred &df) {  // launch the accumulating generation as one batch
  if (df.n == 0) return MLT_OK;
  const int size = st->size, nl = st->model.n_logits, b = (int)(df.gen & 1), n = df.n;
  char *hi = df.h_in + (size_t)b * df.in_set, *di = df.d_in + (size_t)b * df.in_set;
  char *ho = df.h_out + (size_t)b * df.out_set, *dout = df.d_out + (size_t)b * df.out_set;
  const size_t planes = (size_t)MLT_DEFER_CAP * df.plane;
  // org planes, pred planes, poc, qp: four regions of the set, only the first n entries of each are live
  HIP_TRY(ctx, hipMemcpyAsync(di, hi, (size_t)n * df.plane, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(di + planes, hi + planes, (size_t)n * df.plane, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(ctx, hipMemcpyAsync(di + 2 * planes, hi + 2 * planes, (size_t)MLT_DEFER_CAP * 8, hipMemcpyHostToDevice, ctx->stream));
  int32_t *d_poc = (int32_t *)(di + 2 * planes), *d_qp = d_poc + MLT_DEFER_CAP;
  const DeferredOut od = deferred_out(dout, nl), oh = deferred_out(ho, nl);
  const Planes pl{(const int16_t *)di, (const int16_t *)(di + planes), size, (long)(df.plane / 2), size, (long)(df.plane / 2)};
  int rc;
  if (st->guards()) {
    GuardSlot g;
    g.d_flat = od.flat; g.d_idx = od.idx; g.d_count = od.count; g.d_lg = od.lg; g.d_mag = od.mag; g.phase = &df.phase[b]; g.h_count = oh.count;
    rc = run_guarded_async(ctx, *st, n, pl, d_poc, d_qp, od.split, od.lg, g);
    df.guard_pending[b] = true;
  } else {
    rc = run_main(ctx, *st, n, pl.org, pl.org_rs, pl.org_cs, pl.pred, pl.pred_rs, pl.pred_cs, d_poc, d_qp, od.split, od.lg);
    df.guard_pending[b] = false;
  }
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ho, dout, deferred_fetch_bytes(nl), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipEventRecord(df.done[b], ctx->stream));
  df.n_launched[b] = n;
  df.gen_of_set[b] = df.gen;
  ++df.gen;
  df.n = 0;
  return MLT_OK;
}

// first mlt_wait on a finished batch: re-evaluate its flagged CUs with the exact arithmetic (the set's inputs stay in place
// until the set is reused two batches later)
int deferred_guard_fixup(mlt_ctx *ctx, SizeState *st, Deferred &df, int b) {
  if (!df.guard_pending[b]) return MLT_OK;
  df.guard_pending[b] = false;
  const int size = st->size, nl = st->model.n_logits;
  char *di = df.d_in + (size_t)b * df.in_set, *ho = df.h_out + (size_t)b * df.out_set, *dout = df.d_out + (size_t)b * df.out_set;
  const size_t planes = (size_t)MLT_DEFER_CAP * df.plane;
  const DeferredOut od = deferred_out(dout, nl), oh = deferred_out(ho, nl);
  // (the set's two selection counters came back with the results: the launch counted on one and zeroed the other -- GuardSlot.phase -- so their sum is the count)
  const int k = oh.count[0] + oh.count[1];
  if (k == 0) return MLT_OK;
  if (k < 0 || k > df.n_launched[b] || !st->guards()) { ctx->err = "guard: bad flagged-CU count"; return MLT_ERR_HIP; }
  int32_t *d_poc = (int32_t *)(di + 2 * planes), *d_qp = d_poc + MLT_DEFER_CAP;
  const Planes pl{(const int16_t *)di, (const int16_t *)(di + planes), size, (long)(df.plane / 2), size, (long)(df.plane / 2)};
  GuardSlot g;
  g.d_flat = od.flat; g.d_idx = od.idx; g.d_count = od.count; g.d_lg = od.lg; g.d_mag = od.mag; g.h_count = oh.count;
  int rc = guard_fixup_async(ctx, *st, k, pl, d_poc, d_qp, od.split, od.lg, g);
  if (rc) return rc;
  HIP_TRY(ctx, hipMemcpyAsync(ho, dout, deferred_fetch_bytes(nl), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  return MLT_OK;
}
}  // namespace

int mlt_submit(mlt_ctx *ctx, const int16_t *org, int org_stride, const int16_t *pred, int pred_stride, int size, int32_t poc, int32_t qp,
               mlt_ticket *ticket) {
  if (!ctx) return MLT_ERR_ARG;
  if (!ctx->peers.empty() && ticket) {  // multi-device: CUs are dealt round-robin; the ticket's top byte names the device
    const int G = 1 + (int)ctx->peers.size(), g = (int)(ctx->rr++ % (unsigned)G);
    if (g > 0) {
      mlt_ctx *p = device_of(ctx, g);
      const int rc = mlt_submit(p, org, org_stride, pred, pred_stride, size, poc, qp, ticket);
      if (rc) { ctx->err = p->err; return rc; }
      *ticket |= (mlt_ticket)g << 56;
      return MLT_OK;
    }
  }
  if (!org || !pred || !ticket || org_stride < size || pred_stride < size) { ctx->err = "bad argument"; return MLT_ERR_ARG; }
  SizeState *st;
  int rc = check_size(ctx, size, &st);
  if (rc) return rc;
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return MLT_ERR_NO_DEVICE; }
  Deferred &df = ctx->deferred[size_index(size)];
  const int nl = st->model.n_logits;
  if (!df.h_in) {
    df.plane = ((size_t)size * size * 2 + 255) / 256 * 256;
    df.in_set = 2 * (size_t)MLT_DEFER_CAP * df.plane + (size_t)MLT_DEFER_CAP * 8;
    df.out_set = (deferred_fetch_bytes(nl) + (size_t)MLT_DEFER_CAP * 12 + 255) / 256 * 256;
    HIP_TRY(ctx, hipHostMalloc((void **)&df.h_in, 2 * df.in_set, hipHostMallocDefault));
    HIP_TRY(ctx, hipHostMalloc((void **)&df.h_out, 2 * df.out_set, hipHostMallocDefault));
    HIP_TRY(ctx, hipMalloc((void **)&df.d_in, 2 * df.in_set));
    HIP_TRY(ctx, hipMalloc((void **)&df.d_out, 2 * df.out_set));
    HIP_TRY(ctx, hipMemset(df.d_out, 0, 2 * df.out_set));   // (the selection's ticket words start at zero)
    for (int b = 0; b < 2; ++b) HIP_TRY(ctx, hipEventCreateWithFlags(&df.done[b], hipEventDisableTiming));
  }
  if (df.n == MLT_DEFER_CAP && (rc = deferred_launch(ctx, st, df))) return rc;  // full: flush, start the next generation
  const int b = (int)(df.gen & 1);
  if (df.n == 0 && df.gen_of_set[b] != ~0ull) HIP_TRY(ctx, hipEventSynchronize(df.done[b]));  // set b's previous batch fully drained
  char *hi = df.h_in + (size_t)b * df.in_set;
  const size_t planes = (size_t)MLT_DEFER_CAP * df.plane;
  char *ho = hi + (size_t)df.n * df.plane, *hp = hi + planes + (size_t)df.n * df.plane;
  for (int y = 0; y < size; ++y) {  // the gather of EncCu.cpp:810-830
    std::memcpy(ho + (size_t)y * size * 2, org + (size_t)y * org_stride, (size_t)size * 2);
    std::memcpy(hp + (size_t)y * size * 2, pred + (size_t)y * pred_stride, (size_t)size * 2);
  }
  int32_t *h_poc = (int32_t *)(hi + 2 * planes);
  h_poc[df.n] = poc;
  h_poc[MLT_DEFER_CAP + df.n] = qp;
  *ticket = df.gen * MLT_DEFER_CAP + (uint64_t)df.n;
  ++df.n;
  return MLT_OK;
}

int mlt_flush(mlt_ctx *ctx, int size) {
  if (!ctx) return MLT_ERR_ARG;
  for (mlt_ctx *p : ctx->peers) {
    const int rc = mlt_flush(p, size);
    if (rc) { ctx->err = p->err; return rc; }
  }
  SizeState *st;
  int rc = check_size(ctx, size, &st);
  if (rc) return rc;
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return MLT_ERR_NO_DEVICE; }
  Deferred &df = ctx->deferred[size_index(size)];
  return df.h_in ? deferred_launch(ctx, st, df) : MLT_OK;
}

int mlt_wait(mlt_ctx *ctx, int size, mlt_ticket ticket, int32_t *split_mode, float *logits_opt) {
  if (!ctx) return MLT_ERR_ARG;
  if (!ctx->peers.empty()) {
    const int g = (int)(ticket >> 56);
    if (g > (int)ctx->peers.size()) { ctx->err = "unknown ticket"; return MLT_ERR_ARG; }
    if (g > 0) {
      mlt_ctx *p = device_of(ctx, g);
      const int rc = mlt_wait(p, size, ticket & (((mlt_ticket)1 << 56) - 1), split_mode, logits_opt);
      if (rc) ctx->err = p->err;
      return rc;
    }
  }
  if (!split_mode) { ctx->err = "bad argument"; return MLT_ERR_ARG; }
  SizeState *st;
  int rc = check_size(ctx, size, &st);
  if (rc) return rc;
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return MLT_ERR_NO_DEVICE; }
  Deferred &df = ctx->deferred[size_index(size)];
  const uint64_t gen = ticket / MLT_DEFER_CAP;
  const int slot = (int)(ticket % MLT_DEFER_CAP), b = (int)(gen & 1), nl = st->model.n_logits;
  if (!df.h_in || gen > df.gen || (gen == df.gen && slot >= df.n)) { ctx->err = "unknown ticket"; return MLT_ERR_ARG; }
  if (gen == df.gen && (rc = deferred_launch(ctx, st, df))) return rc;  // still accumulating: launch it now
  if (df.gen_of_set[b] != gen || slot >= df.n_launched[b]) { ctx->err = "ticket expired (two newer batches were started)"; return MLT_ERR_ARG; }
  HIP_TRY(ctx, hipEventSynchronize(df.done[b]));
  if ((rc = deferred_guard_fixup(ctx, st, df, b))) return rc;
  const char *ho = df.h_out + (size_t)b * df.out_set;
  *split_mode = ((const int32_t *)ho)[slot];
  if (logits_opt) std::memcpy(logits_opt, ho + (size_t)MLT_DEFER_CAP * 4 + (size_t)slot * nl * 4, (size_t)nl * 4);
  return MLT_OK;
}

void *mlt_alloc_pinned(size_t bytes) {
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
  return p;
}

void mlt_free_pinned(void *p) { if (p) (void)hipHostFree(p); }

int mlt_profile_enable(mlt_ctx *ctx, int on) {
  if (!ctx) return MLT_ERR_ARG;
  if (hipSetDevice(ctx->device) != hipSuccess) { ctx->err = "hipSetDevice failed"; return MLT_ERR_NO_DEVICE; }
  HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
  for (auto &kv : ctx->prof)
    for (auto &ev : kv.second.ev) { (void)hipEventDestroy(ev.first); (void)hipEventDestroy(ev.second); }
  ctx->prof.clear();
  ctx->prof_order.clear();
  ctx->profile = on != 0;
  return MLT_OK;
}

int mlt_profile_read(mlt_ctx *ctx, mlt_kernel_time *out, int cap) {
  if (!ctx) return -1;
  if (hipSetDevice(ctx->device) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess) return -1;
  int i = 0;
  for (const std::string &name : ctx->prof_order) {
    ProfAcc &acc = ctx->prof[name];
    if (out && i < cap) {
      mlt_kernel_time &t = out[i];
      std::memset(&t, 0, sizeof t);
      std::snprintf(t.name, sizeof t.name, "%s", name.c_str());
      t.launches = acc.launches; t.flops = acc.flops; t.bytes = acc.bytes;
      float total = 0.f;
      for (auto &ev : acc.ev) { float ms = 0.f; if (hipEventElapsedTime(&ms, ev.first, ev.second) == hipSuccess) total += ms; }
      t.total_ms = total;
    }
    ++i;
  }
  return i;
}

#pragma GCC visibility pop
}  // extern "C"
