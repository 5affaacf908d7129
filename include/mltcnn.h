/*
 * mltcnn.h -- C ABI of the MI355X-native MLT-CNN inter-CU split predictor.
 *
 * Drop-in boundary for the inline CNN block of the reference encoder
 *   /root/reference/vtm-mlt-cpp/source/Lib/EncoderLib/EncCu.cpp:799-930
 * (the reference has no plugin API; this header IS the interface a maintainer binds, see
 * INTEGRATION.md).  Plain C types only, no C++ exceptions cross it, no torch / OpenCV.
 *
 * Error contract (mirrors the reference's swallow-and-continue, EncCu.cpp:902-905,923-926):
 * every call returns MLT_OK (0) or a non-zero code and NEVER aborts; on failure the caller
 * leaves predictedSplitMode = -1, for which EncModeCtrl::setNewModeList is a no-op
 * (EncModeCtrl.cpp:147-148) and the encoder falls back to exhaustive RDO.
 */
#ifndef MLTCNN_H
#define MLTCNN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLT_ABI_VERSION 4

enum {
  MLT_OK = 0,
  MLT_ERR_ARG = 1,        /* bad pointer / size / batch */
  MLT_ERR_NO_DEVICE = 2,  /* HIP device missing or unusable */
  MLT_ERR_WEIGHTS = 3,    /* weight file / blob missing or malformed */
  MLT_ERR_SIZE_DISABLED = 4, /* CU size not enabled in size_mask or no weights loaded for it */
  MLT_ERR_HIP = 5,        /* a HIP runtime call failed (see mlt_last_error) */
  MLT_ERR_NOMEM = 6
};

/* CU sizes the two reference models cover (EncCu.cpp:754 gate; only 128 is active upstream). */
#define MLT_SIZE_128 0x1u
#define MLT_SIZE_64 0x2u
#define MLT_SIZE_32 0x4u
#define MLT_SIZE_16 0x8u

#define MLT_MAX_LOGITS 15 /* CU model: 2+3+4+6; CTU (128) model: 2+3+4 = 9 */

/* mlt_config.flags -- arithmetic of the conv stack (DESIGN.md "Numerics").
 * fast : fp16 operands on the MFMA units, fp32 accumulate.  Its logit error depends on the weight set (rms 1.4e-4 ... 5.5e-4
 *        over seeded weight sets, tails ~4.5x rms) and on the content (exactly-constant areas carry coherent rounding errors).
 * exact: every weight / activation is an fp16 (hi, lo) pair, 3 MFMA passes, ~fp32 accuracy (|dlogit| ~ 1e-5).
 * Defaults: 128x128 -> fast WHEN the load-time calibration says the weight set meets mlt_config.tolerance with it (else a middle
 * tier or exact, see mlt_load_weights); 64/32/16 -> CONFIGURED exact (few pixels per map, so fp16 rounding is not averaged away by the
 * global pooling, and these models are 5-65x cheaper), but the same calibration may keep layer0 -- or one of its two launch units --
 * on the single-pass or hi+lo-weights kernels when the contract still holds with it (mlt_arith_info.exact == 4, .x_units = what
 * stays exact; admission as for the 128 model but with the largest error held to 0.5 x tolerance instead of 0.65 x: their tails are
 * heavier).  Sizes with ANY non-exact unit are protected by two device-side guards (and keep a second, exact copy of the weights
 * resident), applied on EVERY entry point (single, batch, device-pointer, deferred):
 *   flat guard (default on): CUs in which >= 1/8 of the aligned 4-pixel quads are EXACTLY FLAT (org and |org - pred| each constant or
 *                            exactly linear over the quad) or >= 1/2 are NEAR-FLAT (each spans <= 8 ten-bit steps or is linear to
 *                            within one step: +-1 LSB dither, low-contrast texture, ramps) are re-evaluated with the exact
 *                            arithmetic (their fp16 rounding errors are coherent, the global pooling does not average them away);
 *   decision guard (default on since ABI 4 -- split modes are what the encoder consumes, EncCu.cpp:921 -> EncModeCtrl.cpp:110-149;
 *                            MLT_FLAG_NO_DECISION_GUARD turns it off): CUs whose decision-head top-2 margin is below guard_margin
 *                            (default 3 x tolerance) are re-evaluated too, so the split mode handed to
 *                            EncModeCtrl::setNewModeList is the one ~fp32 arithmetic gives.
 * Both cost a second (exact) copy of the weights on the device (11 MB). */
#define MLT_FLAG_EXACT_128 0x1u
#define MLT_FLAG_FAST_SMALL 0x2u       /* single-pass fp16 for 64/32/16 (measurement only: NO seeded weight set meets 1e-3 with it --
                                          2e-3 ... 6e-3 measured -- and it is neither calibrated nor guarded) */
#define MLT_FLAG_DECISION_GUARD 0x4u    /* ABI <= 3: opt-in to the decision guard.  Since ABI 4 the guard is the default: accepted, no effect */
#define MLT_FLAG_NO_FLAT_GUARD 0x8u    /* fast arithmetic without the flat-content guard (measurement only) */
#define MLT_FLAG_NO_CALIBRATION 0x10u  /* keep the fast arithmetic whatever the weight set (measurement only) */
#define MLT_FLAG_EXACT_LITE 0x40u      /* ABI 4 (round 5, measurement): sizes configured exact (MLT_FLAG_EXACT_128; 64/32/16 with MLT_FLAG_NO_CALIBRATION) run the
                                          "exact-lite" arithmetic: Wh*Xh in fp16, the two cross terms Wl*Xh + Wh*Xl as ONE scaled FP8 MFMA per tap and 32 channels
                                          (2 fp16-equivalent MFMAs per product instead of 3; |dlogit| ~ 1/20 of the single pass's) */
#define MLT_FLAG_NO_MAGNITUDE_GUARD 0x80u /* round 6 (measurement only): never admit a tier behind the magnitude guard (mlt_arith_info.mag_guard_thr): the round-5 search */
#define MLT_FLAG_NO_DECISION_GUARD 0x20u /* ABI 4: fast arithmetic without the decision guard (measurement only: a split whose reference margin is
                                          below ~2 x tolerance may then differ from the reference's) */

typedef struct mlt_ctx mlt_ctx;

typedef struct mlt_config {
  uint32_t struct_size;   /* = sizeof(mlt_config) */
  int32_t device;         /* HIP device ordinal (reference: at::kCUDA hard-coded, EncCu.cpp:804) */
  const char *weights_dir;/* directory holding MLTORPQ_splitMode_<S>.mltw, the blob counterpart of the
                             reference's hard-coded ".../torch_model/MLTORPQ_splitMode_<S>.pt"
                             (EncCu.cpp:897-899).  NULL: load later with mlt_load_weights(). */
  uint32_t size_mask;     /* MLT_SIZE_* bits to enable; 0 => MLT_SIZE_128 (reference default, :754) */
  int32_t head_index[4];  /* decision head per size {128,64,32,16}; -1 => reference default:
                             element [2] for 128, [0] otherwise (EncCu.cpp:913-919) */
  int32_t max_batch;      /* largest n passed to mlt_predict_batch*; 0 => 4096 */
  uint32_t flags;         /* MLT_FLAG_* bits, 0 = defaults */
  float guard_margin;     /* decision-guard threshold on (top1 - top2) of the decision head; <= 0 => 3 x tolerance.  Two logits that are each
                             within `tolerance` of the reference move their difference by at most 2 x tolerance; the admission of the fast
                             arithmetic is CALIBRATED (statistical), not proven, and its tail probes saw single logits at up to ~1.1 x tolerance,
                             hence 3 x rather than 2 x: everything below the threshold is re-evaluated exactly */
  float tolerance;        /* |dlogit| contract the fast arithmetic is calibrated against at load time; <= 0 => 1e-3
                             (BASELINE.json north_star) */
  uint32_t reserved;      /* 0 */
  /* ---- since ABI 3.  ABI 4 is a hard break: struct_size must equal sizeof(mlt_config) (the 56-byte ABI-2 struct is rejected with
     MLT_ERR_ARG -- an ABI-2 binary would also hand mlt_arithmetic a 32-byte mlt_arith_info) ---- */
  int32_t n_devices;      /* 0: the single `device` above.  k >= 1: devices[0..k-1] -- ONE context serving k GPUs of the node (SURVEY.md 8e
                             "CTUs within a frame shard across the GPUs"; the encoder is one process): weights are uploaded (and
                             calibrated) once per device from the one host blob, mlt_predict_batch shards its batch contiguously over the
                             devices (one host thread each, no exchange between them), mlt_submit deals CUs round-robin and mlt_flush /
                             mlt_wait / mlt_synchronize / mlt_load_weights / mlt_shutdown address all of them; mlt_predict, the
                             device-pointer entry, mlt_set_stream and the profile calls address devices[0] (use mlt_device_ctx for the
                             others).  Results are bit-identical to a one-device context.  An ordinal may be listed more than once
                             (several contexts on one GPU: how a 1-GPU box exercises this path). */
  int32_t devices[8];
} mlt_config;
#define MLT_MAX_DEVICES 8

/* Create a context: selects the device, allocates workspaces, loads + folds + packs weights
 * ONCE (the reference re-reads the .pt on every CU, EncCu.cpp:894-900).  Natural home:
 * EncCu::init (EncCu.cpp:233-259).  Thread-compatible: one ctx per EncCu / encoder thread. */
int mlt_init(const mlt_config *cfg, mlt_ctx **out);

/* Multi-device contexts: number of devices served (1 for a plain context) and the context of devices[index] -- a full single-device
 * context owned by `ctx` (never shut it down itself), for the calls that address one device (mlt_predict_batch_device with buffers on
 * that GPU, mlt_set_stream, mlt_profile_*, mlt_arithmetic).  index 0 returns ctx. */
int mlt_num_devices(const mlt_ctx *ctx);
mlt_ctx *mlt_device_ctx(mlt_ctx *ctx, int index);

/* Load weights for one CU size from an in-memory MLTW blob (format: weights.py).  Used when the
 * blob arrives over RCCL broadcast instead of from weights_dir.  A size configured for the fast arithmetic is
 * CALIBRATED here: 560 seeded synthetic CUs in six content classes the flat guard does not catch (texture, i.i.d. uniform,
 * constant org / textured pred, textured org / constant pred, a constant band and a near-flat band just under the guard's two thresholds) run
 * through the fast and the exact arithmetic on the device; the fast arithmetic is kept only if
 * 5.5 x (the worst rms|dlogit| pooled per content class and per head) <= tolerance and max|dlogit| (over 5040 logits) <=
 * 0.65 x tolerance (the largest of 295 k probed logits measured up to 1.7 x the largest of these 5040) -- for a set whose largest
 * error exceeds 5 x its overall rms (heavy tail) the 5.5 grows with that ratio, up to 6.5;
 * first with the default realisation of the weights' tap-diffused rounding, then with five others (mlt_arith_info.rounding);
 * otherwise the 128 model tries the middle tiers the same way -- (hi, lo) pairs for the WEIGHTS only (hi fp16; lo fp16, or e4m3 with a
 * per-layer power-of-two scale where the layer has >= 128 input channels: the lo term carries < 2^-11 of the product), on the W2 forms
 * of the fused kernels, in a SUBSET of the four stages: the 15 subsets are priced in the order of
 * their measured cost and the cheapest one that meets the contract is kept (mlt_arith_info.w2_stages); failing those, the tiers that
 * put one to three stages into the exact arithmetic and the others into (hi, lo) weights (.x_stages), cheapest first -- and a size that
 * meets the contract with none of them is priced in the exact-lite arithmetic (mlt_arith_info.exact == 5; what a TRAINED-like weight set lands
 * on: profiles/r05d_trained_family.txt) and runs exact only if that fails too (mlt_arithmetic reports the outcome).  The
 * admission is STATISTICAL (synthetic content, Gaussian-tail factor), not a bound: "within 1e-3" is calibrated, not proven. */
int mlt_load_weights(mlt_ctx *ctx, int size, const void *blob, size_t bytes);

/* Arithmetic a size runs after loading + what the calibration measured.  The caller sets struct_size = sizeof(mlt_arith_info) BEFORE the
 * call (ABI 4); the library writes only the fields that fit into struct_size bytes and fails with MLT_ERR_ARG when struct_size does not
 * even cover the ABI-4 fields below, so a later, longer struct never overruns an older caller's storage. */
typedef struct mlt_arith_info {
  uint32_t struct_size;   /* in: sizeof(mlt_arith_info) of the caller */
  int32_t exact;          /* 0: fast; 1: exact ((hi, lo) pairs for weights and activations); 2: (hi, lo) weights on fp16 activations in
                             every stage; 3: (hi, lo) weights in SOME stages (w2_stages), single pass in the others; 4: the exact arithmetic in the
                             stages of x_stages, (hi, lo) weights in the others (w2_stages); 5 (round 5): exact-lite in every stage -- (hi, lo) pairs
                             with Wh*Xh in fp16 and both cross terms Wl*Xh + Wh*Xl in ONE scaled FP8 MFMA per tap and 32 channels: 2 fp16-equivalent
                             MFMAs per product instead of 3, |dlogit| ~ 1/20 of the single pass's (1e-5 rms); tried after every fp16 tier, before exact */
  int32_t calibrated;     /* 1: the calibration ran for this size */
  float calib_rms, calib_max; /* |dlogit| of the chosen non-exact tier (or of the fast one if exact was chosen) vs exact over the calibration CUs:
                                 worst rms pooled per content class / per head, and the overall maximum */
  int32_t flat_guard, decision_guard;
  uint64_t guard_reruns;  /* CUs re-evaluated by the guards since init */
  int32_t w2_stages;      /* ABI 3: bit s set = layer s (0..3) runs (hi, lo) weights; 0 for the fast and the exact arithmetic */
  float guard_margin;     /* ABI 3: the decision guard's threshold in effect for this size (0 when the guard is off) */
  int32_t x_stages;       /* ABI 3: bit s set = layer s runs the exact arithmetic inside a mixed tier (exact == 4); 0 otherwise */
  int32_t w2_units;       /* ABI 3: w2_stages at launch-unit granularity: bit 2 s = layer s's first unit (layer0.0 / its stride-2 conv + shortcut),
                             bit 2 s + 1 = its second (layer0.1 / its three stride-1 convs) */
  int32_t x_units;        /* ABI 3: x_stages at the same launch-unit granularity */
  int32_t rounding;       /* ABI 3: which realisation of the single-pass weights' tap-diffused rounding the calibration kept (0 = the default) */
  int32_t calib_cus;      /* ABI 4: CUs the last calibration priced (synthetic + caller's, without those the flat guard re-evaluates exactly anyway) */
  int32_t calib_caller_cus; /* ABI 4: ... of which supplied by the caller through mlt_calibrate */
  /* ---- round 6 (appended: written only when struct_size covers them; a 72-byte ABI-4 struct is still accepted) ----
   * MAGNITUDE guard.  The fp16 tiers' error is relative: it scales with M = max over logits of sum_k |w_ck gap_k|, the size of the feature-driven
   * part of the logits (the head's poc / qp / bias terms are exact).  A weight set that amplifies content far outside its training range -- a
   * residual plane of hundreds of ten-bit steps -- produces logits and absolute errors 20-80 x those of ordinary content there; such a set is
   * admitted to a non-exact tier BEHIND this guard: CUs with M > mag_guard_thr are re-evaluated with the exact arithmetic (third guard beside
   * the flat-content and the decision guard, same re-run path, counted in guard_reruns).  The threshold is the largest magnitude (on a
   * quarter-octave grid) up to which the calibration CUs -- the 560 synthetic ones, the caller's, and 320 further in-distribution CUs (texture +
   * 1/f scenes) -- meet the admission rule in its stricter "refinement" form (5.5 .. 6.5 x rms <= 0.95 x, max <= 0.6 x tolerance) with at most
   * 5 % of the in-distribution CUs above it; calib_rms / calib_max are the figures of the CUs at or below it.  0: the tier was admitted by the
   * plain rule, no such guard. */
  float mag_guard_thr;      /* the threshold in effect (0: no magnitude guard: exact arithmetic, small models, MLT_FLAG_NO_MAGNITUDE_GUARD) */
  float mag_guard_flagged;  /* fraction of the in-distribution calibration CUs (texture, 1/f scenes, the caller's) above the threshold: what the guard costs on ordinary content */
  int32_t mag_guard_kind;   /* 2: the tier was admitted BEHIND the guard (above).  1: RANGE guard -- the plain rule admitted the tier, which is still only validated on
                               the magnitudes its calibration CUs had while its error grows linearly with M: mag_guard_thr = 1.5 x the largest calibrated magnitude
                               (1.5 x the 0.65 x tolerance its largest calibration error may reach = the tolerance); nothing inside the calibrated range is ever
                               flagged.  0: none */
} mlt_arith_info;
int mlt_arithmetic(mlt_ctx *ctx, int size, mlt_arith_info *out);

/* ABI 4: calibrate on the INTEGRATOR's content.  mlt_load_weights decides the arithmetic of a size on 560 synthetic CUs generated inside the
 * library; this call repeats that decision -- the same admission rule, the same search (csrc/mlt_tier_search.h) -- with n CUs of the caller's
 * (HOST memory, dense [n][size][size] int16 org / pred as in mlt_predict_batch, int32 poc / qp; e.g. what host/mlt_split_predictor.hpp's call
 * dump recorded from real sequences: tools/calibrate_from_dump.py) APPENDED to the synthetic set (their own content class: the worst pooled
 * rms over classes counts) or REPLACING it.  Caller CUs the flat-content guard re-evaluates exactly anyway are left out of the statistics
 * (mlt_arith_info.calib_caller_cus = those that counted).  MLT_CALIB_REPLACE with fewer than 256 CUs that count (a tiny n, or content the
 * flat guard takes anyway) cannot carry the statistical admission rule: the call then behaves like MLT_CALIB_APPEND (the synthetic set stays;
 * mlt_arith_info.calib_cus > calib_caller_cus tells) -- no arithmetic is ever admitted on an empty or near-empty set.  1 <= n <= 4096.  The size must have been loaded with mlt_load_weights / weights_dir
 * (the library keeps the blob); sizes configured exact (MLT_FLAG_EXACT_128) or loaded with MLT_FLAG_NO_CALIBRATION are left alone.  Every
 * device of a multi-device context is re-calibrated; like a reload it invalidates captured graphs, and the outcome is read with
 * mlt_arithmetic.  On failure the size is unloaded (the caller keeps -1 / full RDO until it loads weights again). */
#define MLT_CALIB_APPEND 0
#define MLT_CALIB_REPLACE 1
int mlt_calibrate(mlt_ctx *ctx, int size, const int16_t *org, const int16_t *pred, const int32_t *poc, const int32_t *qp, int n, int mode);

/* Replaces EncCu.cpp:806-921 for ONE CU: gathers size x size luma from the original and the
 * prediction buffers (Pel = int16, element strides as AreaBuf exposes them, Buffer.h:94-105),
 * absdiff, 1/1023 normalisation, network, argmax.  Synchronous.
 *   split_mode  <- argmax of the decision head (first maximal index, like torch.argmax)
 *   logits_opt  <- all head logits, lvl1..lvlN concatenated (mlt_num_logits(size) floats) or NULL */
int mlt_predict(mlt_ctx *ctx, const int16_t *org, int org_stride, const int16_t *pred, int pred_stride,
                int size, int32_t poc, int32_t qp, int32_t *split_mode, float *logits_opt);

/* n CUs from HOST memory.  org / pred: dense [n][size][size] int16.  Synchronous.  logits may be NULL.
 * Batches larger than one staging sub-chunk (512 CUs, MLT_STAGE_CHUNK) are pipelined: the H2D copy of the next
 * sub-chunk overlaps the kernels of the current one -- effective only when org / pred are pinned (mlt_alloc_pinned). */
int mlt_predict_batch(mlt_ctx *ctx, int n, int size, const int16_t *org, const int16_t *pred,
                      const int32_t *poc, const int32_t *qp, int32_t *split_mode, float *logits);

/* Same with every pointer in DEVICE memory; enqueues on the context's stream; call mlt_synchronize before reading the
 * results.  With a guard active for `size` the call BLOCKS once per chunk until the chunk's fast pass has delivered the NUMBER of
 * flagged CUs (4 bytes) -- it sleeps for the expected duration of the batch and polls only for the last ~0.2 ms (MLT_GUARD_SPIN_WAIT=1
 * polls from the start, MLT_GUARD_BLOCKING_WAIT=1 sleeps on the event) -- then enqueues their exact re-evaluation; without guards it
 * never synchronises.  This is the
 * HBM-resident path bench.py times. */
int mlt_predict_batch_device(mlt_ctx *ctx, int n, int size, const void *d_org, const void *d_pred,
                             const void *d_poc, const void *d_qp, void *d_split_mode, void *d_logits);

/* Deferred single-CU prediction (encoder-side batching, SURVEY.md 8f N3: "async predict + deferred setNewModeList").
 * mlt_submit copies one CU's planes (same arguments as mlt_predict) into pinned staging and returns a ticket at once;
 * nothing runs yet.  mlt_flush launches every CU submitted so far for that size as ONE batch and returns without
 * waiting; mlt_wait returns a ticket's result, flushing first if its batch has not been launched.  An encoder that
 * can postpone EncModeCtrl::setNewModeList for k independent CUs (CTUs of a wavefront, EncCu.cpp:792-800) pays one
 * ~0.2 ms launch for all k instead of k synchronous calls.  Up to MLT_DEFER_CAP CUs per batch (a full batch is flushed
 * by the next submit); a ticket stays valid until two further batches of its size have been started.  Results are
 * bit-identical to mlt_predict (guards included: flagged CUs of a batch are re-evaluated by its first mlt_wait). */
#define MLT_DEFER_CAP 64
typedef uint64_t mlt_ticket;
int mlt_submit(mlt_ctx *ctx, const int16_t *org, int org_stride, const int16_t *pred, int pred_stride,
               int size, int32_t poc, int32_t qp, mlt_ticket *ticket);
int mlt_flush(mlt_ctx *ctx, int size);
int mlt_wait(mlt_ctx *ctx, int size, mlt_ticket ticket, int32_t *split_mode, float *logits_opt);

int mlt_synchronize(mlt_ctx *ctx);

/* Use an existing hipStream_t (e.g. the caller's) instead of the context's own stream; NULL switches back to a
 * stream owned by the context (mlt_predict replays its kernel chain from a hipGraph only on an owned stream). */
int mlt_set_stream(mlt_ctx *ctx, void *hip_stream);

/* Pinned host staging buffers for mlt_predict_batch. */
void *mlt_alloc_pinned(size_t bytes);
void mlt_free_pinned(void *p);

/* Number of logits returned per CU for `size` (9 for 128, 15 for 64/32/16, 0 if unsupported). */
int mlt_num_logits(int size);

/* Per-kernel device timing (HIP events on the context's stream): enable, run, then read.
 * mlt_profile_read fills up to `cap` entries; returns the number of distinct kernels. */
typedef struct mlt_kernel_time {
  char name[48];
  uint32_t launches;
  float total_ms;
  double flops;  /* algorithmic FLOPs summed over those launches */
  double bytes;  /* algorithmic HBM bytes (inputs + outputs + weights once) summed over launches */
} mlt_kernel_time;
int mlt_profile_enable(mlt_ctx *ctx, int on);
int mlt_profile_read(mlt_ctx *ctx, mlt_kernel_time *out, int cap);

const char *mlt_last_error(const mlt_ctx *ctx); /* ctx may be NULL: last init error */
int mlt_abi_version(void);
/* 16 hex digits: sha256 over the sources (fastintercu-vvc_amd/csrc/, sorted by name) this binary was built from -- what bench.py reports as
 * derived.source_sig and what the Python host side checks against the tree before it uses the library ("unsigned-build!!" for a build outside build.py). */
const char *mlt_build_signature(void);

/* Release everything (natural home: EncCu::destroy, EncCu.cpp:160-206). NULL is allowed. */
void mlt_shutdown(mlt_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* MLTCNN_H */
