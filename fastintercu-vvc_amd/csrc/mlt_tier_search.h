// mlt_tier_search.h -- which arithmetic a weight set runs, as a SEARCH over an abstract pricer (round 5: split out of mlt_load_weights so
// that the order of the candidates, the refinement rule, the one illegal unit pair and the forced masks are testable on the CPU with a
// stub pricer -- tests/test_tier_search_cpu.py drives mlt_tier_search_run through ctypes; no HIP call in this file).
//
// Vocabulary (DESIGN.md "Numerics"): a model is four (128 x 128: MltCnnL3ORPQv4, mlt_ctu_or_pq_arch.py:239-256) or five (64 / 32 / 16:
// MltCnnL4ORPQv4, mlt_cu_or_pq_arch.py:59-79) stages of two LAUNCH UNITS each -- bit 2 s = the first unit of layer s (layer0.0 / the
// stride-2 conv + shortcut), bit 2 s + 1 = the second (layer0.1 / the three stride-1 convs).  A unit runs the single fp16 pass, hi+lo
// WEIGHTS (w2_units) or the exact (hi, lo)-pair arithmetic (x_units).  price(w2_units, x_units, rounding) runs the calibration CUs
// through that configuration on the device and returns three figures of |dlogit| against the exact arithmetic.
//
// The constants below are FROZEN (they were fitted to nine seeded weight sets; round 5 adds no new rule -- new weight
// families and caller-supplied content are probed against them as they are).
#pragma once
#include <algorithm>

namespace mlt {

struct TierPrice {
  float rms = 0.f;   // worst rms pooled per content class / per head
  float max = 0.f;   // largest |dlogit| over the calibration logits
  float tail = 0.f;  // max / (rms pooled over everything)
  // Round 6 -- the same configuration behind the MAGNITUDE guard (mlt_api.cpp, "magnitude guard"): the pricer fills these when the plain
  // figures above miss the contract and it has such a guard.  g_thr = the largest logit magnitude (quarter-octave grid) up to which the
  // calibration CUs meet the REFINED rule; CUs above it are re-evaluated exactly at run time, so g_rms / g_max / g_tail are the figures over
  // the CUs at or below it -- the synthetic set's, the caller's, and a further in-distribution set (texture + 1/f scenes) that restores the
  // sample size; g_flag = the fraction of the in-distribution CUs above the threshold (the price of the guard on ordinary content: each
  // costs an exact re-run).
  bool g_valid = false;
  float g_rms = 0.f, g_max = 0.f, g_tail = 0.f, g_thr = 0.f, g_flag = 0.f;
  TierPrice guarded() const { TierPrice p; p.rms = g_rms; p.max = g_max; p.tail = g_tail; return p; }
};

struct TierPricer {
  virtual ~TierPricer() {}
  // rounding: which realisation of the single-pass weights' tap-diffused rounding (0 = the default)
  virtual int price(unsigned w2_units, unsigned x_units, int rounding, TierPrice &out) = 0;
  // the whole model in the exact-lite arithmetic (round 5: Wh*Xh in fp16, both cross terms in one scaled FP8 MFMA; |dlogit| ~ 1/20 of the single
  // pass's, 1.15 x the exact arithmetic's rate).  > 0: an error code; < 0: this pricer has no such tier (the search skips it)
  virtual int price_lite(TierPrice &out) { (void)out; return -1; }
};

struct TierRules {
  float tolerance = 1e-3f;
  float k_min = 5.5f, k_max = 6.5f, k_tail = 1.1f;   // k = clamp(k_tail x tail, k_min, k_max): k x rms <= tolerance
  float max_frac = 0.65f;                            // max <= max_frac x tolerance (128 model; the small models: 0.5)
  float refine_rms_frac = 0.95f, refine_max_frac = 0.6f;   // refinements of an admitted configuration (greedy drops) are held to a stricter rule
  float guard_flag_max = 0.05f;                      // round 6: a configuration behind the magnitude guard may send at most this fraction of the in-distribution
                                                     // calibration CUs to the exact re-run (5 % of a batch at the exact arithmetic's rate costs ~20 % of the step)
  float k(const TierPrice &p) const { return std::min(k_max, std::max(k_min, k_tail * p.tail)); }
  bool within(const TierPrice &p) const { return k(p) * p.rms <= tolerance && p.max <= max_frac * tolerance; }
  bool within_refined(const TierPrice &p) const { return k(p) * p.rms <= refine_rms_frac * tolerance && p.max <= refine_max_frac * tolerance; }
  // the same two rules on the figures behind the magnitude guard (same constants; the guard's threshold itself comes from max_frac x tolerance)
  bool within_guarded(const TierPrice &p) const { return p.g_valid && p.g_thr > 0.f && p.g_flag <= guard_flag_max && within(p.guarded()); }
  bool within_refined_guarded(const TierPrice &p) const { return p.g_valid && p.g_thr > 0.f && p.g_flag <= guard_flag_max && within_refined(p.guarded()); }
  // how far a configuration is from the line: the larger of its two admission figures, relative to their limits
  float score(const TierPrice &p) const { return std::max(k(p) * p.rms / tolerance, p.max / (max_frac * tolerance)); }
};

// tuning switches (MLT_TUNING=1 environment, mlt_api.cpp): -1 / false = not forced
struct TierForce {
  int rounding = -1;       // MLT_ROUNDING: price exactly this realisation and keep it
  int w2_mask = -1;        // MLT_W2_MASK: exactly this STAGE mask in hi+lo weights, kept whatever it measures
  int x_mask = -1;         // MLT_X_MASK: exactly this stage mask exact, the others hi+lo weights
  int w2_units = -1;       // MLT_W2_UNITS: exactly this unit mask in hi+lo weights
  int small_prefix = -1;   // MLT_SMALL_PREFIX: small models, exactly the prefix of k single-pass stages
  bool no_roundings = false, no_w2 = false, no_xmix = false, no_w2_units = false, no_x_units = false, no_lite = false;
  bool no_mag_guard = false;   // MLT_NO_MAG_GUARD: never admit a configuration behind the magnitude guard (the round-5 search)
};

struct TierChoice {
  bool exact = false;      // no candidate meets the contract: the whole model in the exact arithmetic
  bool lite = false;       // no fp16 tier meets the contract, the exact-lite arithmetic does: the whole model runs it (tried last: the mixed tiers are faster)
  bool w2 = false;         // a middle tier was admitted (w2_units / x_units say where)
  unsigned w2_units = 0, x_units = 0;
  int rounding = 0;        // realisation of the single-pass weights the search ended on
  TierPrice price;         // of the chosen non-exact tier (behind the magnitude guard: the guarded figures); when exact: of the single pass (128) / of the last candidate priced (small)
  int priced = 0;          // number of price() calls
  float mag_thr = 0.f;     // round 6: > 0 = the chosen tier was admitted behind the magnitude guard with this threshold
  float mag_flag = 0.f;    // ... which flags this fraction of the in-distribution calibration CUs
};

inline unsigned units_of_stages(unsigned stages) {
  unsigned u = 0;
  for (int s = 0; s < 8; ++s) if ((stages >> s) & 1u) u |= 3u << (2 * s);
  return u;
}
inline unsigned stages_of_units(unsigned units) {
  unsigned st = 0;
  for (int s = 0; s < 8; ++s) if ((units >> (2 * s)) & 3u) st |= 1u << s;
  return st;
}

// Candidate tables of the 128 model, cheapest first (measured ms per 4096 CUs, round 4):
//   a stage in hi+lo weights adds   layer0 0.88, layer1 0.63, layer2 0.98, layer3 0.86   -> the 15 subsets in the order of their sums
//   an exact stage over its hi+lo form adds layer0 3.8, layer1 2.3, layer2 1.65, layer3 1.7
static const unsigned kW2StageOrder[15] = {0x2, 0x8, 0x1, 0x4, 0xA, 0x3, 0x6, 0x9, 0xC, 0x5, 0xB, 0xE, 0x7, 0xD, 0xF};
static const unsigned kXStageOrder[11] = {0x4, 0x8, 0x2, 0xC, 0x1, 0x6, 0xA, 0x5, 0x9, 0xE, 0x3};
static const int kW2StageDrop[4] = {2, 0, 3, 1};                // greedy drop of hi+lo stages behind exact stages, most expensive first
static const int kW2UnitDrop[8] = {3, 1, 7, 5, 0, 4, 6, 2};     // chain 64 | layer0.1 | chain 256 | chain 128 | layer0.0 | s2 64->128 | s2 128->256 | s2 32->64
static const int kXUnitDrop[8] = {3, 5, 7, 1, 2, 4, 6, 0};      // exact units back to hi+lo weights: the stride-1 chains first, then the stride-2 convs

// layer2 / layer3: a single-pass stride-2 conv in FRONT of a hi+lo-weights (or exact) chain is not a legal pair -- its large launches would
// run the stand-alone single-pass kernel, whose accumulation order is not the one of its small-launch variant (which follows the
// whole-stage kernel): the entry points would differ in the last bits
inline bool illegal_w2_unit_drop(int unit, unsigned w2_units, unsigned x_units) {
  return (unit == 4 || unit == 6) && ((w2_units | x_units) & (1u << (unit + 1)));
}

// The 128 x 128 model.  n_roundings: realisations of the tap-diffused rounding the packer offers (mlt_model.h: MLT_N_ROUNDINGS).
inline int search_tier_128(TierPricer &pr, const TierRules &R, const TierForce &F, int n_roundings, TierChoice &out) {
  out = TierChoice();
  int rc;
  auto price = [&](unsigned w2u, unsigned xu, int r, TierPrice &p) { ++out.priced; return pr.price(w2u, xu, r, p); };
  // Admission: the plain rule first; failing that -- round 6 -- the same rule on the figures behind the magnitude guard (when the pricer supplies
  // them).  Once a configuration has been admitted behind the guard, its refinements are judged behind the guard too (`guarded`), each with
  // its own threshold; `thr` follows the configuration that is kept.
  bool guarded = false;
  float thr = 0.f, flag = 0.f;
  auto admit = [&](const TierPrice &p) -> bool {
    if (R.within(p)) { guarded = false; thr = 0.f; flag = 0.f; return true; }
    if (!F.no_mag_guard && R.within_guarded(p)) { guarded = true; thr = p.g_thr; flag = p.g_flag; return true; }
    return false;
  };
  auto admit_refined = [&](const TierPrice &p) -> bool {
    if (!guarded) return R.within_refined(p);
    if (R.within_refined_guarded(p)) { thr = p.g_thr; flag = p.g_flag; return true; }
    return false;
  };
  auto figures = [&](const TierPrice &p) { return guarded ? p.guarded() : p; };
  TierPrice P;
  if ((rc = price(0, 0, 0, P))) return rc;
  int r = 0;
  const bool force_mask = F.w2_mask >= 0, force_x = F.x_mask >= 0, force_units = F.w2_units >= 0, force_rounding = F.rounding >= 0;
  // 1. the single pass with another REALISATION of the weights' rounding (draws of one error distribution: a set a little over the line may
  //    have one under it); none admitted -> the lower tiers are searched on the realisation that came closest
  if ((!admit(P) || force_rounding) && !F.no_roundings && !force_mask) {
    float best_score = R.score(P);
    int best_v = 0;
    bool got = false;
    for (int v = 1; v < n_roundings && !got; ++v) {
      if (force_rounding) v = std::min(std::max(F.rounding, 0), n_roundings - 1);
      TierPrice Pv;
      if ((rc = price(0, 0, v, Pv))) return rc;
      got = admit(Pv) || force_rounding;
      if (got) { r = v; P = Pv; }
      else if (R.score(Pv) < best_score) { best_score = R.score(Pv); best_v = v; }
      if (force_rounding) break;
    }
    if (!got) {
      r = best_v;
      if ((rc = price(0, 0, r, P))) return rc;  // (figures and tail ratio of the kept realisation again; the pricer switches back to it)
    }
  }
  out.rounding = r;
  out.price = P;
  if (admit(P) && !force_mask) { out.price = figures(P); out.mag_thr = thr; out.mag_flag = flag; return 0; }  // the single pass (plain, or behind the magnitude guard)
  const TierPrice P1 = P;
  bool ok = false;
  unsigned w2_mask = 0, x_mask = 0, w2u = 0, xu = 0;
  // 2. hi+lo WEIGHTS in a subset of the stages, the 15 subsets cheapest first
  if (!F.no_w2) {
    for (int k = 0; k < 15 && !ok; ++k) {
      const unsigned mask = force_mask ? ((unsigned)F.w2_mask & 0xFu) : kW2StageOrder[k];
      if (mask == 0) break;
      if ((rc = price(units_of_stages(mask), 0, r, P))) return rc;
      if ((ok = admit(P) || force_mask)) { w2_mask = mask; w2u = units_of_stages(mask); }  // (a forced mask is kept whatever it measures: knock-out timing builds)
      if (force_mask) break;
    }
  }
  // 3. the tier below exact: some stages exact, the others hi+lo weights -- cheapest first; then the hi+lo stages behind them dropped greedily
  if (!ok && !F.no_w2 && !F.no_xmix && (!force_mask || force_x)) {
    for (int k = 0; k < 11 && !ok; ++k) {
      const unsigned xm = force_x ? ((unsigned)F.x_mask & 0xFu) : kXStageOrder[k];
      if (xm == 0 || xm == 0xFu) break;
      if ((rc = price(units_of_stages(0xFu & ~xm), units_of_stages(xm), r, P))) return rc;
      if ((ok = admit(P) || force_x)) { w2_mask = 0xFu & ~xm; w2u = units_of_stages(w2_mask); x_mask = xm; xu = units_of_stages(xm); }
      if (force_x) break;
    }
    if (ok && !force_x) {
      TierPrice kept = P;
      for (int k = 0; k < 4; ++k) {
        const unsigned bit = 1u << kW2StageDrop[k];
        if (!(w2_mask & bit)) continue;
        TierPrice Pd;
        if ((rc = price(units_of_stages(w2_mask & ~bit), xu, r, Pd))) return rc;
        if (admit_refined(Pd)) { w2_mask &= ~bit; kept = Pd; }
      }
      w2u = units_of_stages(w2_mask);
      P = kept;
    }
  }
  // 4. ... at launch-unit granularity: drop the hi+lo weights unit by unit, the largest saving first
  if (ok && !F.no_w2_units && ((!force_mask && !force_x) || force_units)) {
    if (force_units) {
      w2u = (unsigned)F.w2_units & 0xFFu & ~xu;
      if ((rc = price(w2u, xu, r, P))) return rc;
    } else {
      TierPrice kept = P;
      for (int k = 0; k < 8; ++k) {
        const unsigned bit = 1u << kW2UnitDrop[k];
        if (!(w2u & bit)) continue;
        if (illegal_w2_unit_drop(kW2UnitDrop[k], w2u, xu)) continue;
        TierPrice Pd;
        if ((rc = price(w2u & ~bit, xu, r, Pd))) return rc;
        if (admit_refined(Pd)) { w2u &= ~bit; kept = Pd; }
      }
      P = kept;
    }
  }
  // 5. ... and the exact units back to their hi+lo-weights form, unit by unit
  if (ok && xu && !F.no_x_units && !force_x && !force_mask && !force_units) {
    TierPrice kept = P;
    for (int k = 0; k < 8; ++k) {
      const unsigned bit = 1u << kXUnitDrop[k];
      if (!(xu & bit)) continue;
      TierPrice Pd;
      if ((rc = price(w2u | bit, xu & ~bit, r, Pd))) return rc;
      if (admit_refined(Pd)) { xu &= ~bit; w2u |= bit; kept = Pd; }
    }
    P = kept;
  }
  (void)x_mask;
  if (ok) { out.w2 = true; out.w2_units = w2u; out.x_units = xu; out.price = figures(P); out.mag_thr = guarded ? thr : 0.f; out.mag_flag = guarded ? flag : 0.f; return 0; }
  // 6. (round 5) the exact-lite arithmetic everywhere, before the exact one
  if (!F.no_lite && !force_mask && !force_x) {
    TierPrice PL;
    ++out.priced;
    rc = pr.price_lite(PL);
    if (rc > 0) return rc;
    if (rc == 0 && R.within(PL)) { out.lite = true; out.price = PL; return 0; }
    if (rc < 0) --out.priced;
  }
  out.exact = true; out.price = P1;
  return 0;
}

// The 64 / 32 / 16 models: configured exact; the calibration may keep a PREFIX of stages -- or one launch unit of layer0 -- off the exact
// arithmetic (their time is in the first stages, their error in the last ones).  R.max_frac = 0.5 here (mlt_api.cpp).
inline int search_tier_small(TierPricer &pr, const TierRules &R, const TierForce &F, int n_stages, TierChoice &out) {
  out = TierChoice();
  int rc;
  auto price = [&](unsigned w2u, unsigned xu, TierPrice &p) { ++out.priced; return pr.price(w2u, xu, 0, p); };
  const unsigned all = (1u << n_stages) - 1u;
  const bool force_k = F.small_prefix >= 0;
  bool ok = false;
  TierPrice P;
  for (int k = n_stages - 1; k >= 1 && !ok; --k) {  // the single pass in stages 0 .. k-1, exact from stage k on: longest prefix (cheapest) first
    const int kk = force_k ? std::min(std::max(F.small_prefix, 1), n_stages - 1) : k;
    const unsigned xm = all & ~((1u << kk) - 1u);
    if ((rc = price(0, units_of_stages(xm), P))) return rc;
    ok = R.within(P) || force_k;
    if (ok) { out.x_units = units_of_stages(xm); }
    if (!ok && !F.no_w2 && !(xm & 1u) && (xm | 1u) == all) {  // only layer0 off the exact arithmetic: also with hi+lo WEIGHTS there
      if ((rc = price(units_of_stages(1u), units_of_stages(xm), P))) return rc;
      if ((ok = R.within(P))) { out.x_units = units_of_stages(xm); out.w2_units = units_of_stages(1u); out.w2 = true; }
    }
    if (force_k) break;
  }
  if (!ok && !force_k) {  // HALF of layer0: layer0.1 exact (layer0.0 single pass) | layer0.0 exact
    const unsigned rest = units_of_stages(all & ~1u);
    static const unsigned half[2] = {0x2u, 0x1u};
    for (int h = 0; h < 2 && !ok; ++h) {
      if ((rc = price(0, rest | half[h], P))) return rc;
      if ((ok = R.within(P))) out.x_units = rest | half[h];
    }
  }
  out.price = P;
  if (!ok) {
    out.w2 = false; out.w2_units = out.x_units = 0;
    if (!F.no_lite && !force_k) {  // (round 5) the exact-lite arithmetic before the exact one
      TierPrice PL;
      ++out.priced;
      rc = pr.price_lite(PL);
      if (rc > 0) return rc;
      if (rc == 0 && R.within(PL)) { out.lite = true; out.price = PL; return 0; }
      if (rc < 0) --out.priced;
    }
    out.exact = true;
  }
  return 0;
}

}  // namespace mlt

// CPU test hook (exported from libmltcnn_hip.so, NOT part of include/mltcnn.h): the search over a caller-supplied pricer.
//   kind 0: search_tier_128 (n = realisations of the rounding), kind 1: search_tier_small (n = stages)
//   price_cb(user, w2_units, x_units, rounding, out[9] = {rms, max, tail, g_valid, g_rms, g_max, g_tail, g_thr, g_flag}) -> 0 or an error code (returned
//   unchanged; out arrives zeroed, so a callback that writes the first three figures prices a tier without a magnitude guard); the exact-lite tier is
//   priced as price_cb(user, ~0u, ~0u, 0, out), where a NEGATIVE return means "no such tier"
//   force[12] = {rounding, w2_mask, x_mask, w2_units, small_prefix, no_roundings, no_w2, no_xmix, no_w2_units, no_x_units, no_lite, no_mag_guard}; NULL = nothing forced
//   result[8] <- {exact, w2, w2_units, x_units, rounding, priced, lite, behind the magnitude guard}; figures[4] <- {rms, max, tail, magnitude threshold} of the choice
extern "C" int mlt_tier_search_run(int kind, int n, float tolerance, float max_frac, const int *force,
                                   int (*price_cb)(void *user, unsigned w2_units, unsigned x_units, int rounding, float *out3), void *user,
                                   int *result, float *figures);
