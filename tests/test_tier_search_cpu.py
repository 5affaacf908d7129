"""CPU: the tier search of the load-time calibration (fastintercu-vvc_amd/csrc/mlt_tier_search.h) over a STUB pricer, through the
mlt_tier_search_run hook of libmltcnn_hip.so (ctypes callback; no HIP call on that path): order of the candidates, the refinement
rule, the one illegal launch-unit pair, forced masks, the small models' prefixes, error propagation.  The device-side pricer
(mlt_api.cpp: DevicePricer) only measures; everything decided at load time is decided by the code under test here."""
import ctypes as C

import pytest

TOL = 1e-3
GOOD = (1.0e-4, 4.0e-4, 4.0)        # 5.5 x rms = 0.55 tol, max = 0.4 tol: within and within_refined
EDGE = (1.78e-4, 6.3e-4, 4.0)       # 5.5 x rms = 0.979 tol, max 0.63 tol: within, NOT within_refined (0.95 / 0.6)
BAD = (3.0e-4, 9.0e-4, 4.0)
W2_ORDER = [0x2, 0x8, 0x1, 0x4, 0xA, 0x3, 0x6, 0x9, 0xC, 0x5, 0xB, 0xE, 0x7, 0xD, 0xF]
X_ORDER = [0x4, 0x8, 0x2, 0xC, 0x1, 0x6, 0xA, 0x5, 0x9, 0xE, 0x3]
CB = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint, C.c_uint, C.c_int, C.POINTER(C.c_float))


def units(stages):
    return sum(3 << (2 * s) for s in range(8) if (stages >> s) & 1)


@pytest.fixture(scope="module")
def run(pkg):
    pkg.build.build_lib()
    lib = pkg.capi.load_library()
    lib.mlt_tier_search_run.argtypes = [C.c_int, C.c_int, C.c_float, C.c_float, C.POINTER(C.c_int), CB, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float)]

    def go(kind, n, table, default=BAD, force=None, max_frac=0.0, fail_at=None, lite=BAD):
        """table: {(w2_units, x_units, rounding): (rms, max, tail)} or a callable; lite: the figures of the exact-lite tier (priced as w2 = x = ~0, recorded as
        the string "lite"; None: the pricer has no such tier) -> (result dict, list of priced configurations)"""
        calls = []

        def cb(_, w2u, xu, r, out):
            if w2u == 0xFFFFFFFF and xu == 0xFFFFFFFF:
                if lite is None:
                    return -1
                calls.append("lite")
                out[0], out[1], out[2] = lite
                return 5 if (fail_at is not None and len(calls) == fail_at) else 0
            calls.append((w2u, xu, r))
            if fail_at is not None and len(calls) == fail_at:
                return 5
            v = table(w2u, xu, r) if callable(table) else table.get((w2u, xu, r), default)
            for i, x in enumerate(v):   # 3 figures, or 9: + (g_valid, g_rms, g_max, g_tail, g_thr, g_flag) = the same configuration behind the magnitude guard
                out[i] = x
            return 0
        res, fig = (C.c_int * 8)(), (C.c_float * 4)()
        f = None
        if force:
            f = (C.c_int * 12)(*[force.get(k, d) for k, d in (("rounding", -1), ("w2_mask", -1), ("x_mask", -1), ("w2_units", -1), ("small_prefix", -1),
                                                            ("no_roundings", 0), ("no_w2", 0), ("no_xmix", 0), ("no_w2_units", 0), ("no_x_units", 0), ("no_lite", 0), ("no_mag_guard", 0))])
        rc = lib.mlt_tier_search_run(kind, n, TOL, max_frac, f, CB(cb), None, res, fig)
        return dict(rc=rc, exact=res[0], w2=res[1], w2_units=res[2], x_units=res[3], rounding=res[4], priced=res[5], lite=res[6], guarded=res[7], rms=fig[0], max=fig[1], tail=fig[2],
                    mag_thr=fig[3]), calls
    return go


def test_single_pass_admitted_prices_once(run):
    r, calls = run(0, 6, {(0, 0, 0): GOOD})
    assert r["rc"] == 0 and (r["exact"], r["w2"], r["w2_units"], r["x_units"], r["rounding"]) == (0, 0, 0, 0, 0)
    assert calls == [(0, 0, 0)] and r["priced"] == 1 and abs(r["max"] - 4e-4) < 1e-9


def test_heavy_tail_raises_the_rms_factor(run):
    # k = clamp(1.1 x tail, 5.5, 6.5): rms 1.6e-4 passes at tail 4 (k = 5.5 -> 0.88 tol) and fails at tail 6 (k = 6.5 -> 1.04 tol)
    assert run(0, 1, {(0, 0, 0): (1.6e-4, 4e-4, 4.0)})[0]["w2"] == 0
    r, calls = run(0, 1, {(0, 0, 0): (1.6e-4, 4e-4, 6.0)}, default=GOOD, force={"no_roundings": 1})
    assert r["w2"] == 1 and calls[1] == (units(W2_ORDER[0]), 0, 0)


def test_another_rounding_realisation_is_tried_before_any_tier(run):
    r, calls = run(0, 6, {(0, 0, 0): BAD, (0, 0, 1): BAD, (0, 0, 2): GOOD})
    assert (r["exact"], r["w2"], r["rounding"]) == (0, 0, 2) and calls == [(0, 0, 0), (0, 0, 1), (0, 0, 2)]
    # none admitted: the tiers are searched on the realisation that came CLOSEST (lowest score), which is priced again first
    table = {(0, 0, 0): BAD, (0, 0, 1): (2.9e-4, 9e-4, 4.0), (0, 0, 2): (2.0e-4, 7e-4, 4.0), (0, 0, 3): BAD, (0, 0, 4): BAD, (0, 0, 5): BAD}
    r, calls = run(0, 6, lambda w, x, rr: table[(0, 0, rr)] if (w, x) == (0, 0) else (GOOD if rr == 2 else BAD))
    assert calls[:7] == [(0, 0, v) for v in range(6)] + [(0, 0, 2)] and r["rounding"] == 2 and r["w2"] == 1
    assert all(c[2] == 2 for c in calls[6:])
    # switched off / forced
    r, calls = run(0, 6, {(0, 0, 0): BAD}, default=GOOD, force={"no_roundings": 1})
    assert calls[1] == (units(0x2), 0, 0) and r["rounding"] == 0
    r, calls = run(0, 6, {(0, 0, 0): GOOD, (0, 0, 4): BAD}, force={"rounding": 4})
    assert calls == [(0, 0, 0), (0, 0, 4)] + calls[2:] and r["rounding"] == 4


def test_hi_lo_weight_stage_subsets_cheapest_first_then_unit_refinement(run):
    # stage masks fail until 0xA (layer1 + layer3); the refinement then drops units in the order chain64(3) | chain256(7) | s2 64->128 ... of the set bits
    ok_masks = {units(0xA)}
    seen = []

    def table(w, x, r):
        seen.append((w, x))
        if (w, x) == (0, 0):
            return BAD
        if w in ok_masks:
            return GOOD
        if x == 0 and w == units(0xA) & ~(1 << 3):   # drop the 64-channel chain: fine
            return GOOD
        return BAD
    r, calls = run(0, 1, table, force={"no_roundings": 1})
    assert [c[0] for c in calls[1:6]] == [units(m) for m in W2_ORDER[:5]]
    w = units(0xA)
    # unit drops tried: 3 (accepted) -> 7 -> [6 is the stride-2 conv in front of a chain still on hi+lo weights: illegal, never priced] -> 2
    assert [c[0] for c in calls[6:]] == [w & ~8, w & ~8 & ~0x80, w & ~8 & ~4]
    assert (r["w2"], r["w2_units"], r["x_units"], r["exact"]) == (1, w & ~8, 0, 0)
    assert abs(r["rms"] - GOOD[0]) < 1e-9          # the figures are those of the configuration kept, not of the last candidate priced


def test_refinements_are_held_to_the_stricter_rule(run):
    w = units(0x2)
    r, calls = run(0, 1, {(0, 0, 0): BAD, (w, 0, 0): GOOD, (w & ~8, 0, 0): EDGE, (w & ~4, 0, 0): EDGE}, force={"no_roundings": 1})
    assert r["w2_units"] == w and calls[2:] == [(w & ~8, 0, 0), (w & ~4, 0, 0)]   # EDGE passes `within` but not `within_refined`: both drops refused
    r, _ = run(0, 1, {(0, 0, 0): BAD, (w, 0, 0): EDGE}, force={"no_roundings": 1})
    assert r["w2"] == 1 and r["w2_units"] == w                                       # ... while a first admission only needs `within`


def test_stride2_unit_in_front_of_a_two_plane_chain_is_never_dropped(run):
    # layer2 admitted whole (units 4, 5).  Drop order reaches unit 5 (chain 128) first: refused here -> unit 4 (its stride-2 conv) is an illegal drop.
    w = units(0x4)
    r, calls = run(0, 1, {(0, 0, 0): BAD, (units(0x2), 0, 0): BAD, (units(0x8), 0, 0): BAD, (units(0x1), 0, 0): BAD, (w, 0, 0): GOOD}, force={"no_roundings": 1})
    assert calls[-1] == (w & ~0x20, 0, 0) and (w & ~0x10, 0, 0) not in calls and r["w2_units"] == w
    # ... and once the chain is back on the single pass the stride-2 conv may follow
    r, calls = run(0, 1, {(0, 0, 0): BAD, (units(0x2), 0, 0): BAD, (units(0x8), 0, 0): BAD, (units(0x1), 0, 0): BAD, (w, 0, 0): GOOD, (0x10, 0, 0): GOOD}, force={"no_roundings": 1})
    assert calls[-1] == (0, 0, 0) and r["w2_units"] == 0x10


def test_tier_below_exact_and_its_three_refinements(run):
    x0 = units(0x8)                      # first two exact-stage candidates: 0x4 fails, 0x8 admits
    w0 = units(0x7)

    def table(w, x, r):
        if (w, x) == (w0, x0):
            return GOOD
        if x == x0 and w == units(0x5):  # greedy stage drop of layer1's hi+lo weights (order 2, 0, 3, 1): layer2 refused, layer0 refused, layer1 accepted
            return GOOD
        if w == units(0x5) | 0x40 and x == 0x80:   # exact unit 6 (stride-2 128->256) back to hi+lo weights, chain 256 (7) stays exact
            return GOOD
        return BAD
    r, calls = run(0, 1, table, force={"no_roundings": 1})
    k = 1 + 15
    assert [c[:2] for c in calls[1:k]] == [(units(m), 0) for m in W2_ORDER]
    assert [c[:2] for c in calls[k:k + 2]] == [(units(0xF & ~0x4), units(0x4)), (w0, x0)]
    assert [c[:2] for c in calls[k + 2:k + 5]] == [(units(0x3), x0), (units(0x6), x0), (units(0x5), x0)]   # stage drops 2, 0, [3 is exact], 1
    tail = [c[:2] for c in calls[k + 5:]]
    w1 = units(0x5)
    # unit drops of the hi+lo stages (order 3 1 7 5 0 4 6 2, set bits only; 4 = stride-2 in front of chain 5 still two-plane: skipped after 5 is refused)
    assert tail[:3] == [(w1 & ~2, x0), (w1 & ~0x20, x0), (w1 & ~1, x0)]
    # exact units back to hi+lo weights (order 3 5 7 1 2 4 6 0, set bits only): 7 refused, 6 accepted
    assert tail[3:] == [(w1 | 0x80, x0 & ~0x80), (w1 | 0x40, x0 & ~0x40)]
    assert (r["exact"], r["w2"], r["w2_units"], r["x_units"]) == (0, 1, w1 | 0x40, 0x80)


def test_nothing_meets_the_contract_runs_exact_with_the_single_pass_figures(run):
    r, calls = run(0, 2, lambda w, x, rr: (2.5e-4, 8e-4, 4.0) if (w, x, rr) == (0, 0, 1) else BAD)
    assert r["exact"] == 1 and r["w2"] == 0 and r["w2_units"] == 0 and r["x_units"] == 0 and r["rounding"] == 1
    assert abs(r["rms"] - 2.5e-4) < 1e-9 and len(calls) == 3 + 15 + 11 + 1 and calls[-1] == "lite" and r["lite"] == 0
    r, calls = run(0, 1, {}, force={"no_w2": 1, "no_roundings": 1})
    assert r["exact"] == 1 and calls == [(0, 0, 0), "lite"]
    r, calls = run(0, 1, {}, force={"no_xmix": 1, "no_roundings": 1, "no_lite": 1})
    assert r["exact"] == 1 and len(calls) == 1 + 15


def test_exact_lite_is_the_last_tier_before_exact(run):
    """Round 5: the exact-lite arithmetic (fp16 hi x hi + both cross terms in one scaled FP8 MFMA) is priced after every fp16 tier has failed -- the
    mixed tiers are faster -- and before the exact arithmetic; a pricer without such a tier (negative return) is simply skipped."""
    r, calls = run(0, 1, {}, force={"no_roundings": 1}, lite=GOOD)
    assert (r["exact"], r["lite"], r["w2"], r["w2_units"], r["x_units"]) == (0, 1, 0, 0, 0) and calls[-1] == "lite" and len(calls) == 1 + 15 + 11 + 1
    assert abs(r["rms"] - GOOD[0]) < 1e-9 and r["priced"] == len(calls)
    r, calls = run(0, 1, {(0, 0, 0): BAD, (units(0x2), 0, 0): GOOD}, force={"no_roundings": 1}, lite=GOOD)
    assert r["lite"] == 0 and r["w2"] == 1 and "lite" not in calls                      # an fp16 tier that passes wins
    r, calls = run(0, 1, {}, force={"no_roundings": 1}, lite=None)
    assert r["exact"] == 1 and r["lite"] == 0 and "lite" not in calls and r["priced"] == len(calls)
    r, calls = run(1, 5, {}, max_frac=0.5, lite=(1e-4, 4.9e-4, 4.0))
    assert (r["exact"], r["lite"]) == (0, 1) and calls[-1] == "lite"                    # the small models: after the prefixes and layer0's variants
    r, calls = run(1, 5, {}, max_frac=0.5, lite=(1e-4, 5.1e-4, 4.0))
    assert (r["exact"], r["lite"]) == (1, 0)                                            # ... under THEIR rule (largest error <= 0.5 x tolerance)


def test_forced_masks_are_kept_whatever_they_measure(run):
    r, calls = run(0, 6, {(0, 0, 0): GOOD}, force={"w2_mask": 0x5})
    assert calls == [(0, 0, 0), (units(0x5), 0, 0)] and (r["w2"], r["w2_units"]) == (1, units(0x5))      # even though the single pass would do
    r, calls = run(0, 6, {}, force={"w2_mask": 0x5, "w2_units": 0x11})
    assert calls[-1] == (0x11, 0, 0) and r["w2_units"] == 0x11
    r, calls = run(0, 6, {}, force={"x_mask": 0x2, "no_roundings": 1})
    assert calls[-1] == (units(0xD), units(0x2), 0) and (r["w2_units"], r["x_units"]) == (units(0xD), units(0x2))


def test_small_models_prefixes_then_layer0_variants(run):
    allm = 0x1F
    rest = units(allm & ~1)
    order = [(0, units(allm & ~((1 << k) - 1))) for k in (4, 3, 2, 1)]
    r, calls = run(1, 5, {}, max_frac=0.5)
    assert [c[:2] for c in calls[:-1]] == order + [(units(1), units(0x1E)), (0, rest | 2), (0, rest | 1)] and calls[-1] == "lite" and r["exact"] == 1
    r, calls = run(1, 5, {(0, units(0x1C), 0): GOOD}, max_frac=0.5)
    assert len(calls) == 3 and (r["exact"], r["x_units"], r["w2_units"]) == (0, units(0x1C), 0)
    r, calls = run(1, 5, {(units(1), units(0x1E), 0): GOOD}, max_frac=0.5)
    assert (r["w2"], r["w2_units"], r["x_units"]) == (1, 3, units(0x1E))
    r, calls = run(1, 5, {(0, rest | 2, 0): GOOD}, max_frac=0.5)
    assert (r["exact"], r["x_units"]) == (0, rest | 2)                       # layer0.0 single pass, layer0.1 exact (the 64 x 64 model's outcome)
    # the small models hold the largest error to 0.5 x tolerance: max 0.6 tol passes the 128 model's rule, not theirs
    assert run(1, 5, {(0, units(0x10), 0): (1e-4, 6e-4, 4.0)}, max_frac=0.5)[0]["x_units"] != units(0x10)
    assert run(1, 5, {(0, units(0x10), 0): (1e-4, 6e-4, 4.0)}, max_frac=0.65)[0]["x_units"] == units(0x10)
    r, calls = run(1, 5, {}, max_frac=0.5, force={"small_prefix": 2})
    assert calls == [(0, units(0x1C), 0)] and r["x_units"] == units(0x1C)


def test_a_failing_pricer_stops_the_search_with_its_code(run):
    r, calls = run(0, 6, {}, fail_at=4)
    assert r["rc"] == 5 and len(calls) == 4
    r, calls = run(0, 1, {}, force={"no_roundings": 1, "no_w2": 1}, fail_at=2)   # ... the exact-lite pricing included
    assert r["rc"] == 5 and calls == [(0, 0, 0), "lite"]
    r, calls = run(1, 5, {}, fail_at=1)
    assert r["rc"] == 5 and len(calls) == 1


def guarded(plain, behind, thr=4.0, flag=0.02):
    return tuple(plain) + (1.0,) + tuple(behind) + (thr, flag)


def test_magnitude_guard_admits_what_the_plain_rule_rejects(run):
    """Round 6: a configuration the plain rule rejects may be admitted BEHIND the magnitude guard when the pricer supplies the figures over the
    CUs below the guard's threshold -- same constants, the plain rule first, at most 5 % of the in-distribution CUs above the threshold."""
    # the single pass, default realisation: priced once, admitted behind the guard; the choice reports the guarded figures and the threshold
    r, calls = run(0, 6, {(0, 0, 0): guarded(BAD, GOOD, thr=4.0)})
    assert (r["exact"], r["w2"], r["lite"], r["guarded"]) == (0, 0, 0, 1) and calls == [(0, 0, 0)]
    assert abs(r["mag_thr"] - 4.0) < 1e-6 and abs(r["max"] - GOOD[1]) < 1e-9 and abs(r["rms"] - GOOD[0]) < 1e-9
    # the plain rule has priority: a configuration within it is admitted without the guard even when guarded figures come along
    r, _ = run(0, 6, {(0, 0, 0): guarded(GOOD, GOOD)})
    assert (r["guarded"], r["mag_thr"]) == (0, 0.0)
    # too many in-distribution CUs above the threshold (> 5 %): not admitted, the search goes on as before
    r, calls = run(0, 6, {(0, 0, 0): guarded(BAD, GOOD, flag=0.2)}, default=GOOD, force={"no_roundings": 1})
    assert r["guarded"] == 0 and r["w2"] == 1 and calls[1] == (units(W2_ORDER[0]), 0, 0)
    # guarded figures outside the rule, a threshold of zero, or figures flagged invalid: no admission
    for v in (guarded(BAD, BAD), guarded(BAD, GOOD, thr=0.0), BAD + (0.0,) + GOOD + (4.0, 0.0)):
        r, _ = run(0, 1, {(0, 0, 0): v}, lite=BAD)
        assert (r["exact"], r["guarded"]) == (1, 0)
    # switched off (MLT_FLAG_NO_MAGNITUDE_GUARD / the small models): the round-5 search
    r, _ = run(0, 1, {(0, 0, 0): guarded(BAD, GOOD)}, lite=BAD, force={"no_mag_guard": 1})
    assert (r["exact"], r["guarded"]) == (1, 0)
    # another rounding realisation behind the guard is taken before any hi+lo-weights tier
    r, calls = run(0, 6, {(0, 0, 0): BAD, (0, 0, 1): guarded(BAD, GOOD, thr=5.5)})
    assert (r["rounding"], r["guarded"], r["w2"]) == (1, 1, 0) and calls == [(0, 0, 0), (0, 0, 1)] and abs(r["mag_thr"] - 5.5) < 1e-6


def test_refinements_of_a_guarded_tier_are_judged_behind_the_guard(run):
    """A hi+lo-weights stage subset admitted behind the guard: its unit drops are held to the REFINED rule on the guarded figures, and the threshold
    reported is the one of the configuration that is kept (every configuration has its own: the threshold follows its worst relative error)."""
    base = units(0x2)                                   # layer1 in hi+lo weights: units 2 and 3
    def table(w, x, r):
        if (w, x) == (0, 0):
            return BAD
        if w == base:
            return guarded(BAD, GOOD, thr=6.0)
        if w == base & ~(1 << 3):                       # first drop (unit 3, the 64-channel chain): refined rule holds behind the guard
            return guarded(BAD, GOOD, thr=5.0)
        if w == 0 and x == 0:
            return BAD
        return guarded(BAD, EDGE, thr=4.5)              # further drops: within, but not within the refined rule
    r, calls = run(0, 1, table)
    assert (r["w2"], r["guarded"]) == (1, 1) and r["w2_units"] == base & ~(1 << 3) and abs(r["mag_thr"] - 5.0) < 1e-6
    # a refinement that passes the PLAIN refined rule but brings no guarded figures is not taken while the tier sits behind the guard
    def table2(w, x, r):
        if (w, x) == (0, 0):
            return BAD
        return guarded(BAD, GOOD, thr=6.0) if w == base else GOOD
    r, _ = run(0, 1, table2)
    assert (r["w2"], r["guarded"], r["w2_units"]) == (1, 1, base) and abs(r["mag_thr"] - 6.0) < 1e-6
