"""GPU (MI355X): the HIP path through the C ABI against (a) the committed reference fixtures and
(b) the C oracle on seeded inputs.  Tolerance written here, from BASELINE.json's north_star:
|dlogit| <= 1e-3 and identical split decision wherever the reference's own top-2 margin exceeds
the tolerance."""
import os

import numpy as np
import pytest

from helpers import SIZES, check_splits, decisive, head_slices, load_golden, materialise, variant_state_dict

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-3  # north_star: logits within 1e-3 of the reference
# Default arithmetic (DESIGN.md "Numerics"): 64/32/16 run "exact" (fp16 hi+lo pairs, 3 passes); 128 runs "fast" (single
# fp16 pass) when the load-time calibration finds the weight set within tolerance, else exact, and CUs with large
# exactly-constant areas (coherent fp16 rounding errors) are re-evaluated with the exact arithmetic by a device-side guard.
# EVERY fixture is held to LOGIT_TOL in every shipped configuration; only the measurement-only configuration without guards
# and calibration (test_golden_fixtures_128_raw_fast_arithmetic) documents what the raw single-pass arithmetic does.


@pytest.fixture(scope="module")
def gpu(pkg):
    import torch
    assert torch.cuda.is_available(), "these tests need the MI355X"
    pkg.build.build_lib()
    return pkg


def _ctx(pkg, size, blob, **kw):
    return pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob}, **kw)


def _run_golden(pkg, size, flags, tol_of, guarded=None):
    """guarded: every split is compared (helpers.check_splits) -- None: decide from the configuration (decision guard in `flags`, or an
    arithmetic that is exact anyway)."""
    golden = load_golden(size)
    worst = {}
    undecided = 0
    for case in golden["cases"]:
        blob, org, pred, poc, qp, exp, exp_arg = materialise(pkg, golden, case)
        m = _ctx(pkg, size, blob, flags=flags)
        split, logits = m.predict_batch(org, pred, poc, qp)
        err = float(np.abs(logits - exp).max())
        worst[case["name"]] = err
        a = m.arithmetic(size)
        if size == 128:
            worst[case["name"] + ":arith"] = ("exact" if a["exact"] == 1 else f"tier {a['exact']} stages 0x{a['w2_stages']:x}") + (f" calib rms {a['calib_rms']:.1e} max {a['calib_max']:.1e} reruns {a['guard_reruns']}" if a["calibrated"] else "")
        tol = tol_of(case["name"])
        assert err <= tol, f"{size}/{case['name']}: |dlogit| {err:.3e} > {tol:.0e}"
        dec = 2 if size == 128 else 0
        sl = head_slices([2, 3, 4] if size == 128 else [2, 3, 4, 6])[dec]
        g = guarded if guarded is not None else (a["exact"] == 1 or a["decision_guard"] == 1)
        undecided += check_splits(split, exp, [r[dec] for r in exp_arg], sl, g and tol <= LOGIT_TOL, tol, f"{size}/{case['name']}")
        m.close()
    print(size, "flags", flags, f"{undecided} CUs whose reference margin is below what the configuration can decide",
          {k: (f"{v:.1e}" if isinstance(v, float) else v) for k, v in worst.items()})
    return undecided


@pytest.mark.parametrize("size", SIZES)
def test_golden_fixtures_default_mode(gpu, size):
    """flags = 0 is the SHIPPED configuration (ABI 4): calibrated arithmetic + flat-content guard + decision guard.  EVERY split of EVERY
    fixture -- the near-tie family of round 4 included, whose top-2 margins lie between 1e-5 and 2e-3 -- equals the reference's argmax; the
    only CUs that cannot be decided are those the reference's own fp32 arithmetic ties to within 4e-5 (counted; the exact-tie fixture
    is such a case by construction and resolves to the first index like torch.argmax)."""
    undecided = _run_golden(gpu, size, 0, lambda name: LOGIT_TOL, guarded=True)
    if size == 128:
        assert 2 <= undecided <= 10, undecided   # the 2 exact-tie CUs + the near-tie CUs whose reference margin lands inside +-4e-5 (5 in the round-4 fixtures)


def test_golden_fixtures_128_without_the_decision_guard(gpu):
    """Measurement configuration (MLT_FLAG_NO_DECISION_GUARD; ABI <= 3's default): logits within the tolerance, splits equal wherever the
    reference's own margin exceeds 2 x tolerance -- all an arithmetic with errors up to the tolerance can promise."""
    _run_golden(gpu, 128, gpu.capi.FLAG_NO_DECISION_GUARD, lambda name: LOGIT_TOL, guarded=False)


def test_golden_fixtures_128_fast_arithmetic_with_guards(gpu):
    """The configuration bench.py times with seed-10 weights -- single fp16 pass + flat-content guard -- forced
    (MLT_FLAG_NO_CALIBRATION) for every fixture.  Weight sets the load-time calibration ADMITS to the fast arithmetic must meet
    1e-3 with it; the sets it sends to the exact arithmetic are the ones where single-pass fp16 is marginal (seed 22: 0.8 - 1.1e-3
    depending on the accumulation order), which is what the calibration is for -- they are only held to the measurement bound."""
    pkg = gpu
    golden = load_golden(128)
    admitted = {}
    for case in golden["cases"]:
        key = (case["weight_seed"], case["variant"])
        if key not in admitted:
            blob = materialise(pkg, golden, case)[0]
            m = _ctx(pkg, 128, blob)
            admitted[key] = m.arithmetic(128)["exact"] == 0
            m.close()
    assert admitted[(10, "plain")], "the bench weight set must calibrate to the fast arithmetic"
    by_name = {c["name"]: admitted[(c["weight_seed"], c["variant"])] for c in golden["cases"]}
    print("calibration admits:", by_name)
    _run_golden(pkg, 128, pkg.capi.FLAG_NO_CALIBRATION, lambda name: LOGIT_TOL if by_name[name] else 5e-3)


def test_golden_fixtures_128_raw_fast_arithmetic(gpu):
    """Measurement only: no calibration, no guards.  Documents what the guards are for: the constant-picture fixture exceeds
    1e-3 (1.3e-3 measured) because every pixel carries the same rounding error."""
    _run_golden(gpu, 128, gpu.capi.FLAG_NO_CALIBRATION | gpu.capi.FLAG_NO_FLAT_GUARD | gpu.capi.FLAG_NO_DECISION_GUARD, lambda name: 5e-3)


def test_golden_fixtures_128_exact_mode(gpu):
    _run_golden(gpu, 128, gpu.capi.FLAG_EXACT_128, lambda name: LOGIT_TOL)


@pytest.mark.parametrize("size", (64, 32, 16))
def test_golden_fixtures_small_fast_mode(gpu, size):
    """The single-pass fp16 arithmetic on the small models: same code path as 128, looser bound (few pixels per
    map => fp16 rounding is not averaged away)."""
    _run_golden(gpu, size, gpu.capi.FLAG_FAST_SMALL, lambda name: 1e-2)


@pytest.mark.parametrize("seed", [10, 21, 11])
def test_content_classes_against_oracle(gpu, seed):
    """Round 3: the content real video is full of and the guard used not to see -- a constant band just under the
    guard's 1/8 threshold, one plane constant / the other textured, ramps, +-1 LSB dither, low-contrast texture, zero residual on
    flat org -- 16 CUs per class against the C oracle at LOGIT_TOL, in the shipped configuration (whatever tier the calibration
    picks) AND, for sets the calibration admits to the single-pass arithmetic, with the calibration switched off."""
    from oracle import Oracle
    pkg = gpu
    size, n = 128, 16
    S = pkg.synth
    kinds = [S.KIND_PARTIAL_FLAT, S.KIND_ORG_FLAT_PRED_TEX, S.KIND_ORG_TEX_PRED_FLAT, S.KIND_RAMP, S.KIND_DITHER, S.KIND_LOW_CONTRAST, S.KIND_FLAT_ZERO_RESI]
    blob = pkg.weights.synthetic_blob(0, seed)
    orc = Oracle(blob)
    m = _ctx(pkg, size, blob)
    tier = m.arithmetic(size)["exact"]
    NDG = pkg.capi.FLAG_NO_DECISION_GUARD
    ctxs = [("default", m), ("no decision guard", _ctx(pkg, size, blob, flags=NDG))] + \
           ([("no calibration", _ctx(pkg, size, blob, flags=pkg.capi.FLAG_NO_CALIBRATION | NDG))] if tier == 0 else [])
    sl = head_slices(orc.head_classes)[2]
    report = {}
    for kind in kinds:
        org, pred = S.make_patches(size, n, 7000 + kind, kind)
        poc, qp = S.make_scalars(n, 7000 + kind)
        ref, ref_split = orc.forward(org, pred, poc, qp, threads=8)
        for name, c in ctxs:
            r0 = c.arithmetic(size)["guard_reruns"]
            split, logits = c.predict_batch(org, pred, poc, qp)
            err = float(np.abs(logits - ref).max())
            report[(S.KIND_NAMES[kind], name)] = (f"{err:.1e}", c.arithmetic(size)["guard_reruns"] - r0)
            assert err <= LOGIT_TOL, (S.KIND_NAMES[kind], name, err)
            check_splits(split, ref, ref_split, sl, name == "default", LOGIT_TOL, (S.KIND_NAMES[kind], name))
    print(f"seed {seed} (tier {tier}):", report)
    if tier != 1:  # the widened guard statistic really catches these classes (exact re-run of every CU)
        for k in ("dither", "low_contrast", "flat_zero_resi", "ramp"):   # (round 6: MLT_FLAT_RANGE 8 -> 6 -- amplitude-4 texture is still 69 - 73 % near-flat)
            assert report[(k, "default")][1] == n and report[(k, "no decision guard")][1] == n, (k, report[(k, "default")])
        # partial_flat sits just UNDER the flat guard's 1/8 by construction: it stays on the main arithmetic -- unless the tier was admitted behind the
        # magnitude guard (round 6: seed 21), which runs the flat guard at 1/16 and therefore takes every one of these CUs
        assert report[("partial_flat", "no decision guard")][1] == (n if m.arithmetic(size)["mag_guard_kind"] == 2 else 0)
    for _, c in ctxs:
        c.close()


@pytest.mark.parametrize("seed", [10, 23, 11, 21])
def test_tails_of_the_shipped_tier_on_a_large_sample(gpu, seed):
    """The calibration admits an arithmetic on 560 CUs with a tail factor; this looks at the tail itself: 2048 texture CUs + 512 of
    each other calibration class through the tier the calibration picked (seed 10: single pass; seed 23: single pass with another realisation of
    the weights' tap-diffused rounding -- the default one misses the contract; seed 11: hi+lo weights in some launch units; seed 21: the exact
    arithmetic in one stage + hi+lo weights) and through the exact arithmetic on the device (itself <= 1e-5 from the
    oracle, checked elsewhere): no logit beyond the contract, no decisive split flipped.  (scripts/tail_probe.py is the full-size version:
    294,912 logits per weight set, worst 8.5e-4 over nine sets: profiles/r04p_tail_probe.txt.)"""
    pkg = gpu
    S, size = pkg.synth, 128
    blob = pkg.weights.synthetic_blob(0, seed)
    m = _ctx(pkg, size, blob)
    e = _ctx(pkg, size, blob, flags=pkg.capi.FLAG_EXACT_128)
    # (seed 21, round 6: hi+lo weights in three stages behind the magnitude guard instead of an exact stage -- the plain rule had rejected it on ONE
    # outlier of 5040 logits, max 6.9e-4 = 5.9 x rms, and the outlier's CU has the set's largest logit magnitude)
    assert m.arithmetic(size)["exact"] in {10: (0,), 23: (0,), 11: (3,), 21: (3, 4)}[seed]
    assert m.arithmetic(size)["rounding"] != 0 if seed == 23 else m.arithmetic(size)["rounding"] in range(6)   # (seed 10 keeps the default, the sets in lower tiers the realisation that came closest)
    assert seed != 10 or m.arithmetic(size)["rounding"] == 0
    worst = 0.0
    for kind, n in ((None, 2048), (S.KIND_UNIFORM, 512), (S.KIND_ORG_FLAT_PRED_TEX, 512), (S.KIND_ORG_TEX_PRED_FLAT, 512), (S.KIND_PARTIAL_FLAT, 512)):
        org, pred = S.make_patches_bulk(size, n, 424242) if kind is None else S.make_patches(size, n, 424242 + kind, kind)
        poc, qp = S.make_scalars(n, 424242 + (kind or 0))
        s1, l1 = m.predict_batch(org, pred, poc, qp)
        s2, l2 = e.predict_batch(org, pred, poc, qp)
        d = np.abs(l1.astype(np.float64) - l2)
        worst = max(worst, float(d.max()))
        assert d.max() <= LOGIT_TOL - 2e-5, (kind, float(d.max()))     # (2e-5: the exact arithmetic's own distance from the oracle)
        srt = np.sort(l2[:, 5:9].astype(np.float64), axis=1)
        decisive = (srt[:, -1] - srt[:, -2]) > 2 * LOGIT_TOL
        assert not ((s1 != s2) & decisive).any()
    print(f"seed {seed}: worst |dlogit| over {(2048 + 4 * 512) * 9} logits {worst:.2e}")
    m.close(); e.close()


@pytest.mark.parametrize("size,n", [(128, 12), (64, 24), (32, 40), (16, 70)])
def test_against_oracle_seeded(gpu, size, n):
    """Ragged batch sizes (not multiples of the per-workgroup sample count) on purpose."""
    from oracle import Oracle
    pkg = gpu
    arch = pkg.synth.arch_for_size(size)
    blob = pkg.weights.synthetic_blob(arch, 21)
    org, pred = pkg.synth.make_patches(size, n, 4321)
    poc, qp = pkg.synth.make_scalars(n, 4321)
    ref, ref_split = Oracle(blob).forward(org, pred, poc, qp, threads=8)
    m = _ctx(pkg, size, blob)   # the shipped configuration: every split is compared
    split, logits = m.predict_batch(org, pred, poc, qp)
    err = float(np.abs(logits - ref).max())
    print(size, "max|dlogit| vs oracle", err)
    assert err <= LOGIT_TOL
    sl = head_slices(Oracle(blob).head_classes)[2 if size == 128 else 0]
    assert check_splits(split, ref, ref_split, sl, True, LOGIT_TOL, size) <= 1
    m.close()


def test_single_cu_strided_call_site(gpu):
    """mlt_predict with picture-buffer strides (EncCu.cpp:810-830) == batch path == oracle."""
    from oracle import Oracle
    pkg = gpu
    size = 128
    blob = pkg.weights.synthetic_blob(0, 10)
    org, pred = pkg.synth.make_patches(size, 2, 99)
    pic = np.full((2, size, 1920), -5, np.int16)
    pic[:, :, 640:640 + size] = org
    prd = np.full((2, size, size + 16), 7, np.int16)
    prd[:, :, :size] = pred
    m = _ctx(pkg, size, blob)
    bs, bl = m.predict_batch(org, pred, [8, 16], [32, 37])
    ref, _ = Oracle(blob).forward(org, pred, [8, 16], [32, 37])
    for i in range(2):
        s, l = m.predict(pic[i, :, 640:640 + size], prd[i, :, :size], [8, 16][i], [32, 37][i])
        assert s == bs[i] and np.array_equal(l, bl[i])
        assert np.abs(l - ref[i]).max() <= LOGIT_TOL
    m.close()


def test_first_max_tie_rule(gpu):
    pkg = gpu
    for size in (128, 32):
        arch = pkg.synth.arch_for_size(size)
        blob = pkg.weights.pack_blob(arch, variant_state_dict(pkg, arch, 10, "tie", size))
        org, pred = pkg.synth.make_patches(size, 5, 77)
        poc, qp = pkg.synth.make_scalars(5, 77)
        m = _ctx(pkg, size, blob)
        split, logits = m.predict_batch(org, pred, poc, qp)
        sl = head_slices([2, 3, 4] if size == 128 else [2, 3, 4, 6])[2 if size == 128 else 0]
        assert np.all(logits[:, sl][:, 0] == logits[:, sl][:, 1]), "identical rows must tie exactly"
        assert np.all(split == 0), "torch.argmax returns the first maximal index"
        m.close()


def test_batch_split_invariance_and_determinism(gpu):
    """Result of CU i must not depend on its batch neighbours or on the chunking; run-to-run identical."""
    pkg = gpu
    size = 64
    blob = pkg.weights.synthetic_blob(1, 10)
    org, pred = pkg.synth.make_patches(size, 37, 5)
    poc, qp = pkg.synth.make_scalars(37, 5)
    m = _ctx(pkg, size, blob)
    s_all, l_all = m.predict_batch(org, pred, poc, qp)
    s_again, l_again = m.predict_batch(org, pred, poc, qp)
    assert np.array_equal(l_all, l_again) and np.array_equal(s_all, s_again)
    for lo, hi in ((0, 1), (1, 9), (9, 37)):
        s, l = m.predict_batch(org[lo:hi], pred[lo:hi], poc[lo:hi], qp[lo:hi])
        assert np.array_equal(l, l_all[lo:hi]) and np.array_equal(s, s_all[lo:hi])
    m.close()


def test_guards_replace_flagged_cus_with_exact_results_on_every_entry_point(gpu):
    """Flat guard + decision guard (both on by default; switched off one by one here): CUs with >= 1/8 exactly-constant quads, and CUs whose fast decision-head
    margin is under the threshold, come back with the exact-mode logits / split (bit-identical to an exact-mode context);
    all others keep the fast result.  Same through mlt_predict, mlt_predict_batch, mlt_predict_batch_device and the
    deferred API."""
    import torch
    pkg = gpu
    size, n = 128, 40
    NC, NDG = pkg.capi.FLAG_NO_CALIBRATION, pkg.capi.FLAG_NO_DECISION_GUARD
    blob = pkg.weights.synthetic_blob(0, 21)
    org, pred = pkg.synth.make_patches(size, n, 17)
    o2, p2 = pkg.synth.make_patches(size, 2, 18, pkg.synth.KIND_SATURATED)   # 4x4 checkerboard: every aligned quad is constant
    o3, p3 = pkg.synth.make_patches(size, 2, 19, pkg.synth.KIND_FLAT)
    o4, p4 = pkg.synth.make_patches(size, 2, 20)
    o4[:, :24, :] = o3[:, :24, :]; p4[:, :24, :] = p3[:, :24, :]            # 24 of 128 rows constant: 18.75 % of the quads
    o5, p5 = pkg.synth.make_patches(size, 2, 21)
    o5[:, :8, :] = o3[:, :8, :]; p5[:, :8, :] = p3[:, :8, :]                # 6.25 %: below the threshold, stays fast
    org, pred = np.concatenate([org, o2, o3, o4, o5]), np.concatenate([pred, p2, p3, p4, p5])
    flat_flag = np.zeros(n + 8, bool)
    flat_flag[n:n + 6] = True
    n += 8
    poc, qp = pkg.synth.make_scalars(n, 17)
    raw = _ctx(pkg, size, blob, flags=NC | NDG | pkg.capi.FLAG_NO_FLAT_GUARD)
    fast = _ctx(pkg, size, blob, flags=NC | NDG)
    exact = _ctx(pkg, size, blob, flags=pkg.capi.FLAG_EXACT_128)
    s_r, l_r = raw.predict_batch(org, pred, poc, qp)
    s_f, l_f = fast.predict_batch(org, pred, poc, qp)
    s_e, l_e = exact.predict_batch(org, pred, poc, qp)
    assert fast.arithmetic(size)["exact"] == 0 and fast.arithmetic(size)["guard_reruns"] == 6
    assert np.array_equal(l_f[flat_flag], l_e[flat_flag]) and np.array_equal(s_f[flat_flag], s_e[flat_flag])
    assert np.array_equal(l_f[~flat_flag], l_r[~flat_flag]) and np.array_equal(s_f[~flat_flag], s_r[~flat_flag])
    assert not np.array_equal(l_r[flat_flag], l_e[flat_flag])
    head = slice(5, 9)  # CTU model decision head = element [2] (EncCu.cpp:913-915)
    top2 = np.sort(l_r[:, head], axis=1)
    margins = top2[:, -1] - top2[:, -2]
    thr = float(np.median(margins))
    flagged = (margins < np.float32(thr)) | flat_flag
    assert flat_flag.sum() < flagged.sum() < n
    g = _ctx(pkg, size, blob, flags=NC, guard_margin=thr)
    s_g, l_g = g.predict_batch(org, pred, poc, qp)
    assert np.array_equal(l_g[flagged], l_e[flagged]) and np.array_equal(s_g[flagged], s_e[flagged])
    assert np.array_equal(l_g[~flagged], l_r[~flagged]) and np.array_equal(s_g[~flagged], s_r[~flagged])
    s_only, none = g.predict_batch(org, pred, poc, qp, want_logits=False)
    assert none is None and np.array_equal(s_only, s_g)
    # device-pointer entry (ragged chunks so that flagged CUs fall into several chunks)
    dev = torch.device("cuda", 0)
    d = [torch.from_numpy(x).to(dev) for x in (org, pred, poc, qp)]
    d_split = torch.full((n,), -1, dtype=torch.int32, device=dev)
    d_lg = torch.zeros((n, 9), dtype=torch.float32, device=dev)
    g.predict_batch_device(n, size, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d_split.data_ptr(), d_lg.data_ptr())
    g.synchronize()
    assert np.array_equal(d_lg.cpu().numpy(), l_g) and np.array_equal(d_split.cpu().numpy(), s_g)
    d_split.fill_(-1)
    g.predict_batch_device(n, size, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d_split.data_ptr(), None)
    g.synchronize()
    assert np.array_equal(d_split.cpu().numpy(), s_g)
    # EncCu call-site entry point and the deferred API
    picks = [int(np.flatnonzero(flagged & ~flat_flag)[0]), int(np.flatnonzero(~flagged)[0]), int(np.flatnonzero(flat_flag)[0]), n - 5]
    for i in picks:
        for _ in range(2):  # second call replays the captured graph after the exact pass grew the workspace
            s1, l1 = g.predict(org[i], pred[i], int(poc[i]), int(qp[i]))
            assert s1 == s_g[i] and np.array_equal(l1, l_g[i]), i
    tickets = [g.submit(org[i], pred[i], int(poc[i]), int(qp[i])) for i in range(n)]
    for i in (n - 1, 0, 41, 7, n - 3):
        s1, l1 = g.wait(size, tickets[i])
        assert s1 == s_g[i] and np.array_equal(l1, l_g[i]), i
    for m in (raw, fast, exact, g):
        m.close()


def test_load_time_calibration_picks_the_arithmetic(gpu):
    """mlt_load_weights measures fast vs exact on 560 seeded CUs of six content classes: the bench weight set (seed 10) keeps the fast arithmetic, a
    weight set whose fp16 error is ~4x larger (seed 22: emulated rms 5.5e-4) fails it AND every hi+lo-weights tier and lands in the tier
    below exact (round 4: the exact arithmetic in some stages -- mlt_arith_info.exact == 4, .x_stages), and a tight tolerance switches any
    set to exact.  Reloading other weights into the same context re-calibrates and invalidates the captured graph."""
    import oracle
    pkg = gpu
    size = 128
    org, pred = pkg.synth.make_patches(size, 4, 5)
    poc, qp = pkg.synth.make_scalars(4, 5)
    b10, b22 = pkg.weights.synthetic_blob(0, 10), pkg.weights.synthetic_blob(0, 22)
    m = _ctx(pkg, size, b10)
    a = m.arithmetic(size)
    print("seed 10:", a)
    assert a["calibrated"] == 1 and a["exact"] == 0 and 0 < a["calib_rms"] <= 1e-3 / 5.5 and a["flat_guard"] == 1
    s10 = [m.predict(org[i], pred[i], int(poc[i]), int(qp[i])) for i in range(4)]  # captures the graph with seed-10 weights
    m.load_weights(size, b22)                                                      # frees them: the graph must not survive
    a = m.arithmetic(size)
    print("seed 22:", a)
    assert a["calibrated"] == 1 and a["exact"] == 4 and a["x_stages"] not in (0, 0xF) and (a["x_units"] & a["w2_units"]) == 0
    assert 5.5 * a["calib_rms"] <= 1e-3 and a["calib_max"] <= 0.65e-3 and a["flat_guard"] == 1
    fresh = _ctx(pkg, size, b22)
    ref, ref_split = oracle.Oracle(b22).forward(org, pred, poc, qp)
    for i in range(4):
        s1, l1 = m.predict(org[i], pred[i], int(poc[i]), int(qp[i]))
        s2, l2 = fresh.predict(org[i], pred[i], int(poc[i]), int(qp[i]))
        assert s1 == s2 and np.array_equal(l1, l2), "reloaded context differs from a fresh one"
        assert np.abs(l1 - ref[i]).max() <= LOGIT_TOL
        assert not np.array_equal(l1, s10[i][1])
    m.load_weights(size, b10)
    assert m.arithmetic(size)["exact"] == 0
    for i in range(4):
        s1, l1 = m.predict(org[i], pred[i], int(poc[i]), int(qp[i]))
        assert s1 == s10[i][0] and np.array_equal(l1, s10[i][1])
    # a tolerance no fp16 tier meets: the exact-lite arithmetic (round 5; |dlogit| ~ 1e-5 rms) is tried before the exact one ...
    tight = _ctx(pkg, size, b10, tolerance=2e-4)
    a = tight.arithmetic(size)
    print("tolerance 2e-4:", a)
    assert a["exact"] == 5 and a["calibrated"] == 1 and 5.5 * a["calib_rms"] <= 2e-4 and a["calib_max"] <= 0.65 * 2e-4 and a["flat_guard"] == 1 and a["decision_guard"] == 1   # (round 6: the tier keeps the flat guard, at 1/16)
    sl3, ll3 = tight.predict_batch(org, pred, poc, qp)
    r10, r10s = oracle.Oracle(b10).forward(org, pred, poc, qp)
    assert np.abs(ll3 - r10).max() <= 2e-4 and np.array_equal(sl3, r10s)
    for i in range(4):
        s1, l1 = tight.predict(org[i], pred[i], int(poc[i]), int(qp[i]))
        assert s1 == sl3[i] and np.array_equal(l1, ll3[i]), "exact-lite: single-CU path differs from the batch path"
    # ... and one that it does not meet either runs exact
    tighter = _ctx(pkg, size, b10, tolerance=2e-5)
    assert tighter.arithmetic(size)["exact"] == 1
    for c in (m, fresh, tight, tighter):
        c.close()


def test_middle_tier_hi_lo_weights(gpu):
    """A weight set that fails the calibration of the single-pass arithmetic is priced once more with (hi, lo) WEIGHTS on fp16
    activations (2 MFMAs per product; mlt_arith_info.exact == 3: in layer2 / layer3 only, == 2: in the whole network) before it falls
    back to the exact arithmetic: seeds 13 and 24
    land there, meet the 1e-3 contract against the oracle with the guards on, and give the same bits through every entry point;
    MLT_NO_W2 is not set in the tests.  Seeds 21 and 22 fail these tiers too -- what is left is their ACTIVATION rounding -- and land in the
    tier below exact (== 4): the exact arithmetic (model_exact's per-conv kernels, a lo plane behind the activations) in the stages of
    .x_stages, hi+lo weights or the single pass in the others.  Seed 10 never gets here (fast)."""
    import oracle
    pkg = gpu
    size = 128
    n = 40
    org, pred = pkg.synth.make_patches_bulk(size, n, 4711)
    poc, qp = pkg.synth.make_scalars(n, 4711)
    org[3] = 512; pred[3] = 512  # a constant CU: flagged by the flat-content guard, re-evaluated exactly
    for seed in (13, 24, 11, 21, 22):   # 11: admitted with hi+lo weights in two stages only (exact == 3); 21, 22: exact stages (exact == 4)
        blob = pkg.weights.synthetic_blob(0, seed)
        m = _ctx(pkg, size, blob)
        a = m.arithmetic(size)
        print(f"seed {seed}:", a)
        assert a["exact"] in ((4,) if seed == 22 else (3, 4) if seed == 21 else (2, 3)) and a["calibrated"] == 1 and 5.5 * a["calib_rms"] <= 1e-3 and a["calib_max"] <= 0.65e-3 and a["flat_guard"] == 1
        assert (a["x_stages"] != 0) == (a["exact"] == 4) and (a["x_units"] & a["w2_units"]) == 0
        # (launch-unit granularity: a stage counts as hi+lo weights when at least one of its two units is)
        assert a["w2_stages"] == sum(1 << st for st in range(4) if (a["w2_units"] >> (2 * st)) & 3)
        ref, ref_split = oracle.Oracle(blob).forward(org, pred, poc, qp, threads=8)
        s, l = m.predict_batch(org, pred, poc, qp)
        assert np.abs(l - ref).max() <= LOGIT_TOL
        assert np.abs(l[3] - ref[3]).max() <= 2e-5, "the constant CU must have been re-run exactly"
        assert m.arithmetic(size)["guard_reruns"] >= 1
        sl = head_slices(oracle.Oracle(blob).head_classes)[2]
        assert check_splits(s, ref, ref_split, sl, True, LOGIT_TOL, seed) == 0   # the shipped configuration (decision guard on): every split must be the reference's
        mg = _ctx(pkg, size, blob, flags=pkg.capi.FLAG_NO_DECISION_GUARD)     # ... and the tier itself, unguarded, meets the logit contract
        sg, lg_ = mg.predict_batch(org, pred, poc, qp)
        assert mg.arithmetic(size)["exact"] == a["exact"] and mg.arithmetic(size)["decision_guard"] == 0
        check_splits(sg, ref, ref_split, sl, False, LOGIT_TOL, seed)
        assert np.abs(lg_ - ref).max() <= LOGIT_TOL
        mg.close()
        for i in (0, 3, 7):
            s1, l1 = m.predict(org[i], pred[i], int(poc[i]), int(qp[i]))
            assert s1 == s[i] and np.array_equal(l1, l[i]), "single-CU graph path differs from the batch path"
        s2, l2 = m.predict_batch(org[:9], pred[:9], poc[:9], qp[:9])
        assert np.array_equal(l2, l[:9])
        m.load_weights(size, pkg.weights.synthetic_blob(0, 10))
        assert m.arithmetic(size)["exact"] == 0
        m.close()


def test_small_models_calibrated_prefix(gpu):
    """Round 4: the 64 / 32 / 16 models are configured exact, but the load-time calibration may keep layer0 (the largest maps, where their
    time is) -- or one of its two launch units -- on the single-pass or the hi+lo-weights kernels when the 1e-3 contract still holds
    (mlt_arith_info.exact == 4, .x_units = what stays exact); a set that does not admit it runs exact (== 1).  Either way: the oracle within LOGIT_TOL, flat CUs re-run exactly,
    the same bits through every entry point, and MLT_FLAG_NO_CALIBRATION keeps the exact arithmetic."""
    import oracle
    pkg = gpu
    n = 48
    seen = set()
    for size in (64, 32, 16):
        org, pred = pkg.synth.make_patches_bulk(size, n, 991 + size)
        poc, qp = pkg.synth.make_scalars(n, 991 + size)
        org[5] = 300; pred[5] = 300  # constant CU: the flat-content guard re-runs it exactly when layer0 is not exact
        for seed in (10, 13):
            blob = pkg.weights.synthetic_blob(1, seed)
            m = _ctx(pkg, size, blob)
            a = m.arithmetic(size)
            print(f"size {size} seed {seed}:", a)
            assert a["exact"] in (1, 4, 5)   # (5, round 5: no prefix passes but the exact-lite arithmetic does -- 64 x 64 with seed 13)
            if size in (32, 16):   # round 6: only the 64 x 64 model is calibrated down -- at 16 x 16 a prefix tier saves 9 % of the step and one guard re-run per batch costs
                                   # 79 %; at 32 x 32 the exact-lite tier gains <= 2 % and loses 29 % with 5 % flat CUs in the batch
                assert a["exact"] == 1 and a["calibrated"] == 0 and a["flat_guard"] == 0 and a["decision_guard"] == 0, a
            seen.add(a["exact"])
            if a["exact"] == 4:
                # (every stage behind layer0 exact; of layer0 at most one launch unit -- the 64 x 64 model keeps layer0.1 exact)
                assert a["x_units"] in (0x3FC, 0x3FE, 0x3FD) and a["w2_stages"] in (0, 1) and a["calibrated"] == 1 and a["flat_guard"] == 1
                assert a["x_stages"] == sum(1 << st for st in range(5) if (a["x_units"] >> (2 * st)) & 3)
                assert 5.5 * a["calib_rms"] <= 1e-3 and a["calib_max"] <= 0.65e-3
            ref, ref_split = oracle.Oracle(blob).forward(org, pred, poc, qp, threads=8)
            s, l = m.predict_batch(org, pred, poc, qp)
            assert np.abs(l - ref).max() <= LOGIT_TOL
            # (the exact-lite tier has no flat-content failure mode -- its weights keep their lo parts -- and therefore no flat guard: the constant CU
            # carries the tier's own error, a fifth of the contract at most)
            assert np.abs(l[5] - ref[5]).max() <= (2e-4 if a["exact"] == 5 else 2e-5)
            sl = head_slices(oracle.Oracle(blob).head_classes)[0]
            check_splits(s, ref, ref_split, sl, a["exact"] == 1 or a["decision_guard"] == 1, LOGIT_TOL, f"{size}/{seed}")
            for i in (0, 5, 11):
                s1, l1 = m.predict(org[i], pred[i], int(poc[i]), int(qp[i]))
                assert s1 == s[i] and np.array_equal(l1, l[i]), "single-CU path differs from the batch path"
            e = _ctx(pkg, size, blob, flags=pkg.capi.FLAG_NO_CALIBRATION)
            assert e.arithmetic(size)["exact"] == 1
            se, le = e.predict_batch(org, pred, poc, qp)
            assert np.abs(le - ref).max() <= 2e-5
            m.close(); e.close()
    assert 4 in seen, "no small model landed in the calibrated-prefix tier: the test does not cover it"


def test_repeated_runs_bit_identical_128(gpu):
    """The conv kernels synchronise by hand (LDS-DMA landing published by counted vmcnt + barrier, fragment reads behind
    counted lgkmcnt): a missing wait shows up as run-to-run differences.  Ragged batch => partial tiles / tail workgroups."""
    pkg = gpu
    size, n = 128, 301
    blob = pkg.weights.synthetic_blob(0, 12)
    org, pred = pkg.synth.make_patches_bulk(size, n, 9)
    poc, qp = pkg.synth.make_scalars(n, 9)
    m = _ctx(pkg, size, blob, flags=pkg.capi.FLAG_NO_CALIBRATION)  # the fast kernels are the ones under test
    s0, l0 = m.predict_batch(org, pred, poc, qp)
    assert np.isfinite(l0).all()
    for _ in range(6):
        s1, l1 = m.predict_batch(org, pred, poc, qp)
        assert np.array_equal(l0, l1) and np.array_equal(s0, s1)
    # and the first CUs do not depend on how many follow (different tile counts / persistent-grid shapes)
    s2, l2 = m.predict_batch(org[:40], pred[:40], poc[:40], qp[:40])
    assert np.array_equal(l2, l0[:40]) and np.array_equal(s2, s0[:40])
    m.close()


def test_pipelined_host_batch_equals_single_pass(gpu):
    """mlt_predict_batch pipelines host batches in 512-CU sub-chunks over two staging sets (H2D under compute); the result
    must be bit-identical to evaluating the same CUs in separate small calls, ragged last sub-chunk included."""
    pkg = gpu
    size, n = 64, 512 * 3 + 77
    blob = pkg.weights.synthetic_blob(1, 4)
    org, pred = pkg.synth.make_patches_bulk(size, n, 21)
    poc, qp = pkg.synth.make_scalars(n, 21)
    m = _ctx(pkg, size, blob)
    s_all, l_all = m.predict_batch(org, pred, poc, qp)
    s_only, _ = m.predict_batch(org, pred, poc, qp, want_logits=False)
    assert np.array_equal(s_only, s_all)
    for lo, hi in ((0, 300), (300, 512), (512, 1024), (1536, n)):
        s, l = m.predict_batch(org[lo:hi], pred[lo:hi], poc[lo:hi], qp[lo:hi])
        assert np.array_equal(l, l_all[lo:hi]) and np.array_equal(s, s_all[lo:hi])
    m.close()


def test_deferred_submit_flush_wait(gpu):
    """mlt_submit / mlt_flush / mlt_wait (encoder-side batching): tickets resolve to exactly what mlt_predict returns, in any
    wait order, across an automatic flush of a full batch, and expired / unknown tickets are rejected."""
    pkg = gpu
    size = 128
    cap = 64  # MLT_DEFER_CAP
    blob = pkg.weights.synthetic_blob(0, 8)
    n = cap + 9
    org, pred = pkg.synth.make_patches_bulk(size, n, 31)
    poc, qp = pkg.synth.make_scalars(n, 31)
    pic = np.zeros((n, size, 200), np.int16)  # CUs taken out of wider rows: exercises the strided gather
    pic[:, :, 40:40 + size] = org
    m = _ctx(pkg, size, blob)
    want = [m.predict(pic[i, :, 40:40 + size], pred[i], int(poc[i]), int(qp[i])) for i in range(n)]
    tickets = [m.submit(pic[i, :, 40:40 + size], pred[i], int(poc[i]), int(qp[i])) for i in range(n)]  # 64 -> auto flush -> 9 pending
    assert tickets == list(range(n))
    for i in list(range(n - 1, cap - 1, -1)) + [5, 0, 63, 17]:  # second batch first (flushes it), then the first batch
        s, l = m.wait(size, tickets[i])
        assert s == want[i][0] and np.array_equal(l, want[i][1])
    m.flush(size)  # nothing pending: no-op
    t2 = [m.submit(org[i], pred[i], int(poc[i]), int(qp[i])) for i in range(3)]
    m.flush(size)
    t3 = m.submit(org[3], pred[3], int(poc[3]), int(qp[3]))
    assert m.wait(size, t3)[0] == want[3][0]
    assert m.wait(size, t2[1])[0] == want[1][0]  # previous generation is still readable
    with pytest.raises(pkg.MltError):
        m.wait(size, tickets[0])  # two newer batches were started since
    with pytest.raises(pkg.MltError):
        m.wait(size, t3 + 5)  # never issued
    m.close()


def test_kernel_variant_switches_do_not_change_results(gpu):
    """Small launches use latency variants of the >= 64-channel convs (<= 16384 output pixels per launch: 64@32 up to 16 CUs,
    128@16 / 64->128 up to 64, 256@8 / 128->256 up to 256); above that the three stride-1 convs of the 64-channel stage run as
    one chain launch (b0 through HBM in the writing wave's order, sc chunk-major) and the 128- and 256-channel stages as ONE
    whole-stage launch each (chain_kernel S2: stride-2 conv + shortcut + three convs, intermediates in LDS / registers; 257 and
    258 also exercise its half-empty last tile of two samples).  A CU's logits must be bit-identical on either side of every switch
    point, and equal to the oracle within the tolerance."""
    import oracle
    pkg = gpu
    size = 128
    blob = pkg.weights.synthetic_blob(0, 15)
    nmax = 258
    org, pred = pkg.synth.make_patches_bulk(size, nmax, 41)
    poc, qp = pkg.synth.make_scalars(nmax, 41)
    m = _ctx(pkg, size, blob, flags=pkg.capi.FLAG_NO_CALIBRATION)  # the fast kernels are the ones under test
    s8, l8 = m.predict_batch(org[:8], pred[:8], poc[:8], qp[:8])
    ref, ref_split = oracle.Oracle(blob).forward(org[:8], pred[:8], poc[:8], qp[:8])
    assert np.abs(l8 - ref).max() <= LOGIT_TOL and np.array_equal(s8, ref_split)
    for n in (1, 16, 17, 64, 65, 256, 257, 258):
        s, l = m.predict_batch(org[:n], pred[:n], poc[:n], qp[:n])
        k = min(n, 8)
        assert np.array_equal(l[:k], l8[:k]) and np.array_equal(s[:k], s8[:k]), n
    m.close()


def test_padding_from_beyond_the_lds_equals_the_masked_kernels(gpu, monkeypatch):
    """The chain / whole-stage kernels take conv padding from DS reads beyond the LDS allocation (zeros on gfx950; probed per context at
    mlt_init) instead of zero masks.  A context made with MLT_NO_LDS_OOB=1 runs the masked form of the same kernels -- what a device
    that fails the probe would get: same bits."""
    pkg = gpu
    size, n = 128, 70  # above every switch point of the 64- and 128-channel stages, two half-filled tiles for the 256 stage
    blob = pkg.weights.synthetic_blob(0, 10)
    org, pred = pkg.synth.make_patches_bulk(size, n, 8)
    poc, qp = pkg.synth.make_scalars(n, 8)
    m = _ctx(pkg, size, blob, flags=pkg.capi.FLAG_NO_CALIBRATION)
    s0, l0 = m.predict_batch(org, pred, poc, qp)
    m.close()
    monkeypatch.setenv("MLT_TUNING", "1")  # the library's switches are honoured only with it (an encoder's environment cannot flip them by accident)
    monkeypatch.setenv("MLT_NO_LDS_OOB", "1")
    mm = _ctx(pkg, size, blob, flags=pkg.capi.FLAG_NO_CALIBRATION)
    s1, l1 = mm.predict_batch(org, pred, poc, qp)
    mm.close()
    assert np.array_equal(l0, l1) and np.array_equal(s0, s1)


def test_head_index_option_and_errors(gpu):
    pkg = gpu
    blob = pkg.weights.synthetic_blob(1, 10)
    org, pred = pkg.synth.make_patches(16, 6, 3)
    poc, qp = pkg.synth.make_scalars(6, 3)
    m0 = _ctx(pkg, 16, blob)
    m3 = _ctx(pkg, 16, blob, head_index={16: 3})
    s0, l0 = m0.predict_batch(org, pred, poc, qp)
    s3, l3 = m3.predict_batch(org, pred, poc, qp)
    assert np.array_equal(l0, l3)
    assert np.array_equal(s0, np.argmax(l0[:, 0:2], axis=1)) and np.array_equal(s3, np.argmax(l0[:, 9:15], axis=1))
    with pytest.raises(pkg.MltError) as ei:  # size not enabled -> caller keeps -1 -> full RDO
        m0.predict_batch(*pkg.synth.make_patches(32, 1, 3), [0], [30])
    assert ei.value.code == 4
    with pytest.raises(pkg.MltError):
        pkg.MltCnn(device=0, sizes=(128,), blobs={128: blob})  # CU-arch blob for the 128 slot
    with pytest.raises(pkg.MltError):
        pkg.MltCnn(device=0, sizes=(128,), blobs={128: b"garbage"})
    m0.close(); m3.close()


def test_cpp_call_site_demo_matches_python_binding(gpu, tmp_path):
    """host/mlt_split_predictor.hpp (the C++ mirror of EncCu.cpp:746-756,806-921) driven from a plain g++ program,
    weights read from <dir>/MLTORPQ_splitMode_128.mltw like the reference reads its .pt (EncCu.cpp:897-899)."""
    import os
    import subprocess
    pkg = gpu
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    blob = pkg.weights.synthetic_blob(0, 10)
    (tmp_path / "MLTORPQ_splitMode_128.mltw").write_bytes(blob)
    exe = str(tmp_path / "callsite_demo")
    lib_dir = os.path.dirname(pkg.build.LIB)
    subprocess.check_call(["g++", "-std=c++17", "-O2", os.path.join(root, "host", "callsite_demo.cpp"), "-o", exe,
                           "-L" + lib_dir, "-lmltcnn_hip", "-Wl,-rpath," + lib_dir])
    out = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    # same CU through the Python binding (the demo's LCG picture, re-created here)
    picW, picH, cux, cuy, cuw = 1920, 1080, 256, 128, 128
    lcg = 12345
    pic = np.empty(picW * picH, np.int16)
    vals = []
    for _ in range(picW * picH):
        lcg = (lcg * 1664525 + 1013904223) & 0xFFFFFFFF
        vals.append((lcg >> 22) & 1023)
    pic[:] = vals
    pic = pic.reshape(picH, picW)
    pred = np.empty((cuw, cuw), np.int16)
    for y in range(cuw):
        for x in range(cuw):
            lcg = (lcg * 1664525 + 1013904223) & 0xFFFFFFFF
            pred[y, x] = min(max(int(pic[cuy + y, cux + x]) + ((lcg >> 24) % 41) - 20, 0), 1023)
    m = _ctx(pkg, 128, blob)
    split, logits = m.predict(pic[cuy:cuy + cuw, cux:cux + cuw], pred, 8, 32)
    m.close()
    assert f"predictedSplitMode = {split} " in out.stdout, out.stdout
    got = [float(v) for v in out.stdout.split("=")[-1].split()]
    assert np.allclose(got, logits[5:9], atol=2e-4), (got, logits[5:9])


@pytest.mark.parametrize("size", SIZES)
def test_full_batch_4096_properties(gpu, size):
    """BASELINE.json's full sizes (configs[1]: 4096 x 128x128, configs[2]: 4096 x 64 / 32 / 16): size-independent properties,
    all bit-exact -- (1) run-to-run determinism, (2) permutation equivariance (a CU's result does not depend on its batch
    position or neighbours), (3) invariance to the internal chunking (ragged MLT_CHUNK), (4) device-pointer entry ==
    host-pointer entry -- and ALL 4096 CUs of that batch against the oracle (logits and splits)."""
    import os
    import torch
    from oracle import Oracle
    pkg = gpu
    n = 4096
    arch = pkg.synth.arch_for_size(size)
    blob = pkg.weights.synthetic_blob(arch, 10)
    org, pred = pkg.synth.make_patches_bulk(size, n, 0xBEEF)
    poc, qp = pkg.synth.make_scalars(n, 0xBEEF)
    nl = sum(pkg.synth.HEAD_CLASSES[arch])
    m = _ctx(pkg, size, blob)
    s0, l0 = m.predict_batch(org, pred, poc, qp)
    s1, l1 = m.predict_batch(org, pred, poc, qp)
    m_arith = m.arithmetic(size)
    assert np.array_equal(l0, l1) and np.array_equal(s0, s1), "not deterministic"
    perm = np.random.RandomState(7).permutation(n)
    sp, lp = m.predict_batch(org[perm], pred[perm], poc[perm], qp[perm])
    assert np.array_equal(lp, l0[perm]) and np.array_equal(sp, s0[perm]), "result depends on batch position"
    # device-pointer entry (what bench.py times)
    dev = torch.device("cuda", 0)
    d = [torch.from_numpy(x).to(dev) for x in (org, pred, poc, qp)]
    d_split = torch.full((n,), -1, dtype=torch.int32, device=dev)
    d_logits = torch.zeros((n, nl), dtype=torch.float32, device=dev)
    m.predict_batch_device(n, size, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d_split.data_ptr(), d_logits.data_ptr())
    m.synchronize()
    assert np.array_equal(d_logits.cpu().numpy(), l0) and np.array_equal(d_split.cpu().numpy(), s0)
    m.close()
    os.environ["MLT_TUNING"] = "1"
    os.environ["MLT_CHUNK"] = "1000"  # ragged chunks: 1000,1000,1000,1000,96
    try:
        mc = _ctx(pkg, size, blob)
        sc, lc = mc.predict_batch(org, pred, poc, qp)
        d_split.fill_(-1)
        mc.predict_batch_device(n, size, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d_split.data_ptr(), d_logits.data_ptr())
        mc.synchronize()
        mc.close()
    finally:
        del os.environ["MLT_CHUNK"]
        del os.environ["MLT_TUNING"]
    assert np.array_equal(lc, l0) and np.array_equal(sc, s0), "result depends on chunking"
    assert np.array_equal(d_logits.cpu().numpy(), l0) and np.array_equal(d_split.cpu().numpy(), s0), "device entry depends on chunking"
    # EVERY CU of the full batch against the oracle (round 5; the C oracle does 4096 x 128x128 in ~20 s on the box's host cores, the small
    # sizes are 5-65x cheaper): logits within the tolerance, and -- flags = 0 is the shipped configuration, decision guard on -- every split
    # equal to the reference's wherever the reference's own fp32 margin exceeds 4e-5 (the CUs below that are counted)
    ref, ref_split = Oracle(blob).forward(org, pred, poc, qp, threads=os.cpu_count() or 8)
    err = float(np.abs(l0 - ref).max())
    assert err <= LOGIT_TOL, err
    sl = head_slices(pkg.synth.HEAD_CLASSES[arch])[2 if size == 128 else 0]
    a = m_arith
    assert a["exact"] == 1 or a["decision_guard"] == 1
    undecided = check_splits(s0, ref, ref_split, sl, True, LOGIT_TOL, size)
    print(f"{size}: full batch of {n} CUs vs the oracle: max|dlogit| {err:.2e}, {undecided} CUs the oracle itself ties to within 4e-5, tier {a['exact']}, "
          f"{a['guard_reruns']} guard re-runs over all passes")
    assert undecided <= 8


def test_two_host_threads_two_contexts(gpu):
    """SURVEY 8b threading contract: one context per EncCu thread, mlt_init thread-safe.  Two host threads create their own
    contexts concurrently (different CU sizes and weight sets) and interleave mlt_predict / mlt_predict_batch calls; every
    result equals the single-threaded one."""
    import threading
    pkg = gpu
    jobs = {}
    for t, (size, seed) in enumerate(((128, 10), (32, 7))):
        blob = pkg.weights.synthetic_blob(pkg.synth.arch_for_size(size), seed)
        org, pred = pkg.synth.make_patches_bulk(size, 24, 60 + t)
        poc, qp = pkg.synth.make_scalars(24, 60 + t)
        m = _ctx(pkg, size, blob)
        jobs[t] = dict(size=size, blob=blob, data=(org, pred, poc, qp), want=m.predict_batch(org, pred, poc, qp))
        m.close()
    errors = []
    start = threading.Barrier(2)

    def worker(t):
        try:
            j = jobs[t]
            org, pred, poc, qp = j["data"]
            start.wait()
            m = _ctx(pkg, j["size"], j["blob"])  # concurrent mlt_init + weight load + calibration
            for rnd in range(6):
                for i in range(0, 24, 5):
                    s, l = m.predict(org[i], pred[i], int(poc[i]), int(qp[i]))
                    assert s == j["want"][0][i] and np.array_equal(l, j["want"][1][i]), (t, rnd, i)
                s, l = m.predict_batch(org, pred, poc, qp)
                assert np.array_equal(s, j["want"][0]) and np.array_equal(l, j["want"][1]), (t, rnd)
            m.close()
        except Exception as e:  # noqa: BLE001
            errors.append((t, repr(e)))

    th = [threading.Thread(target=worker, args=(t,)) for t in jobs]
    for x in th:
        x.start()
    for x in th:
        x.join(timeout=300)
    assert not errors, errors


def test_empty_and_single_batches(gpu):
    pkg = gpu
    blob = pkg.weights.synthetic_blob(1, 10)
    m = _ctx(pkg, 32, blob)
    org, pred = pkg.synth.make_patches(32, 3, 9)
    poc, qp = pkg.synth.make_scalars(3, 9)
    s_e, l_e = m.predict_batch(org[:0], pred[:0], poc[:0], qp[:0])
    assert s_e.shape == (0,) and l_e.shape == (0, 15)
    s3, l3 = m.predict_batch(org, pred, poc, qp)
    for i in range(3):
        s1, l1 = m.predict(org[i], pred[i], int(poc[i]), int(qp[i]))
        assert s1 == s3[i] and np.array_equal(l1, l3[i])
    m.close()


@pytest.mark.parametrize("size", (128, 32))
def test_out_of_range_pels_follow_the_reference_casts(gpu, size):
    """Pel is int16; the reference casts to uint16, takes |o - p| on uint16 and clips the scaled value to [0, 1]
    (EncCu.cpp:816,827,833,848-867).  Negative and > 10-bit samples must go through the same casts on the GPU (fused first
    layer for 128, stem5 kernel for 32) as in the oracle."""
    import oracle
    pkg = gpu
    n = 6
    blob = pkg.weights.synthetic_blob(pkg.synth.arch_for_size(size), 3)
    org, pred = pkg.synth.make_patches_bulk(size, n, 77)
    rng = np.random.default_rng(5)
    weird = np.array([-1, -5, -32768, 32767, 1024, 2000, 1023, 0], np.int16)
    for a in (org, pred):
        idx = rng.integers(0, a.size, size=a.size // 50)
        a.reshape(-1)[idx] = rng.choice(weird, size=idx.size)
    poc, qp = pkg.synth.make_scalars(n, 77)
    ref, ref_split = oracle.Oracle(blob).forward(org, pred, poc, qp)
    m = _ctx(pkg, size, blob)
    s, l = m.predict_batch(org, pred, poc, qp)
    assert np.abs(l - ref).max() <= LOGIT_TOL
    hs = head_slices(pkg.synth.HEAD_CLASSES[pkg.synth.arch_for_size(size)])
    dec = hs[2] if size == 128 else hs[0]
    check_splits(s, ref, ref_split, dec, m.arithmetic(size)["exact"] == 1, 2 * LOGIT_TOL, size)
    m.close()


def test_device_entry_point_chunk_loop(gpu, monkeypatch):
    """mlt_predict_batch_device walks batches larger than MLT_CHUNK in chunks (workspace sized for one chunk); with a tiny
    chunk the result must still equal the single-pass result, ragged last chunk included."""
    import torch
    pkg = gpu
    size, n = 64, 250
    blob = pkg.weights.synthetic_blob(1, 6)
    org, pred = pkg.synth.make_patches_bulk(size, n, 13)
    poc, qp = pkg.synth.make_scalars(n, 13)
    ref = _ctx(pkg, size, blob)
    s_ref, l_ref = ref.predict_batch(org, pred, poc, qp)
    ref.close()
    monkeypatch.setenv("MLT_TUNING", "1")
    monkeypatch.setenv("MLT_CHUNK", "96")
    m = _ctx(pkg, size, blob, max_batch=n)
    dev = torch.device("cuda", 0)
    d = [torch.from_numpy(x).to(dev) for x in (org, pred, poc, qp)]
    d_split = torch.full((n,), -7, dtype=torch.int32, device=dev)
    d_lg = torch.zeros((n, m.num_logits(size)), dtype=torch.float32, device=dev)
    m.predict_batch_device(n, size, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d_split.data_ptr(), d_lg.data_ptr())
    m.synchronize()
    assert np.array_equal(d_split.cpu().numpy(), s_ref) and np.array_equal(d_lg.cpu().numpy(), l_ref)
    m.close()


def test_one_context_serving_all_four_cu_sizes(gpu):
    """one context with the 128 model and the three CU-model weight sets loaded, calls of different sizes
    interleaved (single-CU, batch and deferred entry points share workspaces, graphs and staging) == single-size contexts."""
    pkg = gpu
    blobs = {s: pkg.weights.synthetic_blob(pkg.synth.arch_for_size(s), 20 + s) for s in SIZES}
    allm = pkg.MltCnn(device=0, sizes=SIZES, blobs=blobs)
    data, want = {}, {}
    for s in SIZES:
        org, pred = pkg.synth.make_patches_bulk(s, 6, 50 + s)
        poc, qp = pkg.synth.make_scalars(6, 50 + s)
        data[s] = (org, pred, poc, qp)
        one = _ctx(pkg, s, blobs[s])
        want[s] = one.predict_batch(org, pred, poc, qp)
        one.close()
    tickets = {}
    for rnd in range(2):
        for s in (16, 128, 32, 64, 128, 16):
            org, pred, poc, qp = data[s]
            i = (rnd * 3 + s) % 6
            sp, lg = allm.predict(org[i], pred[i], int(poc[i]), int(qp[i]))
            assert sp == want[s][0][i] and np.array_equal(lg, want[s][1][i]), (s, i)
            tickets.setdefault(s, []).append((i, allm.submit(org[i], pred[i], int(poc[i]), int(qp[i]))))
        for s in SIZES:
            org, pred, poc, qp = data[s]
            sb, lb = allm.predict_batch(org, pred, poc, qp)
            assert np.array_equal(sb, want[s][0]) and np.array_equal(lb, want[s][1]), s
    for s, lst in tickets.items():
        for i, t in lst:
            sp, lg = allm.wait(s, t)
            assert sp == want[s][0][i] and np.array_equal(lg, want[s][1][i]), (s, i)
    allm.close()


def test_bench_contract_two_ranks_on_one_gpu(gpu):
    """Plain `python bench.py --gpus 2` (no torchrun in the command: bench.py starts its own ranks as a child process) with
    the two ranks sharing the one GPU of this box (gloo for the init-time collectives): rank 0 prints exactly one JSON line
    carrying the contract keys, n_gpus = 2, and parity-clean results over the checked CUs."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MLT_BENCH_OVERSUBSCRIBE="1")
    env.pop("WORLD_SIZE", None)
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "256", "--sustain-s", "0"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=root)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stderr[-2000:]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "parity", "rccl"):
        assert k in d
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["vs_baseline"] is None and d["value"] > 0
    assert d["rccl"]["world"] == 2 and len(d["rccl"]["devices"]) == 2 and d["rccl"]["backend"] == "gloo"  # nccl (RCCL) when ranks <= GPUs
    assert d["roofline"]["bound"] in ("hbm", "mfma") and 0 < d["roofline"]["frac"] < 1
    assert d["parity"]["max_abs_dlogit"] <= LOGIT_TOL and d["parity"]["split_mismatch_decisive"] == 0
    # without the oversubscription switch two ranks on a one-GPU box must fail loudly instead of measuring one GPU twice
    env.pop("MLT_BENCH_OVERSUBSCRIBE")
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "64", "--sustain-s", "0"],
                         env=env, capture_output=True, text=True, timeout=900, cwd=root)
    assert bad.returncode != 0 and not [l for l in bad.stdout.splitlines() if l.startswith("{")]
    assert d["per_rank"] and len(d["per_rank"]["cu_per_s"]) == 2 and d["per_rank"]["min"] <= d["per_rank"]["max"]   # a straggler rank would show


def test_rccl_single_rank_weight_broadcast_and_bench_line(gpu, tmp_path):
    """What a 1-GPU box can prove of the multi-GPU path (north_star "RCCL broadcast of weights over xGMI"): RCCL itself
    (torch.distributed backend "nccl") initialises, shard.broadcast_blob moves the weight blob through DEVICE tensors and returns
    it unchanged, and bench.py launched the way the driver launches it (torch.distributed.run, one rank) reports rccl.backend ==
    "nccl" with parity-clean results.  Both in child processes: a process group is process-global state."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29000 + (os.getpid() % 2000)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MLT_BENCH_OVERSUBSCRIBE"):
        env.pop(k, None)
    probe = tmp_path / "rccl_probe.py"
    probe.write_text(f"""
import hashlib, os, sys
sys.path.insert(0, {root!r})
import torch, torch.distributed as dist
import mltcnn_pkg
pkg = mltcnn_pkg.load()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="{port}")
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", rank=0, world_size=1)
assert dist.get_backend() == "nccl"
blob = pkg.weights.synthetic_blob(0, 10)
got = pkg.shard.broadcast_blob(blob, dist, torch.device("cuda", 0))
assert got == blob, "blob changed on its way through RCCL"
t = torch.arange(8, dtype=torch.int32, device="cuda")
dist.all_reduce(t)
assert t.cpu().tolist() == list(range(8))
dist.destroy_process_group()
print("RCCL_OK", hashlib.sha256(got).hexdigest()[:16], len(got))
""")
    out = subprocess.run([sys.executable, str(probe)], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port + 1),
           os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "256", "--no-cpu-baseline", "--sustain-s", "0"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=root)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert out.returncode == 0 and len(lines) == 1, out.stdout[-1500:] + out.stderr[-1500:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["rccl"]["backend"] == "nccl" and d["rccl"]["world"] == 1 and d["rccl"]["weight_blob_bytes"] > 1e6
    assert d["parity"]["max_abs_dlogit"] <= LOGIT_TOL and d["parity"]["split_mismatch_decisive"] == 0
    assert d["per_rank"]["cu_per_s"] and abs(d["per_rank"]["max"] - d["value"]) / d["value"] < 0.05


def test_contexts_on_every_device_of_one_process(gpu):
    """ABI 3 multi-device entry (SURVEY 8b "device list", 8e): ONE context over every ordinal hipGetDeviceCount() reports (+ ordinal 0 a
    second time, so a 1-GPU box runs the sharded path with two internal contexts).  mlt_predict_batch shards contiguously over the
    devices (one host thread each), mlt_submit deals CUs round-robin, weights are uploaded / calibrated once per device from the one
    blob: every result is bit-identical to a plain one-device context, ragged shard sizes included."""
    import torch
    pkg = gpu
    size, n = 128, 29
    blob = pkg.weights.synthetic_blob(0, 10)
    org, pred = pkg.synth.make_patches(size, n, 31)
    org[5] = 300; pred[5] = 310                        # a constant CU: flagged on whichever device it lands, re-run exactly there
    poc, qp = pkg.synth.make_scalars(n, 31)
    ndev = torch.cuda.device_count()
    devices = list(range(ndev)) + [0]
    one = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob})
    multi = pkg.MltCnn(sizes=(size,), blobs={size: blob}, devices=devices)
    assert multi.num_devices() == len(devices) >= 2 and one.num_devices() == 1
    tiers = [multi.arithmetic_of_device(i, size) for i in range(len(devices))]
    assert all(t["exact"] == tiers[0]["exact"] and t["w2_stages"] == tiers[0]["w2_stages"] and t["calibrated"] == 1 for t in tiers)
    s0, l0 = one.predict_batch(org, pred, poc, qp)
    for m in (n, 7, 2, 1):                             # shards of 10/9/10, 3/2/2, 1/0/1 ... CUs
        s, l = multi.predict_batch(org[:m], pred[:m], poc[:m], qp[:m])
        assert np.array_equal(s, s0[:m]) and np.array_equal(l, l0[:m]), m
    assert sum(multi.arithmetic_of_device(i, size)["guard_reruns"] for i in range(len(devices))) >= 1
    s1, l1 = multi.predict(org[3], pred[3], int(poc[3]), int(qp[3]))
    assert s1 == s0[3] and np.array_equal(l1, l0[3])
    tickets = [multi.submit(org[i], pred[i], int(poc[i]), int(qp[i])) for i in range(11)]   # round-robin: top byte = device index
    assert sorted({t >> 56 for t in tickets}) == list(range(len(devices)))
    multi.flush(size)
    for i in (10, 0, 5, 3, 7):
        s, l = multi.wait(size, tickets[i])
        assert s == s0[i] and np.array_equal(l, l0[i]), i
    with pytest.raises(pkg.MltError):
        multi.wait(size, (len(devices) + 1) << 56)     # a device index that does not exist
    multi.synchronize()
    multi.load_weights(size, pkg.weights.synthetic_blob(0, 8))                                # reload reaches every device
    fresh = pkg.MltCnn(device=0, sizes=(size,), blobs={size: pkg.weights.synthetic_blob(0, 8)})
    s2, l2 = multi.predict_batch(org[:9], pred[:9], poc[:9], qp[:9])
    s3, l3 = fresh.predict_batch(org[:9], pred[:9], poc[:9], qp[:9])
    assert np.array_equal(s2, s3) and np.array_equal(l2, l3)
    try:
        pkg.MltCnn(sizes=(size,), blobs={size: blob}, devices=[0, ndev])   # one past the last ordinal: fails loudly, no fallback to device 0
        raise AssertionError("mlt_init accepted a device ordinal that does not exist")
    except pkg.capi.MltError as e:
        assert e.code == 2, e   # MLT_ERR_NO_DEVICE (include/mltcnn.h)
    for c in (one, multi, fresh):
        c.close()


def test_calibrate_on_caller_content(gpu):
    """ABI 4 mlt_calibrate: the load-time decision repeated with the INTEGRATOR's CUs -- natural-statistics scenes here, the
    class the library's synthetic set does not hold -- appended to the 560 synthetic CUs (own content class) or replacing them.  CUs the
    flat-content guard re-evaluates exactly anyway do not count; the context keeps working (captured graphs dropped, same bits through every
    entry point, the oracle within the tolerance); argument errors are errors."""
    from oracle import Oracle
    pkg = gpu
    size, n = 128, 320
    blob = pkg.weights.synthetic_blob(0, 10)
    org, pred, _ = pkg.synth.make_mix_bulk(size, n, 0xCA11B, 0.0, True)       # 1/f-spectrum scenes + motion-shifted prediction
    org[:8] = 400; pred[:8] = 404                                             # constant CUs: flagged by the flat guard, must not count
    poc, qp = pkg.synth.make_scalars(n, 0xCA11B)
    m = _ctx(pkg, size, blob)
    a0 = m.arithmetic(size)
    assert a0["calibrated"] == 1 and a0["calib_cus"] == 560 and a0["calib_caller_cus"] == 0
    s_before = [m.predict(org[i], pred[i], int(poc[i]), int(qp[i])) for i in (20, 21)]   # captures the single-CU graph
    m.calibrate(size, org, pred, poc, qp)
    a1 = m.arithmetic(size)
    print("append:", a1)
    assert a1["calibrated"] == 1 and 0 < a1["calib_caller_cus"] <= n - 8 and a1["calib_cus"] == 560 + a1["calib_caller_cus"]
    assert a1["exact"] in (0, 1, 2, 3, 4) and (a1["exact"] == 1 or (5.5 * a1["calib_rms"] <= 1e-3 and a1["calib_max"] <= 0.65e-3))
    ref, ref_split = Oracle(blob).forward(org[:32], pred[:32], poc[:32], qp[:32], threads=8)
    s, l = m.predict_batch(org[:32], pred[:32], poc[:32], qp[:32])
    assert np.abs(l - ref).max() <= LOGIT_TOL
    check_splits(s, ref, ref_split, head_slices([2, 3, 4])[2], True, LOGIT_TOL, "calibrated")
    for j, i in enumerate((20, 21)):
        s1, l1 = m.predict(org[i], pred[i], int(poc[i]), int(qp[i]))
        assert s1 == s[i] and np.array_equal(l1, l[i])
        if (a1["exact"], a1["w2_units"], a1["x_units"], a1["rounding"]) == (a0["exact"], a0["w2_units"], a0["x_units"], a0["rounding"]):
            assert s1 == s_before[j][0] and np.array_equal(l1, s_before[j][1])       # same tier -> same bits as before the re-calibration
    m.calibrate(size, org, pred, poc, qp, replace=True)
    a2 = m.arithmetic(size)
    print("replace:", a2)
    assert a2["calib_cus"] == a2["calib_caller_cus"] == a1["calib_caller_cus"] >= 256
    # REPLACE needs >= 256 CUs that count: a caller whose content is ALL caught by the flat guard, or a handful of CUs, cannot carry
    # the admission rule -- the synthetic set stays and the call is an APPEND; no tier is ever admitted on an empty set
    m.calibrate(size, org[:8], pred[:8], poc[:8], qp[:8], replace=True)
    a3 = m.arithmetic(size)
    print("replace, all flat:", a3)
    assert a3["calib_cus"] == 560 and a3["calib_caller_cus"] == 0 and a3["calib_max"] > 0.0
    assert (a3["exact"], a3["w2_units"], a3["x_units"], a3["rounding"]) == (a0["exact"], a0["w2_units"], a0["x_units"], a0["rounding"])
    m.calibrate(size, org[8:108], pred[8:108], poc[8:108], qp[8:108], replace=True)
    a4 = m.arithmetic(size)
    assert a4["calib_caller_cus"] > 0 and a4["calib_cus"] == 560 + a4["calib_caller_cus"]
    # a tolerance no tier below it meets: the caller's CUs take the size to the exact arithmetic like the synthetic ones do
    t = _ctx(pkg, size, blob, tolerance=2e-5)
    t.calibrate(size, org, pred, poc, qp, replace=True)
    assert t.arithmetic(size)["exact"] == 1
    with pytest.raises(pkg.MltError) as ei:
        m.calibrate(size, org[:0], pred[:0], poc[:0], qp[:0])
    assert ei.value.code == 1
    e = _ctx(pkg, size, blob, flags=pkg.capi.FLAG_EXACT_128)
    e.calibrate(size, org, pred, poc, qp)                                     # configured exact: nothing to decide, not an error
    assert e.arithmetic(size)["exact"] == 1 and e.arithmetic(size)["calibrated"] == 0
    # the small models go through the same entry (their search: prefixes of single-pass stages)
    b64 = pkg.weights.synthetic_blob(1, 10)
    o64, p64 = pkg.synth.make_patches_bulk(64, 64, 77)
    c64 = _ctx(pkg, 64, b64)
    c64.calibrate(64, o64, p64, poc[:64], qp[:64])
    a64 = c64.arithmetic(64)
    assert a64["calib_caller_cus"] > 0 and a64["exact"] in (1, 4)
    r64, _ = Oracle(b64).forward(o64[:16], p64[:16], poc[:16], qp[:16], threads=8)
    assert np.abs(c64.predict_batch(o64[:16], p64[:16], poc[:16], qp[:16])[1] - r64).max() <= LOGIT_TOL
    for c in (m, t, e, c64):
        c.close()


def test_exact_lite_arithmetic_against_the_oracle(gpu):
    """Round 5: the exact-lite arithmetic forced for every conv (MLT_FLAG_EXACT_128 | MLT_FLAG_EXACT_LITE: conv_mfma_kernel
    NSPLIT = 6 -- Wh*Xh in fp16, Wl*Xh + Wh*Xl as one v_mfma_scale_f32_32x32x64_f8f6f4 per tap and 32 channels with per-K-block E8M0 scales)
    on the golden fixtures' weight sets: |dlogit| <= 2e-4 on every fixture (measured <= 1.3e-4, rms ~1e-5: 1/20 of the single pass), the
    same bits through the large tiles and the small-launch variants, and strictly between the exact and the single-pass arithmetic in error."""
    from oracle import Oracle
    pkg = gpu
    size = 128
    golden = load_golden(size)
    worst = 0.0
    for case in golden["cases"]:
        blob, org, pred, poc, qp, exp, exp_arg = materialise(pkg, golden, case)
        m = _ctx(pkg, size, blob, flags=pkg.capi.FLAG_EXACT_128 | pkg.capi.FLAG_EXACT_LITE)
        s, l = m.predict_batch(org, pred, poc, qp)
        err = float(np.abs(l - exp).max())
        worst = max(worst, err)
        assert err <= 2e-4, (case["name"], err)
        s1, l1 = m.predict(org[0], pred[0], int(poc[0]), int(qp[0]))
        assert s1 == s[0] and np.array_equal(l1, l[0]), case["name"]
        m.close()
    print(f"exact-lite on {len(golden['cases'])} fixtures: worst |dlogit| {worst:.2e}")
    # bit-identity across launch sizes (latency variants <= 16 k output pixels vs the large tiles) and against the oracle on a larger batch
    blob = pkg.weights.synthetic_blob(0, 22)
    n = 300
    org, pred = pkg.synth.make_patches_bulk(size, n, 515)
    poc, qp = pkg.synth.make_scalars(n, 515)
    m = _ctx(pkg, size, blob, flags=pkg.capi.FLAG_EXACT_128 | pkg.capi.FLAG_EXACT_LITE)
    s, l = m.predict_batch(org, pred, poc, qp)
    for k in (1, 17, 65, 257):
        sk, lk = m.predict_batch(org[:k], pred[:k], poc[:k], qp[:k])
        assert np.array_equal(lk, l[:k]) and np.array_equal(sk, s[:k]), k
    ref, _ = Oracle(blob).forward(org[:64], pred[:64], poc[:64], qp[:64], threads=8)
    e_lite = float(np.abs(l[:64] - ref).max())
    ex = _ctx(pkg, size, blob, flags=pkg.capi.FLAG_EXACT_128)
    e_exact = float(np.abs(ex.predict_batch(org[:64], pred[:64], poc[:64], qp[:64])[1] - ref).max())
    fa = _ctx(pkg, size, blob, flags=pkg.capi.FLAG_NO_CALIBRATION | pkg.capi.FLAG_NO_FLAT_GUARD | pkg.capi.FLAG_NO_DECISION_GUARD)
    e_fast = float(np.abs(fa.predict_batch(org[:64], pred[:64], poc[:64], qp[:64])[1] - ref).max())
    print(f"seed 22, 64 CUs vs the oracle: exact {e_exact:.2e}  exact-lite {e_lite:.2e}  single pass {e_fast:.2e}")
    assert e_exact <= 2e-5 and e_lite <= 2e-4 and e_lite < 0.25 * e_fast
    for c in (m, ex, fa):
        c.close()


def test_layer0_stream_is_bit_identical(gpu):
    """(Also covers layer1_stream_kernel, the 64-channel stage's streaming launch, which switches on at the same batch size.)
    Round 5: large batches (>= 128 CUs since round 6) of 128 x 128 run ALL of layer0 in one streaming launch (layer0_stream_kernel: four row stages handing rows to
    each other through LDS rings, b0 never in HBM); smaller batches keep the two tiled launches (stem_block_kernel -> block32_kernel).  Same
    arithmetic in the same order: the logits must agree bit for bit -- 301 CUs (workgroups with one CU and with two; pipeline fill and drain across
    the CU boundary) against the same CUs in sub-batches of 100, flat CUs included (the guard statistic is gathered by the streaming kernel too)."""
    pkg = gpu
    n = 301
    blob = pkg.weights.synthetic_blob(0, 10)
    org, pred = pkg.synth.make_patches_bulk(128, n, 5150)
    poc, qp = pkg.synth.make_scalars(n, 5150)
    org[7] = 512; pred[7] = 512                       # constant CU (flat-content guard)
    org[300] = 0; pred[300] = 1023                    # extreme residual in the last CU (a workgroup's only CU)
    m = _ctx(pkg, 128, blob)
    assert m.arithmetic(128)["exact"] == 0
    s, l = m.predict_batch(org, pred, poc, qp)
    for a in range(0, n, 100):
        b = min(n, a + 100)
        s1, l1 = m.predict_batch(org[a:b], pred[a:b], poc[a:b], qp[a:b])
        assert np.array_equal(l1, l[a:b]) and np.array_equal(s1, s[a:b]), f"streaming layer0 differs from the tiled form in CUs {a}..{b}"
    # round 6: the streaming launches start at 128 CUs (the measured crossover; 256 in round 5): batches on either side of the threshold and
    # inside the 128 .. 255 range, where half of the chip's workgroup slots stay empty
    for k in (127, 128, 129, 200, 255, 256):
        s1, l1 = m.predict_batch(org[:k], pred[:k], poc[:k], qp[:k])
        assert np.array_equal(l1, l[:k]) and np.array_equal(s1, s[:k]), f"batch of {k} CUs differs"
    assert m.arithmetic(128)["guard_reruns"] >= 1
    m.close()


@pytest.mark.parametrize("seed", [23, 13, 24, 11])
def test_layer0_stream_under_the_other_tiers(gpu, seed):
    """The streaming launch also serves the tiers that keep layer0 on the single pass: seed 23 (single pass, another rounding realisation: five
    stages), seed 24 (hi+lo weights from layer2 on: five stages), seeds 13 / 11 (hi+lo weights in layer1's stride-2 conv / in all of layer1: the
    fifth stage must stay out, layer0's output goes to HBM for the two-plane kernels).  320 CUs at once against the same CUs in sub-batches of
    80 (tiled launches): bit for bit; and the batch against the oracle."""
    import oracle
    pkg = gpu
    n = 320
    blob = pkg.weights.synthetic_blob(0, seed)
    org, pred = pkg.synth.make_patches_bulk(128, n, 6060 + seed)
    poc, qp = pkg.synth.make_scalars(n, 6060 + seed)
    m = _ctx(pkg, 128, blob)
    a = m.arithmetic(128)
    assert a["exact"] in (0, 3) and not (a["w2_units"] & 3), f"seed {seed}: layer0 is expected on the single pass, got {a}"
    s, l = m.predict_batch(org, pred, poc, qp)
    for lo in range(0, n, 80):
        s1, l1 = m.predict_batch(org[lo:lo + 80], pred[lo:lo + 80], poc[lo:lo + 80], qp[lo:lo + 80])
        assert np.array_equal(l1, l[lo:lo + 80]) and np.array_equal(s1, s[lo:lo + 80]), f"seed {seed}: streaming layer0 differs from the tiled form in CUs {lo}.."
    ref, ref_split = oracle.Oracle(blob).forward(org[:64], pred[:64], poc[:64], qp[:64], threads=8)
    assert np.abs(l[:64] - ref).max() <= LOGIT_TOL
    m.close()


def test_guard_selection_in_heads_kernel_matches_select_kernel(gpu, monkeypatch):
    """Round 6: a batch's guard selection (flat content, decision margin, logit magnitude) is a tail of the heads kernel -- an unordered list through an
    atomic append, the count published by the last workgroup, both words re-armed for the next launch -- instead of a launch of its own
    (MLT_GUARD_SELECT_KERNEL=1 under MLT_TUNING=1: round 5's guard_select_kernel, ascending list).  Same CUs selected, same bits, on repeated calls of
    different sizes (the counters must come back to zero every time), through the batch, device-pointer and deferred entry points."""
    import torch
    pkg = gpu
    size = 128
    blob = pkg.weights.synthetic_blob(0, 10)
    n = 300
    ot, pt = pkg.synth.make_patches_bulk(size, n - 90, 9201)
    of, pf = pkg.synth.make_patches(size, 40, 9202, pkg.synth.KIND_FLAT)
    on, pn = pkg.synth.natural_patches(size, 50, 9203)
    org = np.concatenate([ot, of, on]); pred = np.concatenate([pt, pf, pn])
    perm = np.random.default_rng(9204).permutation(n)
    org, pred = org[perm], pred[perm]
    poc, qp = pkg.synth.make_scalars(n, 9205)
    fused = _ctx(pkg, size, blob, max_batch=n)
    monkeypatch.setenv("MLT_TUNING", "1")
    monkeypatch.setenv("MLT_GUARD_SELECT_KERNEL", "1")
    plain = _ctx(pkg, size, blob, max_batch=n)
    monkeypatch.delenv("MLT_GUARD_SELECT_KERNEL")
    for k in (n, 129, 300, 17, 2, 256):   # (repeats and shrinking / growing batches: a stale ticket or count would show as a wrong number of re-runs)
        rf, rp = fused.arithmetic(size)["guard_reruns"], plain.arithmetic(size)["guard_reruns"]
        sf, lf = fused.predict_batch(org[:k], pred[:k], poc[:k], qp[:k])
        sp, lp = plain.predict_batch(org[:k], pred[:k], poc[:k], qp[:k])
        rf, rp = fused.arithmetic(size)["guard_reruns"] - rf, plain.arithmetic(size)["guard_reruns"] - rp
        assert rf == rp and (k < 100 or rf > 0), (k, rf, rp)
        assert np.array_equal(sf, sp) and np.array_equal(lf, lp), k
    dev = torch.device("cuda", 0)
    d = [torch.from_numpy(x).to(dev) for x in (org, pred, poc, qp)]
    d_split = torch.full((n,), -7, dtype=torch.int32, device=dev)
    d_lg = torch.zeros((n, 9), dtype=torch.float32, device=dev)
    s_all, l_all = plain.predict_batch(org, pred, poc, qp)
    for _ in range(3):
        fused.predict_batch_device(n, size, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d_split.data_ptr(), d_lg.data_ptr())
        fused.synchronize()
        assert np.array_equal(d_split.cpu().numpy(), s_all) and np.array_equal(d_lg.cpu().numpy(), l_all)
    for lo in (0, 20, 150):   # deferred generations of 24 CUs
        tk = [fused.submit(org[i], pred[i], int(poc[i]), int(qp[i])) for i in range(lo, lo + 24)]
        fused.flush(size)
        for j, t in enumerate(tk):
            s1, l1 = fused.wait(size, t)
            assert s1 == s_all[lo + j] and np.array_equal(l1, l_all[lo + j]), ("deferred", lo + j)
    fused.close(); plain.close()


def test_small_guard_reruns_carry_the_same_bits_at_any_count(gpu):
    """Round 6.  The exact re-run of a FEW flagged CUs (what the encoder's calls and the small models' batches see: k = 1 .. 8) against the same CUs
    inside a batch with many flagged ones and against the exact arithmetic itself -- repeated calls, several k, a workspace re-allocation in between, every entry
    point, a caller-owned stream.  (Written for a variant that replayed the re-run's ~20 launches from a hipGraph per (size, k); that variant measured EQUAL --
    profiles/r06h_ab_rerun_graph.txt: the re-run of one 16 x 16 CU is bound by its kernels, each streaming a whole layer's two weight planes through a handful of
    workgroups, not by launch overhead -- and was dropped; the test stays.)"""
    import torch
    pkg = gpu
    for size, seed in ((64, 10), (128, 10)):
        arch = pkg.synth.arch_for_size(size)
        blob = pkg.weights.synthetic_blob(arch, seed)
        n = 64
        org, pred = pkg.synth.make_patches_bulk(size, n, 9301)
        of, pf = pkg.synth.make_patches(size, 12, 9302, pkg.synth.KIND_FLAT)
        flat_at = [1, 5, 9, 17, 18, 30, 31, 40, 41, 42, 50, 63]
        for j, i in enumerate(flat_at):
            org[i], pred[i] = of[j], pf[j]
        poc, qp = pkg.synth.make_scalars(n, 9303)
        m = _ctx(pkg, size, blob, max_batch=256)
        a = m.arithmetic(size)
        if a["exact"] == 1:
            m.close()
            continue   # (a size the calibration left exact has no guards)
        r0 = m.arithmetic(size)["guard_reruns"]
        s_all, l_all = m.predict_batch(org, pred, poc, qp)          # >= 12 flagged: eager re-run
        assert m.arithmetic(size)["guard_reruns"] - r0 >= 12
        ex = _ctx(pkg, size, blob, flags=pkg.capi.FLAG_EXACT_128 if size == 128 else pkg.capi.FLAG_NO_CALIBRATION)
        se, le = ex.predict_batch(org[flat_at], pred[flat_at], poc[flat_at], qp[flat_at])
        ex.close()
        assert np.array_equal(l_all[flat_at], le) and np.array_equal(s_all[flat_at], se)
        for lo, hi in ((0, 4), (0, 8), (0, 16), (16, 20), (38, 44), (0, 4), (16, 20)):   # 1, 2, 3, 2, 3 flat CUs (+ whatever the decision guard adds); repeats replay
            for rep in range(3):
                r0 = m.arithmetic(size)["guard_reruns"]
                s, l = m.predict_batch(org[lo:hi], pred[lo:hi], poc[lo:hi], qp[lo:hi])
                k = m.arithmetic(size)["guard_reruns"] - r0
                assert 1 <= k <= 8, (size, lo, hi, k)
                assert np.array_equal(l, l_all[lo:hi]) and np.array_equal(s, s_all[lo:hi]), (size, lo, hi, rep)
        # a larger batch re-allocates the workspace: the graphs captured on the old one must not be replayed
        big_o, big_p = pkg.synth.make_patches_bulk(size, 256, 9304)
        big_c, big_q = pkg.synth.make_scalars(256, 9304)
        m.predict_batch(big_o, big_p, big_c, big_q)
        for rep in range(2):
            s, l = m.predict_batch(org[0:8], pred[0:8], poc[0:8], qp[0:8])
            assert np.array_equal(l, l_all[0:8]) and np.array_equal(s, s_all[0:8])
        # device-pointer entry on torch's stream (a stream the context does not own: the capture runs on the context's own capture stream)
        dev = torch.device("cuda", 0)
        d = [torch.from_numpy(np.ascontiguousarray(x[0:16])).to(dev) for x in (org, pred, poc, qp)]
        d_split = torch.full((16,), -7, dtype=torch.int32, device=dev)
        d_lg = torch.zeros((16, m.num_logits(size)), dtype=torch.float32, device=dev)
        m.set_stream(torch.cuda.current_stream().cuda_stream)
        for rep in range(3):
            m.predict_batch_device(16, size, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), d[3].data_ptr(), d_split.data_ptr(), d_lg.data_ptr())
            torch.cuda.synchronize()
            assert np.array_equal(d_split.cpu().numpy(), s_all[0:16]) and np.array_equal(d_lg.cpu().numpy(), l_all[0:16]), rep
        m.set_stream(None)
        # one CU per call and the deferred path
        for i in (1, 5, 2, 1):
            s1, l1 = m.predict(org[i], pred[i], int(poc[i]), int(qp[i]))
            assert s1 == s_all[i] and np.array_equal(l1, l_all[i]), i
        for rep in range(2):
            tk = [m.submit(org[i], pred[i], int(poc[i]), int(qp[i])) for i in range(36, 44)]
            m.flush(size)
            for j, t in enumerate(tk):
                s1, l1 = m.wait(size, t)
                assert s1 == s_all[36 + j] and np.array_equal(l1, l_all[36 + j]), ("deferred", 36 + j)
        m.close()


def test_magnitude_guard(gpu):
    """Round 6.  A weight set that amplifies the residual plane -- what training leaves behind; here the deterministic stand-in
    weights.amplifying_blob -- has logits, and absolute fp16 errors, many times larger on content with huge residuals (the calibration classes
    "uniform", "constant org / pred") than on ordinary content: the plain admission rule keeps it out of every fp16 tier (round 5: exact-lite at
    a quarter of the headline).  The error is RELATIVE to the logit magnitude M (heads_kernel: max over logits of sum_k |w_ck gap_k|), so the set
    is admitted BEHIND the magnitude guard: CUs with M above the threshold are re-evaluated exactly, the others meet the contract.
    Checked: the tier and its figures, the threshold's meaning, the oracle on mixed content (ordinary CUs on the fp16 tier, huge-residual CUs
    re-run), the same bits through every entry point, and MLT_FLAG_NO_MAGNITUDE_GUARD = the round-5 outcome."""
    from oracle import Oracle
    pkg = gpu
    size = 128
    blob = pkg.weights.amplifying_blob(0, 10)
    m = _ctx(pkg, size, blob)
    a = m.arithmetic(size)
    print("behind the magnitude guard:", a)
    assert a["calibrated"] == 1 and a["exact"] in (0, 2, 3, 4) and a["mag_guard_thr"] > 0.0 and a["mag_guard_kind"] == 2, a
    assert 0.0 <= a["mag_guard_flagged"] <= 0.05 and 5.5 * a["calib_rms"] <= 1e-3 and a["calib_max"] <= 0.65e-3 and a["calib_cus"] == 560
    plain = _ctx(pkg, size, blob, flags=pkg.capi.FLAG_NO_MAGNITUDE_GUARD)
    ap = plain.arithmetic(size)
    print("plain rule:", ap)
    assert ap["mag_guard_thr"] == 0.0 and ap["mag_guard_kind"] == 0 and ap["exact"] in (1, 4, 5), ap          # (what round 5 gave this family: exact stages, exact-lite or exact)
    # mixed content: texture (ordinary), natural scenes, uniform noise / constant planes (huge residuals)
    nt, nn, nu = 96, 64, 24
    ot, pt = pkg.synth.make_patches_bulk(size, nt, 8101)
    on, pn = pkg.synth.natural_patches(size, nn, 8102)
    ou, pu = pkg.synth.make_patches(size, nu, 8103, pkg.synth.KIND_UNIFORM)
    oc, pc = pkg.synth.make_patches(size, nu, 8104, pkg.synth.KIND_ORG_FLAT_PRED_TEX)
    org = np.concatenate([ot, on, ou, oc]); pred = np.concatenate([pt, pn, pu, pc])
    n = len(org)
    poc, qp = pkg.synth.make_scalars(n, 8105)
    ref, ref_split = Oracle(blob).forward(org, pred, poc, qp, threads=8)
    r0 = m.arithmetic(size)["guard_reruns"]
    s, l = m.predict_batch(org, pred, poc, qp)
    reruns = m.arithmetic(size)["guard_reruns"] - r0
    err = np.abs(l - ref).max(axis=1)
    print(f"max |dlogit|: texture {err[:nt].max():.2e} natural {err[nt:nt + nn].max():.2e} uniform {err[nt + nn:nt + nn + nu].max():.2e} constant org {err[-nu:].max():.2e}; "
          f"{reruns} of {n} CUs re-run exactly; |logit| max {np.abs(ref).max():.1f}")
    assert err.max() <= LOGIT_TOL
    check_splits(s, ref, ref_split, head_slices([2, 3, 4])[2], True, LOGIT_TOL, "magnitude guard")
    assert reruns >= 2 * nu - 4, "the huge-residual CUs are what the guard exists for"
    assert reruns <= 2 * nu + (nt + nn) // 4, "ordinary content must stay on the fp16 tier"
    # the re-run CUs carry the exact arithmetic's bits
    ex = _ctx(pkg, size, blob, flags=pkg.capi.FLAG_EXACT_128)
    se, le = ex.predict_batch(org[-nu:], pred[-nu:], poc[-nu:], qp[-nu:])
    same = (l[-nu:] == le).all(axis=1)
    assert same.sum() >= nu - 2, same
    # every entry point: one CU per call, the deferred path, sub-batches
    for i in (0, nt + 3, nt + nn + 1, n - 1):
        s1, l1 = m.predict(org[i], pred[i], int(poc[i]), int(qp[i]))
        assert s1 == s[i] and np.array_equal(l1, l[i]), i
    tk = [m.submit(org[i], pred[i], int(poc[i]), int(qp[i])) for i in range(nt + nn - 4, nt + nn + 8)]
    m.flush(size)
    for j, i in enumerate(range(nt + nn - 4, nt + nn + 8)):
        s1, l1 = m.wait(size, tk[j])
        assert s1 == s[i] and np.array_equal(l1, l[i]), ("deferred", i)
    s2, l2 = m.predict_batch(org[nt - 8:nt + nn + 8], pred[nt - 8:nt + nn + 8], poc[nt - 8:nt + nn + 8], qp[nt - 8:nt + nn + 8])
    assert np.array_equal(l2, l[nt - 8:nt + nn + 8]) and np.array_equal(s2, s[nt - 8:nt + nn + 8])
    # an ABI-4 caller's 72-byte mlt_arith_info (before the two magnitude-guard floats) is still accepted and nothing is written behind it
    import ctypes as C
    raw = (C.c_ubyte * 104)(*([0xAB] * 104))
    C.cast(raw, C.POINTER(C.c_uint32))[0] = 72
    lib = pkg.capi.load_library()
    assert lib.mlt_arithmetic(m._h, size, C.cast(raw, C.POINTER(pkg.capi.MltArithInfo))) == 0
    assert all(b == 0xAB for b in raw[72:]) and C.cast(raw, C.POINTER(C.c_int32))[1] == a["exact"]
    C.cast(raw, C.POINTER(C.c_uint32))[0] = 64
    assert lib.mlt_arithmetic(m._h, size, C.cast(raw, C.POINTER(pkg.capi.MltArithInfo))) == 1     # MLT_ERR_ARG: does not cover the ABI-4 fields
    # the seeded bench set is admitted by the plain rule: it gets the RANGE guard only -- 1.5 x the largest magnitude of its calibration CUs, which
    # nothing of the calibrated content classes reaches (no re-run on 96 texture + 64 natural + 24 uniform CUs) -- and keeps the flat guard at 1/8
    b10 = _ctx(pkg, size, pkg.weights.synthetic_blob(0, 10))
    a10 = b10.arithmetic(size)
    assert a10["exact"] == 0 and a10["mag_guard_kind"] == 1 and a10["mag_guard_thr"] > 0.0 and a10["mag_guard_flagged"] == 0.0
    r0 = a10["guard_reruns"]
    b10.predict_batch(org[:nt + nn + nu], pred[:nt + nn + nu], poc[:nt + nn + nu], qp[:nt + nn + nu])
    assert b10.arithmetic(size)["guard_reruns"] - r0 <= 6, "the range guard must not touch content inside the calibrated range (a few flat / near-tie CUs aside)"
    for c in (m, plain, ex, b10):
        c.close()
