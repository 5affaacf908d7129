"""CPU: the dispatcher's LAUNCH PLANS.  run_network (csrc/mlt_api.cpp) decides, per batch, which kernels serve which launch units --
tier x batch class x alignment x which neighbour is a whole-stage kernel (layouts between them) -- and round 5 left that decision observable only on a
GPU.  The library's plan mode (mlt_ctx::plan) runs the same code with every launch RECORDED instead of enqueued; mlt_plan_describe (host-only hook, no
device) returns the list.  Pinned here: the launches of the shipped tiers at the batch classes the encoder and the bench use, and the layout hand-offs
(chunk-major outputs exactly where the consumer is a whole-stage kernel).  Round 6, second half: the decision is a pure function of (models, unit masks,
batch class) -- plan_network -- whose result the hot path caches and walks; the detailed records (aligned | 2) carry every launch's buffers, so the
hand-offs between launches (what one writes is what the next reads, lo planes exactly where an exact unit is on either side) are checked here too.
scripts/plan_matrix.py dumps the detailed plans of ~9000 (size, batch, tier, masks, alignment) cases: the before / after check of a dispatcher change."""
import re
import ctypes as C

import pytest


@pytest.fixture(scope="module")
def plan(pkg):
    pkg.build.build_lib()
    lib = pkg.capi.load_library()
    lib.mlt_plan_describe.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_uint, C.c_uint, C.c_int, C.c_char_p, C.c_size_t]
    blobs = {}

    def go(size, n, tier=0, w2_units=0, x_units=0, aligned=1, seed=10):
        arch = pkg.synth.arch_for_size(size)
        if (arch, seed) not in blobs:
            blobs[(arch, seed)] = pkg.weights.synthetic_blob(arch, seed)
        b = blobs[(arch, seed)]
        buf = C.create_string_buffer(1 << 15)
        k = lib.mlt_plan_describe(b, len(b), size, n, tier, w2_units, x_units, aligned, buf, 1 << 15)
        lines = buf.value.decode().splitlines()
        assert k == len(lines) and k > 0, (k, lines)
        return lines
    return go


def names(lines):
    return [l.split(" [")[0] for l in lines]


def test_single_pass_batches_are_four_fused_launches_and_the_heads(plan):
    p = plan(128, 4096)
    assert names(p) == ["layer0_stream_h64(stem+layer0+layer1.0.conv1+sc)", "layer1_stream_h32(conv2+conv1+conv2)", "stage_128_h16(s2+sc,conv2,conv1,conv2)",
                        "stage_256_h8(s2+sc,conv2,conv1,conv2)", "heads"]
    # layouts: every producer of a whole-stage kernel's input writes chunk-major, the last stage writes no activation at all (GAP sums only)
    assert "y chunk-major" in p[1] and "x chunk-major" in p[2] and "y chunk-major" in p[2] and "x chunk-major" in p[3] and "y chunk-major" not in p[3]
    assert all("GAP" in l for l in p[1:4])
    # round 6: the streaming launches start at 128 CUs (the measured crossover), not 256
    assert names(plan(128, 128))[:2] == names(p)[:2]
    assert names(plan(128, 127))[:4] == ["stem+block_s2_2to32_h64(layer0.0)", "block_s1_32_h64(conv1+conv2)", "conv3x3_s2_32to64_h32+sc", "chain3_s1_64_h32(conv2+conv1+conv2)"]


def test_one_cu_call_runs_the_latency_variants(plan):
    p = plan(128, 1)
    assert len(p) == 15 and names(p)[:3] == ["stem+block_s2_2to32_h64(layer0.0)", "block_s1_32_h64(conv1+conv2)", "conv3x3_s2_32to64_h32+sc"]
    assert all("latency tiles" in l for l in p[3:14]) and names(p)[-1] == "heads"
    assert not any("chunk-major" in l for l in p)          # per-conv launches hand NHWC to each other
    # 8 CUs (an N3 flush at 1080p): the same 15 launches
    assert names(plan(128, 8)) == names(p)


def test_hi_lo_weight_units_compose_with_the_single_pass_kernels(plan):
    # layer1 in hi+lo weights (units 2, 3): the fifth stage stays out of the layer0 launch, the stride-2 conv and the 64-channel chain run their two-plane forms
    p = plan(128, 4096, w2_units=0xC)
    assert names(p) == ["layer0_stream_h64(stem+layer0.0+layer0.1)", "conv3x3_s2_32to64_h32+sc", "chain3_s1_64_h32(conv2+conv1+conv2)",
                        "stage_128_h16(s2+sc,conv2,conv1,conv2)", "stage_256_h8(s2+sc,conv2,conv1,conv2)", "heads"]
    assert "hi+lo weights" in p[1] and "hi+lo weights" in p[2] and "single pass" in p[3]
    # layer2 in hi+lo weights: no whole-stage form there (the stride-2 conv is a launch of its own), so layer1 writes NHWC and layer2 chunk-major for layer3
    p = plan(128, 4096, w2_units=0x30)
    assert names(p)[2:4] == ["conv3x3_s2_64to128_h16+sc", "chain3_s1_128_h16(conv2+conv1+conv2)"]
    assert "chunk-major" not in p[1] and "y chunk-major" in p[3] and "x chunk-major" in p[4]
    # the first trained family's tier (round 6): hi+lo weights in layer0.0 and layer1's stride-2 conv, the streaming kernels on the rest
    p = plan(128, 4096, w2_units=0x5)
    assert names(p)[:4] == ["stem+block_s2_2to32_h64(layer0.0)", "block_s1_32_h64(conv1+conv2)", "conv3x3_s2_32to64_h32+sc", "layer1_stream_h32(conv2+conv1+conv2)"]
    assert "hi+lo weights" in p[0] and "single pass" in p[1] and "hi+lo weights" in p[2]


def test_exact_and_exact_lite_are_per_conv_launches(plan):
    for tier, word in ((1, "exact"), (5, "exact-lite")):
        p = plan(128, 4096, tier=tier)
        assert len(p) == 17 and names(p)[0] == "stem5x5_s2_2to32_h64+sc" and names(p)[-1] == "heads"
        assert all(f"[{word}" in l for l in p[1:16]), p
    # an exact unit inside an fp16 tier: layer2 exact behind single-pass layer1 (the `exact = 4` tiers)
    p = plan(128, 4096, x_units=0x30)
    assert names(p)[:2] == ["layer0_stream_h64(stem+layer0+layer1.0.conv1+sc)", "layer1_stream_h32(conv2+conv1+conv2)"]
    assert [("[exact" in l) for l in p[2:6]] == [True] * 4 and "stage_256" in p[6]


def test_unaligned_planes_fall_back_to_the_two_step_front(plan):
    """stem_block_kernel / layer0_stream_kernel fetch 4-pixel quads with 8-byte loads; planes that are not 8-byte aligned (a picture buffer at an odd
    offset) run stem5_kernel + per-conv layer0.0 + block32 -- and the flat-content statistic comes from its own kernel."""
    p = plan(128, 4096, aligned=0)
    assert names(p)[:5] == ["guard_flat_stat", "stem5x5_s2_2to32_h64+sc", "conv3x3_s1_32to32_h64", "block_s1_32_h64(conv1+conv2)", "conv3x3_s2_32to64_h32+sc"]
    assert names(p)[5:] == ["layer1_stream_h32(conv2+conv1+conv2)", "stage_128_h16(s2+sc,conv2,conv1,conv2)", "stage_256_h8(s2+sc,conv2,conv1,conv2)", "heads"]


def test_small_models(plan):
    p = plan(64, 4096, tier=1)
    assert len(p) == 21 and all("[exact" in l for l in p[:20])
    # the 16 x 16 model's calibrated tier: layer0 on the single pass, exact from layer1 on; 1 x 1 maps take the centre-tap kernels
    p = plan(16, 4096, x_units=0x3FC)
    assert names(p)[:2] == ["guard_flat_stat", "stem5x5_s2_2to32_h8+sc"] and all("single pass" in l for l in p[1:5]) and all("[exact" in l for l in p[5:21])
    assert sum("centre tap" in l for l in p) == 7


def args(line):
    """{'x': int, ...} of a detailed record."""
    m = re.search(r"\{(.*)\}$", line)
    return {k: int(v, 0) for k, v in (kv.split("=") for kv in m.group(1).split())}


def test_buffers_are_handed_from_launch_to_launch(plan):
    # the four fused launches of a batch: layer0's streaming launch writes t and sc, layer1's reads them; every stage reads its predecessor's output; the heads read
    # the three GAP buffers the last launches of layer1 .. layer3 wrote
    p = plan(128, 4096, aligned=3)
    a = [args(l) for l in p]
    assert a[0]["y_t"] == a[1]["t"] and a[0]["y_sc"] == a[1]["sc"] and "y" not in a[0]          # layer0's own output never reaches HBM
    assert a[1]["y"] == a[2]["x"] and a[2]["y"] == a[3]["x"] and "y" not in a[3]                # the last stage writes GAP sums only
    assert [a[4][f"gap{i}"] for i in range(3)] == [a[1]["gap"], a[2]["gap"], a[3]["gap"]]
    assert (a[4]["slots0"], a[4]["slots1"], a[4]["slots2"]) == (32, 8, 2) and (a[4]["hw0"], a[4]["hw1"], a[4]["hw2"]) == (1024, 256, 64) and a[4]["heads"] == 3
    # every buffer of the workspace is distinct where it must be: t, sc, the three stage outputs, the GAP sums
    bufs = [a[0]["y_t"], a[0]["y_sc"], a[1]["y"], a[2]["y"], a[1]["gap"], a[2]["gap"], a[3]["gap"]]
    assert len(set(bufs)) == len(bufs)
    # the one-CU call: a chain of per-conv launches, each reading what the one before wrote; residuals come from two launches back
    q = [args(l) for l in plan(128, 1, aligned=3)]
    assert q[1]["x"] == q[0]["y"] and q[2]["x"] == q[1]["y"]
    for i in (3, 7, 11):                                      # conv2 of block 0: t -> b0, + sc
        assert q[i]["x"] == q[i - 1]["y"] and q[i]["res"] == q[i - 1]["y_sc"]
        assert q[i + 1]["x"] == q[i]["y"] and q[i + 2]["x"] == q[i + 1]["y"] and q[i + 2]["res"] == q[i]["y"]   # block 1: conv1, conv2 + b0
        if i < 11:
            assert q[i + 3]["x"] == q[i + 2]["y"]             # next stage's stride-2 conv reads the stage output
    assert "y" not in q[13] and q[14]["gap2"] == q[13]["gap"]


def test_lo_planes_exist_exactly_where_an_exact_unit_is_involved(plan):
    # layer1 hi+lo weights, layer2 + layer3 exact (units 4 .. 7): the exact stride-2 conv of layer2 reads a single-plane tensor (x_lo = 0: "no lo plane") and writes
    # two planes; inside the exact stages every tensor has its lo plane one plane's bytes behind the hi plane
    p = plan(128, 4096, tier=4, w2_units=0xC, x_units=0xF0, aligned=3)
    a = [args(l) for l in p]
    n = 4096
    s2 = a[3]
    assert "exact" in p[3] and s2["x_lo"] == 0 and s2["y_lo"] == s2["ysc_lo"] == n * 16 * 16 * 128 * 2
    for r in a[4:7]:
        assert r["x_lo"] == r["y_lo"] == n * 16 * 16 * 128 * 2
    s2 = a[7]
    assert s2["x_lo"] == n * 16 * 16 * 128 * 2 and s2["y_lo"] == n * 8 * 8 * 256 * 2      # layer3's input HAS a lo plane (its producer is exact)
    # the single-plane launches before it carry no lo offsets at all
    assert a[1]["x_lo"] == a[1]["y_lo"] == a[1]["ysc_lo"] == 0
    # an exact unit 1 behind a single-plane unit 0 (layer2.s1 exact only): t, sc and the stage input have no lo plane, b0 and everything behind it does
    q = [args(l) for l in plan(128, 4096, tier=4, x_units=0x20, aligned=3)]
    convs = [r for r, l in zip(q, plan(128, 4096, tier=4, x_units=0x20, aligned=3)) if "128to128" in l]
    assert convs[0]["x_lo"] == 0 and convs[0]["res_lo"] == 0 and convs[0]["y_lo"] > 0
    assert convs[1]["x_lo"] == convs[1]["y_lo"] == convs[0]["y_lo"] and convs[2]["res_lo"] == convs[0]["y_lo"]


def test_committed_traffic_figures_cover_the_headline_launches(plan, pkg):
    """bench.py looks a launch's HBM traffic up in profiles/pmc_traffic.json by the name the launch is profiled under (47 characters of it: mlt_kernel_time.name) +
    "@<batch>".  Round 6's evidence runs r06A .. r06C were post-processed by a script that still mapped layer0_stream_kernel<F5, M16> by its old one-argument
    template signature and filed the headline's roofline kernel under the four-stage form's name: `roofline.traffic` came out null although the figures were there.
    Pinned: while the committed file belongs to the tree's sources, every fused launch of the headline batch has its entry."""
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    path = os.path.join(root, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        pytest.skip("no committed traffic figures")
    tj = json.load(open(path))
    if tj.get("_meta", {}).get("source_sig") != pkg.build.source_signature():
        pytest.skip("profiles/pmc_traffic.json was measured on other kernel sources (bench.py then reports traffic = null and says so)")
    for name in names(plan(128, 4096))[:4]:
        key = f"{name[:47]}@4096"
        assert key in tj and tj[key]["hbm_bytes_per_launch"] > 0 and tj[key]["dispatches_averaged"] == 2, key
