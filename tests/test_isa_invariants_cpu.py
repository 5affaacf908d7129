"""ISA-level invariants of the whole-stage / chain kernels (hipcc cross-compiles gfx950 here, no GPU needed).

What these guard ("compiler-made drains"): a scratch reload or a plain load whose first use sits inside a tile / step
loop is preceded by the compiler's s_waitcnt vmcnt(0), which also drains every LDS-DMA piece in flight -- a serialised HBM round
trip per occurrence.  Round 2 removed them from the three default chain kernels; this keeps them out."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "scripts"))

STAGE_128 = "chain_kernel<128, 4, 0, 2, 2, 2, 4, 3, 2, 2, 2, 3, true, true, true, %s, false, false>"
STAGE_256 = "chain_kernel<256, 3, 1, 2, 1, 2, 4, 3, 2, 2, 2, 3, true, true, true, %s, false, false>"
CHAIN_64 = "chain_kernel<64, 5, 0, 2, 4, 1, 8, 2, 2, 1, 2, 3, true, false, false, %s, false, false>"
# hi+lo-weights forms (round 4): two weight planes per ring step, no stride-2 front conv, padding from beyond the LDS only
W2_STAGE_128 = "chain_kernel<128, 4, 0, 2, 2, 2, 4, 1, 2, 1, 2, 3, true, false, true, true, true, true>"
W2_STAGE_256 = "chain_kernel<256, 3, 1, 2, 1, 2, 4, 1, 2, 1, 2, 3, true, false, true, true, true, true>"
W2_CHAIN_64 = "chain_kernel<64, 5, 0, 2, 4, 1, 8, 1, 2, 1, 2, 3, true, false, false, true, true, false>"
FORMS = ("true", "false")  # conv padding from beyond the LDS (the probed default) / from zero masks (fallback)


@pytest.fixture(scope="module")
def stats():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("hipcc not available")
    import isa_waits
    return isa_waits.collect([])


def _find(stats, prefix):
    hits = [v for k, v in stats.items() if k.startswith(prefix)]
    assert len(hits) == 1, f"{prefix}: {len(hits)} instantiations"
    return hits[0]


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("kernel", [STAGE_128, STAGE_256])
def test_whole_stage_kernels_have_no_scratch_and_no_compiler_drain_in_loops(stats, kernel, form):
    kernel = kernel % form
    st = _find(stats, kernel)
    assert st["scratch"] == 0, f"{kernel}: {st['scratch']} scratch ops (register spills)"
    inner = [w for w in st["waits"] if w[1] >= 1]
    assert not inner, f"{kernel}: compiler-generated vmcnt waits inside loops: {inner[:5]}"
    assert st["glds"] > 50  # the LDS-DMA staging is there (guards against matching a different kernel)


@pytest.mark.parametrize("form", FORMS)
def test_64_channel_chain_spills_stay_out_of_the_step_loops(stats, form):
    st = _find(stats, CHAIN_64 % form)
    assert st["scratch"] <= 8, f"{st['scratch']} scratch ops (round 2: 4, all at the tile end)"
    deep = [w for w in st["waits"] if w[1] >= 2]
    assert not deep, f"compiler-generated vmcnt waits inside the step loops: {deep[:5]}"
    assert len([w for w in st["waits"] if w[1] >= 1]) <= 4


def test_stem_block_pipeline_has_no_scratch_and_no_drain_inside_its_block_loops(stats):
    """stem_block_kernel (round 3: producer / consumer pipeline): no spills, and the only compiler-made vmcnt waits inside loops are the
    raw-patch commit's at the top of an interval (tile-loop depth 1) -- none inside the 32-pixel block loops (depth 2), where a wait would
    stall a wave that has no partner on its SIMD to hide it and would drain the next tile's raw-plane prefetch."""
    st = _find(stats, "stem_block_kernel<false>")
    assert st["scratch"] == 0, f"{st['scratch']} scratch ops"
    deep = [w for w in st["waits"] if w[1] >= 2]
    assert not deep, f"compiler-generated vmcnt waits inside the block loops: {deep[:5]}"
    assert 0 < len([w for w in st["waits"] if w[1] == 1]) <= 24   # the commits of the two roles (<= 2 items x 2 planes each, counted)
    assert st["glds"] >= 2                                          # conv2's weights + the border k-steps arrive by LDS-DMA


@pytest.mark.parametrize("kernel", [W2_STAGE_128, W2_STAGE_256, W2_CHAIN_64])
def test_hi_lo_weights_chains_keep_their_step_loops_free_of_compiler_drains(stats, kernel):
    """chain_kernel<..., W2> (round 4): the stage forms do not spill; the only compiler-made vmcnt waits sit at tile-loop depth (the
    residual tile of conv 0, loaded from HBM by ordinary loads in the forms without the stride-2 front conv) -- none in a step loop."""
    st = _find(stats, kernel)
    assert st["scratch"] <= (2 if kernel is W2_CHAIN_64 else 0), f"{kernel}: {st['scratch']} scratch ops"
    deep = [w for w in st["waits"] if w[1] >= 2]
    assert not deep, f"{kernel}: compiler-generated vmcnt waits inside the step loops: {deep[:5]}"
    assert st["glds"] > 50


@pytest.mark.parametrize("kernel", ["stem_block_kernel<true>", "block32_kernel<3, true>"])
def test_hi_lo_weights_front_kernels_do_not_spill(stats, kernel):
    """The W2 forms of the layer0 kernels read their lo fragments through opaque addresses: left to itself the compiler hoists all 18
    block-invariant LDS reads out of the block loop (72 VGPRs) and spills (180 B of scratch in the first build)."""
    st = _find(stats, kernel)
    assert st["scratch"] == 0, f"{kernel}: {st['scratch']} scratch ops"


# with / without layer1's stride-2 conv as a fifth stage x the MFMA shape of the conv stages (round 6: true = v_mfma_f32_16x16x32_f16, the default)
@pytest.mark.parametrize("form", ["true, true", "true, false", "false, false"])   # (the four-stage form exists on 32x32x16 only: mlt_launch_layer0_stream)
def test_layer0_stream_kernel_keeps_its_raw_prefetch_counted_and_its_step_loops_free_of_scratch(stats, form):
    """layer0_stream_kernel (round 5): 16 waves at 128 VGPRs -- the handful of spilled dwords are stage set-up values, written and read outside
    the step loops (an A fragment reloaded from scratch per row was the first asm-pipelined build's mistake); S1's raw rows are issued RD / 2 .. RD
    steps before they are converted, so the compiler's waits inside its step group are COUNTED (vmcnt(4..12)) apart from the one at the group's
    end where the rows change registers -- a vmcnt(0) per step there made every step wait for an HBM round trip."""
    st = _find(stats, f"layer0_stream_kernel<{form}>")
    assert st["scratch"] <= 6, f"{st['scratch']} scratch ops"   # one dword, stored once and reloaded in the stages' preheaders
    assert st["glds"] == 0
    inner = [w for w in st["waits"] if w[1] >= 1]
    drains = [w for w in inner if "vmcnt(0)" in w[2]]
    # (the fifth-stage form's S1 step has inner labels that end the collector's loop bookkeeping: count its waits at any depth)
    pool = st["waits"]
    counted = [w for w in pool if re.search(r"vmcnt\((\d+)\)", w[2]) and int(re.search(r"vmcnt\((\d+)\)", w[2]).group(1)) >= 4]
    assert len(drains) <= (2 if form.endswith("false") else 5), f"compiler drains inside loops: {drains}"   # the LDS-clearing loop's neighbourhood at the top + the end of S1's step group (+ the 16x16x32 forms' weight gathers in front of their row loops)
    # no scratch op inside a loop (the first 16x16x32 build of the four-stage form had hoisted S2's shortcut weight fragments out of the row loop and
    # reloaded them from scratch every row)
    import isa_waits
    assert not isa_waits.scratch_in_loops(f"layer0_stream_kernel<{form}>"), "scratch ops inside a loop"
    assert len(counted) >= 4, f"S1's raw-row waits are no longer counted: {inner}"


@pytest.mark.parametrize("form", ("true", "false"))
def test_layer1_stream_kernel_fits_its_waves_without_scratch(stats, form):
    """layer1_stream_kernel (round 5): twelve conv waves with 18 resident A fragments each, at 128 VGPRs -- no spills (a reload in a row loop is a
    vector-memory round trip per row), and its loader waves' waits are counted.  Both MFMA shapes (round 6: <true> = v_mfma_f32_16x16x32_f16, the
    default; <false> = round 5's 32x32x16)."""
    st = _find(stats, f"layer1_stream_kernel<{form}>")
    assert st["scratch"] == 0, f"{st['scratch']} scratch ops"
    drains = [w for w in st["waits"] if w[1] >= 1 and "vmcnt(0)" in w[2]]
    assert len(drains) <= 3, f"compiler drains inside loops: {drains}"   # two beside the LDS-clearing loop at the top, one at the end of the loaders' step group
