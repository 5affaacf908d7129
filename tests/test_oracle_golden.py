"""CPU: the C oracle (oracle/mlt_oracle.c) against the fixtures produced by the reference's
own PyTorch modules (tools/gen_golden.py).  Tolerance: fp32 rounding, 2e-5 * max(1,|logit|max)
(summation order differs between ATen NCHW and the oracle's NHWC loops)."""
import numpy as np
import pytest

from helpers import SIZES, decisive, head_slices, load_golden, materialise


@pytest.mark.parametrize("size", SIZES)
def test_oracle_matches_reference_fixtures(pkg, size):
    from oracle import Oracle
    golden = load_golden(size)
    worst = 0.0
    for case in golden["cases"]:
        blob, org, pred, poc, qp, exp, exp_arg = materialise(pkg, golden, case)
        orc = Oracle(blob)
        logits, split = orc.forward(org, pred, poc, qp, threads=8)
        tol = 2e-5 * max(1.0, float(np.abs(exp).max()))
        err = float(np.abs(logits - exp).max())
        worst = max(worst, err)
        assert err <= tol, f"{case['name']}: |dlogit| {err:.3e} > {tol:.1e}"
        sls = head_slices(orc.head_classes)
        dec = 2 if size == 128 else 0  # EncCu.cpp:913-919
        for i in range(case["n"]):
            for h, sl in enumerate(sls):
                if decisive(exp[i], sl, 4e-5):   # (twice the fp32 restatement's own distance from the reference; the near-tie family of round 4 lives just above it)
                    assert int(np.argmax(logits[i, sl])) == exp_arg[i][h], (case["name"], i, h)
            if decisive(exp[i], sls[dec], 4e-5):
                assert split[i] == exp_arg[i][dec]
    print(f"size {size}: worst |dlogit| {worst:.2e}")


def test_oracle_first_max_tie_rule(pkg):
    """torch.argmax returns the first maximal index (EncCu.cpp:921); identical head rows tie exactly
    in the oracle because both rows run the same fp32 operation sequence."""
    from oracle import Oracle
    from helpers import variant_state_dict
    for size in (128, 16):
        arch = pkg.synth.arch_for_size(size)
        blob = pkg.weights.pack_blob(arch, variant_state_dict(pkg, arch, 10, "tie", size))
        org, pred = pkg.synth.make_patches(size, 3, 77)
        poc, qp = pkg.synth.make_scalars(3, 77)
        logits, split = Oracle(blob).forward(org, pred, poc, qp)
        sl = head_slices(Oracle(blob).head_classes)[2 if size == 128 else 0]
        assert np.all(logits[:, sl][:, 0] == logits[:, sl][:, 1])
        assert np.all(split == 0)


def test_oracle_strided_and_batched_consistency(pkg):
    """Strided views (picture-buffer rows, EncCu.cpp:816: stride = picture stride) and batch order."""
    from oracle import Oracle
    size = 32
    blob = pkg.weights.synthetic_blob(1, 10)
    orc = Oracle(blob)
    org, pred = pkg.synth.make_patches(size, 4, 5)
    poc, qp = pkg.synth.make_scalars(4, 5)
    ref, _ = orc.forward(org, pred, poc, qp)
    pic = np.zeros((4, size, size + 24), np.int16) - 7
    pic[:, :, 8:8 + size] = org
    got, _ = orc.forward(pic[:, :, 8:8 + size], pred, poc, qp)
    assert np.array_equal(ref, got)
    single = np.concatenate([orc.forward(org[i:i + 1], pred[i:i + 1], poc[i:i + 1], qp[i:i + 1])[0] for i in range(4)])
    assert np.array_equal(ref, single)


def test_oracle_rejects_bad_size(pkg):
    from oracle import Oracle
    orc = Oracle(pkg.weights.synthetic_blob(0, 10))
    org, pred = pkg.synth.make_patches(64, 1, 5)
    with pytest.raises(RuntimeError):
        orc.forward(org, pred, [0], [30])


@pytest.mark.parametrize("size", SIZES)
def test_torch_cpu_port_matches_reference_fixtures_and_c_oracle(pkg, size):
    """oracle/torch_port.py (the CPU baseline bench.py times beside the C oracle) against the reference-generated fixtures
    and against the C oracle on a seeded batch."""
    import oracle
    from oracle.torch_port import TorchPort
    g = load_golden(size)
    for case in g["cases"]:
        blob, org, pred, poc, qp, exp, am = materialise(pkg, g, case)
        lg, sp = TorchPort(blob).forward(org, pred, poc, qp, threads=4)
        assert np.abs(lg - exp).max() <= 2e-5 * max(1.0, np.abs(exp).max()), case["name"]
    blob = pkg.weights.synthetic_blob(pkg.synth.arch_for_size(size), 77)
    n = 3 if size == 128 else 8
    org, pred = pkg.synth.make_patches(size, n, 4242)
    poc, qp = pkg.synth.make_scalars(n, 4242)
    a, sa = oracle.Oracle(blob).forward(org, pred, poc, qp)
    b, sb = TorchPort(blob).forward(org, pred, poc, qp, threads=4)
    assert np.abs(a - b).max() <= 2e-5 * max(1.0, np.abs(a).max())
