"""tools/eval_harness.py: VTM log parsing, BD-rate, time saving -- closed-form checks."""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("eval_harness", os.path.join(ROOT, "tools", "eval_harness.py"))
eh = importlib.util.module_from_spec(spec)
spec.loader.exec_module(eh)


def _log(frames, rate, y, u, v, yuv, user, elapsed, hdr=False):
    timer = (f" Encoding Time (Total Time): {user:12.3f} ( {user + 1:12.3f} ) sec. [user] {elapsed:12.3f} ( {elapsed + 1:12.3f} ) sec. [elapsed]\n"
             if hdr else f" Total Time: {user:12.3f} sec. [user] {elapsed:12.3f} sec. [elapsed]\n")
    return ("POC    0 TId: 0 ( IDR_N_LP, I-SLICE, QP 27 )     123456 bits [Y 40.1 dB    U 42.0 dB    V 42.5 dB]\n\n\nSUMMARY --------------------------------------------------------\n"
            "LayerId  0\n\n\tTotal Frames |   Bitrate     Y-PSNR    U-PSNR    V-PSNR    YUV-PSNR \n"
            f"\t{frames:9d}    a {rate:12.4f}   {y:8.4f}   {u:8.4f}   {v:8.4f}   {yuv:8.4f}\n\n\n\tI Slices--------------------------------------------------------\n"
            "\tTotal Frames |   Bitrate     Y-PSNR    U-PSNR    V-PSNR    YUV-PSNR \n"
            f"\t{1:9d}    i {rate * 3:12.4f}   {y + 2:8.4f}   {u:8.4f}   {v:8.4f}   {yuv:8.4f}\n\n finished @ Thu Jan  1 00:00:00 1970\n" + timer)


def test_parse_vtm_log_both_timer_formats():
    for hdr in (False, True):
        s = eh.parse_vtm_log(_log(50, 320.2912, 33.8790, 39.4090, 39.1925, 35.0427, 1234.567, 1300.25, hdr))
        assert s["frames"] == 50 and s["bitrate_kbps"] == pytest.approx(320.2912) and s["psnr_y"] == pytest.approx(33.8790)
        assert s["psnr_yuv"] == pytest.approx(35.0427) and s["time_user_s"] == pytest.approx(1234.567) and s["time_elapsed_s"] == pytest.approx(1300.25)
    with pytest.raises(ValueError):
        eh.parse_vtm_log("no summary here\n")


def _curve(scale=1.0, shift=0.0):
    # a plausible RD curve: PSNR = 30 + 6 log10(rate / 100); rate points of QP 37..22
    rates = np.array([150.0, 400.0, 1100.0, 3000.0])
    return [(r * scale, 30.0 + 6.0 * np.log10(r / 100.0) + shift) for r in rates]


@pytest.mark.parametrize("method", ["pchip", "poly"])
def test_bd_rate_closed_forms(method):
    a = _curve()
    assert eh.bd_rate(a, a, method) == pytest.approx(0.0, abs=1e-9)
    # same quality at 10 % more / 20 % fewer bits everywhere
    assert eh.bd_rate(a, _curve(scale=1.10), method) == pytest.approx(10.0, abs=1e-6)
    assert eh.bd_rate(a, _curve(scale=0.80), method) == pytest.approx(-20.0, abs=1e-6)
    # on a log-linear curve a PSNR offset of d dB equals a rate factor 10^(-d/6)
    assert eh.bd_rate(a, _curve(shift=-0.3), method) == pytest.approx((10 ** (0.3 / 6.0) - 1) * 100, rel=2e-3)
    # antisymmetry in the log domain
    f = 1 + eh.bd_rate(a, _curve(scale=1.25), method) / 100
    g = 1 + eh.bd_rate(_curve(scale=1.25), a, method) / 100
    assert f * g == pytest.approx(1.0, abs=1e-9)
    with pytest.raises(ValueError):
        eh.bd_rate(a[:3], a, method)


def test_time_saving_and_directory_compare(tmp_path):
    assert eh.time_saving([100, 200], [50, 100]) == pytest.approx(50.0)
    assert eh.time_saving([100, 100, 100, 100], [70, 60, 50, 40]) == pytest.approx(45.0)
    ad, td = tmp_path / "anchor", tmp_path / "test"
    ad.mkdir(); td.mkdir()
    for qp, (r, q) in zip((37, 32, 27, 22), _curve()):
        (ad / f"BasketballPass_q{qp}.txt").write_text(_log(50, r, q, 40, 41, q + 1, 1000.0 + qp, 1010.0))
        (td / f"BasketballPass_q{qp}.txt").write_text(_log(50, r * 1.02, q, 40, 41, q + 1, (1000.0 + qp) * 0.6, 700.0))
    (td / "notes.md").write_text("ignored")
    rows = eh.compare(str(ad), str(td))
    assert len(rows) == 1 and rows[0][0] == "BasketballPass" and rows[0][1] == [22, 27, 32, 37]
    assert rows[0][2] == pytest.approx(2.0, abs=1e-6) and rows[0][3] == pytest.approx(40.0, abs=1e-9)
    assert eh.main(["eval_harness.py", str(ad), str(td)]) == 0
