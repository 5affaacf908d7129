#!/usr/bin/env python3
"""Evaluation harness for the encoder-level metrics of SURVEY.md 8d/8f-N4: BD-rate and encoding-time saving of a
test encoder (VTM + MLT-CNN split prediction through libmltcnn_hip) against an anchor (stock VTM-11.0).

Inputs are the encoder logs the reference's run scripts produce (`script_128/<Seq>_enc_<fps>.sh` redirects every run to
`<Seq>_q<QP>.txt`): the summary block printed by `EncGOP::printOutSummary` (EncGOP.cpp:3667ff; "Total Frames | Bitrate
Y-PSNR ...", then the 'a' line) and the timer line of encmain.cpp:329-337 (" Total Time: ... sec. [user] ... sec.
[elapsed]").  Blocked on sequences + trained weights in this repo (none are distributed), so the tests drive it with
synthetic logs and closed-form rate-distortion curves.

  python tools/eval_harness.py <anchor_log_dir> <test_log_dir>

BD-rate (Bjontegaard, VCEG-M33): average horizontal distance between the two log10(rate)-vs-PSNR curves over the common
PSNR interval; `method="pchip"` integrates a piecewise-cubic Hermite interpolant (what the JVET reporting sheets use),
`method="poly"` the original cubic polynomial fit through four points."""
from __future__ import annotations

import math
import os
import re
import sys

import numpy as np

_SUMMARY_HDR = re.compile(r"Total Frames\s*\|\s*Bitrate\s+Y-PSNR")
_TIME = re.compile(r"Total Time[^:]*:\s*([0-9.]+)(?:\s*\(\s*[0-9.]+\s*\))?\s*sec\.\s*\[user\]\s*([0-9.]+)(?:\s*\(\s*[0-9.]+\s*\))?\s*sec\.\s*\[elapsed\]")


def parse_vtm_log(text: str) -> dict:
    """Summary of one encoder run: first 'Total Frames | Bitrate ...' block (the all-frames line) + the timer line."""
    lines = text.splitlines()
    out = {}
    for i, ln in enumerate(lines):
        if _SUMMARY_HDR.search(ln):
            for nxt in lines[i + 1:i + 4]:
                tok = nxt.split()
                if len(tok) >= 6 and tok[0].isdigit() and tok[1] == "a":
                    out.update(frames=int(tok[0]), bitrate_kbps=float(tok[2]), psnr_y=float(tok[3]), psnr_u=float(tok[4]),
                               psnr_v=float(tok[5]), psnr_yuv=float(tok[6]) if len(tok) > 6 else float("nan"))
                    break
            if "frames" in out:
                break
    m = _TIME.search(text)
    if m:
        out.update(time_user_s=float(m.group(1)), time_elapsed_s=float(m.group(2)))
    if "bitrate_kbps" not in out or "time_user_s" not in out:
        raise ValueError("not a complete VTM encoder log (summary block or Total Time line missing)")
    out.update(parse_predictor_stats(text))
    return out


_STATS = re.compile(r"^mltcnn-stats (.*)$", re.M)


def parse_predictor_stats(text: str) -> dict:
    """The one line host/mlt_split_predictor.hpp prints at encoder shutdown under MLTCNN_STATS=1: calls and wall-clock seconds inside
    predictSplitMode / submit / wait / flush, calls per CU size.  -> {} when the log carries none (an anchor run)."""
    m = _STATS.search(text)
    if not m:
        return {}
    kv = dict(tok.split("=", 1) for tok in m.group(1).split() if "=" in tok)
    calls = {k: int(kv.get(k + "_calls", 0)) for k in ("predict", "submit", "wait", "flush")}
    secs = {k: float(kv.get(k + "_s", 0.0)) for k in ("predict", "submit", "wait", "flush")}
    return {"cnn_calls": calls["predict"] + calls["submit"], "cnn_seconds": sum(secs.values()), "cnn_calls_by_entry": calls, "cnn_seconds_by_entry": secs,
            "cnn_calls_by_size": {s: int(kv.get(f"calls_{s}", 0)) for s in (128, 64, 32, 16)}, "cnn_failed": int(kv.get("failed", 0)),
            "cnn_init_seconds": float(kv["init_s"]) if "init_s" in kv else None}   # round 6: mlt_init (weights + load-time calibration), once per encoder process


def batch_histogram(batch_log_text: str) -> dict:
    """MLTCNN_BATCH_LOG (one line per flushed anti-diagonal: 'poc P diagonal D ctus C probed K') -> {batch size: count}."""
    hist: dict = {}
    for ln in batch_log_text.splitlines():
        tok = ln.split()
        if len(tok) >= 8 and tok[0] == "poc" and tok[6] == "probed":
            hist[int(tok[7])] = hist.get(int(tok[7]), 0) + 1
    return dict(sorted(hist.items()))


def _pchip_slopes(x, y):
    """Fritsch-Carlson slopes of the monotone piecewise-cubic Hermite interpolant (same scheme as scipy's PchipInterpolator)."""
    h, d = np.diff(x), np.diff(y) / np.diff(x)
    n = len(x)
    m = np.zeros(n)
    for k in range(1, n - 1):
        if d[k - 1] * d[k] > 0:
            w1, w2 = 2 * h[k] + h[k - 1], h[k] + 2 * h[k - 1]
            m[k] = (w1 + w2) / (w1 / d[k - 1] + w2 / d[k])

    def end(h0, h1, d0, d1):
        s = ((2 * h0 + h1) * d0 - h0 * d1) / (h0 + h1)
        if s * d0 <= 0:
            return 0.0
        if d0 * d1 <= 0 and abs(s) > 3 * abs(d0):
            return 3 * d0
        return s
    if n == 2:
        m[:] = d[0]
    else:
        m[0] = end(h[0], h[1], d[0], d[1])
        m[-1] = end(h[-1], h[-2], d[-1], d[-2])
    return m


def _pchip_integral(x, y, lo, hi):
    m = _pchip_slopes(x, y)
    total = 0.0
    for k in range(len(x) - 1):
        a, b = max(lo, x[k]), min(hi, x[k + 1])
        if b <= a:
            continue
        h = x[k + 1] - x[k]

        def prim(t):  # integral of the Hermite cubic from x[k] to x[k] + t*h
            return h * (y[k] * (t - t ** 3 + t ** 4 / 2) + y[k + 1] * (t ** 3 - t ** 4 / 2)
                        + h * m[k] * (t ** 2 / 2 - 2 * t ** 3 / 3 + t ** 4 / 4) + h * m[k + 1] * (-t ** 3 / 3 + t ** 4 / 4))
        total += prim((b - x[k]) / h) - prim((a - x[k]) / h)
    return total


def bd_rate(anchor, test, method: str = "pchip") -> float:
    """anchor / test: iterables of (bitrate, psnr), >= 4 points each (VTM CTC: QP 22/27/32/37).  Returns the average
    bitrate difference in percent at equal PSNR (negative = the test encoder needs fewer bits)."""
    def prep(pts):
        p = sorted(((float(q), math.log10(float(r))) for r, q in pts))
        q, lr = np.array([v[0] for v in p]), np.array([v[1] for v in p])
        if len(q) < 4 or np.any(np.diff(q) <= 0):
            raise ValueError("need >= 4 points with distinct PSNR")
        return q, lr
    qa, ra = prep(anchor)
    qt, rt = prep(test)
    lo, hi = max(qa[0], qt[0]), min(qa[-1], qt[-1])
    if hi <= lo:
        raise ValueError("PSNR ranges do not overlap")
    if method == "poly":
        ia = np.polyint(np.polyfit(qa, ra, 3))
        it = np.polyint(np.polyfit(qt, rt, 3))
        da = np.polyval(ia, hi) - np.polyval(ia, lo)
        dt = np.polyval(it, hi) - np.polyval(it, lo)
    elif method == "pchip":
        da, dt = _pchip_integral(qa, ra, lo, hi), _pchip_integral(qt, rt, lo, hi)
    else:
        raise ValueError(method)
    return (10.0 ** ((dt - da) / (hi - lo)) - 1.0) * 100.0


def time_saving(anchor_s, test_s) -> float:
    """Mean over the rate points of (T_anchor - T_test) / T_anchor, in percent (the figure fast-encoder papers report)."""
    a, t = np.asarray(anchor_s, float), np.asarray(test_s, float)
    if a.shape != t.shape or a.size == 0:
        raise ValueError("need matching, non-empty time lists")
    return float(np.mean((a - t) / a) * 100.0)


def collect(log_dir: str) -> dict:
    """{sequence: {qp: summary}} from files named <Sequence>_q<QP>.txt (the reference scripts' naming)."""
    out: dict = {}
    for fn in sorted(os.listdir(log_dir)):
        m = re.match(r"(.+)_q(\d+)\.txt$", fn)
        if not m:
            continue
        with open(os.path.join(log_dir, fn), errors="replace") as f:
            out.setdefault(m.group(1), {})[int(m.group(2))] = parse_vtm_log(f.read())
    return out


def compare(anchor_dir: str, test_dir: str, method: str = "pchip"):
    a, t = collect(anchor_dir), collect(test_dir)
    rows = []
    for seq in sorted(set(a) & set(t)):
        qps = sorted(set(a[seq]) & set(t[seq]))
        if len(qps) < 4:
            continue
        bd = bd_rate([(a[seq][q]["bitrate_kbps"], a[seq][q]["psnr_y"]) for q in qps],
                     [(t[seq][q]["bitrate_kbps"], t[seq][q]["psnr_y"]) for q in qps], method)
        ts = time_saving([a[seq][q]["time_user_s"] for q in qps], [t[seq][q]["time_user_s"] for q in qps])
        rows.append((seq, qps, bd, ts))
    return rows


def main(argv):
    if len(argv) != 3:
        print(__doc__)
        return 2
    rows = compare(argv[1], argv[2])
    print(f"{'sequence':28s} {'QPs':16s} {'BD-rate Y [%]':>14s} {'time saving [%]':>16s}")
    for seq, qps, bd, ts in rows:
        print(f"{seq:28s} {','.join(map(str, qps)):16s} {bd:14.2f} {ts:16.2f}")
    if rows:
        print(f"{'average':28s} {'':16s} {np.mean([r[2] for r in rows]):14.2f} {np.mean([r[3] for r in rows]):16.2f}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
