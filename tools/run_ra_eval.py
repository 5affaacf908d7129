#!/usr/bin/env python3
"""End-to-end encodes under the reference's evaluation tool set (BASELINE configs[3] stand-in): the patched
reference encoder (oracle/_ref/vtm/EncoderApp = the authors' VTM-11.0 tree + patches/*.patch, tools/build_vtm.sh) on a synthetic clip,
one process per (leg, QP), several at a time.

  --mode cpu   build container, no GPU: every inference fails by fault injection (MLTCNN_FAULT_INJECT=1 -> -1 -> EncModeCtrl::setNewModeList
               is a no-op, EncModeCtrl.cpp:147-148).  Configuration: the REFERENCE's own cfg/encoder_randomaccess_vtm.cfg (GOP 32 `:15`,
               CTU 128 `:112`, MTT depth 3 `:119`, BIO / CIIP / Geo `:139-141`, LMCS `:145`, DMVR `:151`) when /root/reference is mounted.
               Checks: N1 -- bitstream(inject) == bitstream(anchor = no CU size enabled); N3 -- bitstream(MLTCNN_BATCH=1) ==
               bitstream(MLTCNN_BATCH=0) under WaveFrontSynchro=1; the batched stream decodes to the encoder's reconstruction.
               --force-splits 2,1: the N3 pair again with every prediction "succeeding" with that split mode (MLTCNN_FORCE_SPLIT, test hook):
               the decision-dependent encoder paths (setNewModeList with BT_H / QT) without a GPU -- how the global uni-MV reuse cache was found.
  --mode gpu   GPU box (the reference tree does not exist there): the same TOOL SET in tests/data/vtm_ra_tools.cfg (tools/make_ra_cfg.py),
               real decisions from seeded weights.  Per QP: anchor, serial (one mlt_predict per CU), MLTCNN_BATCH=0 and =1 under WPP
               (bit-identical), wall-clock / CPU time, time saving vs anchor, share of the encode inside the predictor (MLTCNN_STATS), batch-size
               histogram; at one QP every predictSplitMode call (128 only, and all four sizes with MLTCNN_SIZE_MASK=0xF) is dumped and
               re-checked against the CPU oracle.  BD-rate is computed by tools/eval_harness.py from the logs the encoder wrote -- with SEEDED
               (untrained) weights it measures the harness, not the method.

  --mode alone (round 6) GPU box: what the call costs the ENCODER when it has the GPU to itself.  The legs that use the GPU (serial:
               one mlt_predict per CU; MLTCNN_BATCH=0 / =1 under WPP; with >= 2 GPUs also MLTCNN_BATCH=1 over MLTCNN_DEVICES=0,1) run ONE AFTER THE
               OTHER -- round 5's table ran 24 encoders against one device and read 1-15 ms per call -- while the anchors (no CNN, no GPU) encode
               beside them on other cores.  Per leg: us per predictSplitMode call, share of the encode's elapsed time, seconds inside mlt_init
               (weights + load-time calibration: MLTCNN_STATS init_s), batch-size histogram; batched == serial bitstream.  --width 1920 --height 1080
               --frames 9 is the shape of the reference's script_128/BasketballDrive_enc_50.sh:4-19 (135 CTUs of 128 x 128 per picture).

Writes <out>/summary.json, <out>/table.md and the encoder logs (<out>/<leg>/synth_q<QP>.txt, the reference scripts' naming)."""
import argparse
import concurrent.futures as cf
import hashlib
import json
import os
import struct
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
VTM = os.path.join(ROOT, "oracle", "_ref", "vtm")
REF_CFG = "/root/reference/vtm-mlt-cpp/cfg/encoder_randomaccess_vtm.cfg"
OWN_CFG = os.path.join(ROOT, "tests", "data", "vtm_ra_tools.cfg")
ENV_KEYS = ("MLTCNN_FAULT_INJECT", "MLTCNN_FORCE_SPLIT", "MLTCNN_CALL_DUMP_FILE", "MLTCNN_SIZE_MASK", "MLTCNN_WEIGHTS_DIR", "MLTCNN_DEVICE", "MLTCNN_FLAGS", "MLTCNN_DEVICES",
            "MLTCNN_BATCH", "MLTCNN_BATCH_LOG", "MLTCNN_STATS")


def base_env():
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(ROOT, "fastintercu-vvc_amd") + ":" + env.get("LD_LIBRARY_PATH", "")
    for k in ENV_KEYS:
        env.pop(k, None)
    return env


def read_call_dump(path):
    """Records written by mlt::SplitPredictor::dumpCall (host/mlt_split_predictor.hpp, -DMLTCNN_TEST_HOOKS builds)."""
    import numpy as np
    out = []
    with open(path, "rb") as f:
        while True:
            hdr = f.read(24)
            if len(hdr) < 24:
                break
            magic, cuw, poc, qp, split, nl = struct.unpack("<6i", hdr)
            assert magic == 0x4D4C5443
            lg = np.frombuffer(f.read(15 * 4), "<f4")[:nl].copy()
            org = np.frombuffer(f.read(cuw * cuw * 2), "<i2").reshape(cuw, cuw).copy()
            pred = np.frombuffer(f.read(cuw * cuw * 2), "<i2").reshape(cuw, cuw).copy()
            out.append(dict(cuw=cuw, poc=poc, qp=qp, split=split, logits=lg, org=org, pred=pred))
    return out


def check_dump_against_oracle(dump, blobs, tol=1e-3):
    """Every dumped call vs the CPU oracle: |dlogit| <= tol, split == the oracle's argmax of the reference's head (element [2] for 128,
    [0] otherwise, EncCu.cpp:913-919) wherever the oracle's own margin exceeds 4e-5 (the library's decision guard is on)."""
    import numpy as np
    from oracle import Oracle
    calls = read_call_dump(dump)
    by_size = {}
    for c in calls:
        by_size.setdefault(c["cuw"], []).append(c)
    rep = {}
    for s, cs in sorted(by_size.items(), reverse=True):
        orc = Oracle(blobs[s])
        org = np.stack([c["org"] for c in cs]); pred = np.stack([c["pred"] for c in cs])
        poc = np.array([c["poc"] for c in cs], np.int32); qp = np.array([c["qp"] for c in cs], np.int32)
        ref, ref_split = orc.forward(org, pred, poc, qp, threads=min(os.cpu_count() or 8, 64))
        got = np.stack([c["logits"] for c in cs])
        lo = sum(orc.head_classes[:2]) if s == 128 else 0
        sl = slice(lo, lo + orc.head_classes[2 if s == 128 else 0])
        srt = np.sort(ref[:, sl].astype(np.float64), axis=1)
        decisive = (srt[:, -1] - srt[:, -2]) > 4e-5
        split = np.array([c["split"] for c in cs])
        rep[s] = {"calls": len(cs), "max_abs_dlogit": float(np.abs(got - ref).max()), "split_mismatch_decisive": int(((split != ref_split) & decisive).sum()),
                  "oracle_ties": int((~decisive).sum()), "splits": np.bincount(split, minlength=6).tolist(), "pocs": sorted({int(p) for p in poc})[:40],
                  "within_tolerance": bool(np.abs(got - ref).max() <= tol)}
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=("cpu", "gpu", "alone"), required=True)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "ra_eval"))
    ap.add_argument("--width", type=int, default=832)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--frames", type=int, default=17)
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--qps", default=None, help="comma list; default 32 (cpu) / 22,27,32,37 (gpu)")
    ap.add_argument("--dump-qp", type=int, default=32)
    ap.add_argument("--jobs", type=int, default=0)
    ap.add_argument("--cfg", default=None)
    ap.add_argument("--weight-seed", type=int, default=10)
    ap.add_argument("--force-splits", default="", help="cpu mode: comma list of split modes forced through the test hook MLTCNN_FORCE_SPLIT, one N3 pair each")
    a = ap.parse_args()
    enc, dec = os.path.join(VTM, "EncoderApp"), os.path.join(VTM, "DecoderApp")
    assert os.path.exists(enc) and os.path.exists(dec), "patched EncoderApp not built (tools/build_vtm.sh, build container)"
    cfg = a.cfg or (REF_CFG if a.mode == "cpu" and os.path.exists(REF_CFG) else OWN_CFG)
    qps = [int(q) for q in (a.qps or ("22,27,32,37" if a.mode == "gpu" else "32")).split(",")]
    jobs = a.jobs or max(1, min(24, (os.cpu_count() or 8) // (1 if a.mode == "cpu" else 4)))
    os.makedirs(a.out, exist_ok=True)
    import make_synth_yuv
    yuv = os.path.join(a.out, "synth.yuv")
    make_synth_yuv.write_yuv(yuv, make_synth_yuv.make_frames(a.width, a.height, a.frames, a.seed))

    blobs, wdir = {}, os.path.join(a.out, "torch_model")
    if a.mode in ("gpu", "alone"):
        import mltcnn_pkg
        pkg = mltcnn_pkg.load()
        os.makedirs(wdir, exist_ok=True)
        for s in (128, 64, 32, 16):
            blobs[s] = pkg.weights.synthetic_blob(pkg.synth.arch_for_size(s), a.weight_seed)
            open(os.path.join(wdir, f"MLTORPQ_splitMode_{s}.mltw"), "wb").write(blobs[s])

    legs = []   # (leg, qp, env overrides, extra args)
    wpp = ["--WaveFrontSynchro=1"]
    for q in qps:
        legs.append(("anchor", q, {"MLTCNN_SIZE_MASK": "0x100"}, []))
        if a.mode == "alone":
            import torch
            w = {"MLTCNN_WEIGHTS_DIR": wdir, "MLTCNN_STATS": "1"}
            legs.append(("wpp_anchor", q, {"MLTCNN_SIZE_MASK": "0x100", "MLTCNN_BATCH": "0"}, wpp))
            legs.append(("serial", q, dict(w), []))
            legs.append(("wpp_serial", q, dict(w, MLTCNN_BATCH="0"), wpp))
            legs.append(("wpp_batch", q, dict(w, MLTCNN_BATCH="1", MLTCNN_BATCH_LOG=os.path.join(a.out, f"batch_q{q}.log")), wpp))
            if torch.cuda.device_count() >= 2:   # one encoder process over two GPUs (mlt_config.devices: submitted CUs are dealt round-robin)
                legs.append(("wpp_batch_2gpu", q, dict(w, MLTCNN_BATCH="1", MLTCNN_DEVICES="0,1"), wpp))
        elif a.mode == "cpu":
            legs.append(("inject", q, {"MLTCNN_FAULT_INJECT": "1"}, []))
            legs.append(("wpp_serial", q, {"MLTCNN_FAULT_INJECT": "1", "MLTCNN_BATCH": "0"}, wpp))
            legs.append(("wpp_batch", q, {"MLTCNN_FAULT_INJECT": "1", "MLTCNN_BATCH": "1", "MLTCNN_BATCH_LOG": os.path.join(a.out, f"batch_q{q}.log")}, wpp))
            for fs in [v for v in a.force_splits.split(",") if v]:
                legs.append((f"wpp_serial_fs{fs}", q, {"MLTCNN_FAULT_INJECT": "1", "MLTCNN_FORCE_SPLIT": fs, "MLTCNN_BATCH": "0"}, wpp))
                legs.append((f"wpp_batch_fs{fs}", q, {"MLTCNN_FAULT_INJECT": "1", "MLTCNN_FORCE_SPLIT": fs, "MLTCNN_BATCH": "1"}, wpp))
        else:
            w = {"MLTCNN_WEIGHTS_DIR": wdir, "MLTCNN_STATS": "1"}
            legs.append(("serial", q, dict(w), []))
            legs.append(("wpp_anchor", q, {"MLTCNN_SIZE_MASK": "0x100", "MLTCNN_BATCH": "0"}, wpp))
            legs.append(("wpp_serial", q, dict(w, MLTCNN_BATCH="0"), wpp))
            legs.append(("wpp_batch", q, dict(w, MLTCNN_BATCH="1", MLTCNN_BATCH_LOG=os.path.join(a.out, f"batch_q{q}.log")), wpp))
            if q == a.dump_qp:
                legs.append(("dump128", q, dict(w, MLTCNN_CALL_DUMP_FILE=os.path.join(a.out, "calls128.bin")), []))
                legs.append(("allsizes", q, dict(w, MLTCNN_SIZE_MASK="0xF", MLTCNN_CALL_DUMP_FILE=os.path.join(a.out, "calls_all.bin")), []))
    for f in ("calls128.bin", "calls_all.bin") + tuple(f"batch_q{q}.log" for q in qps):
        if os.path.exists(os.path.join(a.out, f)):
            os.remove(os.path.join(a.out, f))

    def run(leg):
        name, q, envo, extra = leg
        d = os.path.join(a.out, name)
        os.makedirs(d, exist_ok=True)
        cmd = [enc, "-c", cfg, "-i", yuv, "-wdt", str(a.width), "-hgt", str(a.height), "-fr", "30", "-f", str(a.frames), "--InputBitDepth=10",
               "--InputChromaFormat=420", "-q", str(q), "-b", os.path.join(d, f"synth_q{q}.bin"), "-o", os.path.join(d, f"synth_q{q}_rec.yuv")] + extra
        env = base_env()
        env.update(envo)
        t0 = time.time()
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        open(os.path.join(d, f"synth_q{q}.txt"), "w").write(r.stdout)
        return name, q, r.returncode, time.time() - t0

    t_all = time.time()
    if a.mode == "alone":
        # the legs that touch the GPU one at a time (this thread); the anchors -- no CNN, no device -- beside them on the pool
        uses_gpu = lambda leg: "MLTCNN_WEIGHTS_DIR" in leg[2]
        with cf.ThreadPoolExecutor(4) as ex:
            fut = [ex.submit(run, l) for l in legs if not uses_gpu(l)]
            done = [run(l) for l in legs if uses_gpu(l)] + [f.result() for f in fut]
        jobs = 1
    else:
        with cf.ThreadPoolExecutor(jobs) as ex:
            done = list(ex.map(run, legs))
    bad = [d for d in done if d[2] != 0]
    assert not bad, bad

    import eval_harness as eh
    sha = lambda leg, q: hashlib.sha256(open(os.path.join(a.out, leg, f"synth_q{q}.bin"), "rb").read()).hexdigest()
    summ = {"mode": a.mode, "cfg": os.path.relpath(cfg, ROOT) if cfg.startswith(ROOT) else cfg, "clip": f"{a.width}x{a.height}, {a.frames} frames, 10-bit 4:2:0, tools/make_synth_yuv.py seed {a.seed}",
            "qps": qps, "jobs": jobs, "host_cpus": os.cpu_count(), "wall_s": None, "checks": {}, "rows": []}
    logs = {}
    for name, q, _, wall in done:
        s = eh.parse_vtm_log(open(os.path.join(a.out, name, f"synth_q{q}.txt"), errors="replace").read())
        s["wall_s"] = wall
        s["sha256"] = sha(name, q)
        s["hello"] = open(os.path.join(a.out, name, f"synth_q{q}.txt"), errors="replace").read().count("Hello")
        logs[(name, q)] = s
    ok = True
    for q in qps:
        c = {}
        if a.mode == "cpu":
            c["n1_inject_equals_anchor"] = logs[("inject", q)]["sha256"] == logs[("anchor", q)]["sha256"]
            c["inject_hello_count"] = logs[("inject", q)]["hello"]
        c["n3_batch_equals_serial"] = logs[("wpp_batch", q)]["sha256"] == logs[("wpp_serial", q)]["sha256"]
        if ("wpp_batch_2gpu", q) in logs:
            c["two_gpus_equal_one"] = logs[("wpp_batch_2gpu", q)]["sha256"] == logs[("wpp_batch", q)]["sha256"]
            ok &= c["two_gpus_equal_one"]
        for fs in [v for v in a.force_splits.split(",") if v and a.mode == "cpu"]:
            c[f"n3_batch_equals_serial_forced_split_{fs}"] = logs[(f"wpp_batch_fs{fs}", q)]["sha256"] == logs[(f"wpp_serial_fs{fs}", q)]["sha256"]
            c[f"forced_split_{fs}_differs_from_full_rdo"] = logs[(f"wpp_serial_fs{fs}", q)]["sha256"] != logs[("wpp_serial", q)]["sha256"]
        # the batched stream decodes to the encoder's own reconstruction (SEI picture hashes verified when the cfg writes them)
        d = os.path.join(a.out, "wpp_batch")
        r = subprocess.run([dec, "-b", os.path.join(d, f"synth_q{q}.bin"), "-o", os.path.join(d, f"synth_q{q}_dec.yuv"), "-d", "10"], env=base_env(), capture_output=True, text=True)
        c["batch_decodes_to_recon"] = r.returncode == 0 and open(os.path.join(d, f"synth_q{q}_dec.yuv"), "rb").read() == open(os.path.join(d, f"synth_q{q}_rec.yuv"), "rb").read()
        bl = os.path.join(a.out, f"batch_q{q}.log")
        c["batch_histogram"] = eh.batch_histogram(open(bl).read()) if os.path.exists(bl) else {}
        summ["checks"][q] = c
        ok &= c["n3_batch_equals_serial"] and c["batch_decodes_to_recon"] and c.get("n1_inject_equals_anchor", True)
        ok &= all(v for k, v in c.items() if k.startswith("n3_batch_equals_serial_forced_split_"))
    if a.mode == "gpu":
        for key, f in (("dump128", "calls128.bin"), ("allsizes", "calls_all.bin")):
            rep = check_dump_against_oracle(os.path.join(a.out, f), blobs)
            summ["checks"][key] = rep
            ok &= all(v["within_tolerance"] and v["split_mismatch_decisive"] == 0 for v in rep.values())
        ok &= set(summ["checks"]["allsizes"]) == {128, 64, 32, 16}          # the sub-128 call sites really ran (EncCu.cpp:754's commented-out clauses)
        ok &= set(summ["checks"]["dump128"]) == {128}
        # calibrate on what the encoder really sent: the dump of this RA encode through mlt_calibrate, appended and replacing
        import calibrate_from_dump as cfd
        summ["checks"]["calibrate_on_the_dump"] = {mode: cfd.calibrate(pkg, os.path.join(a.out, "calls128.bin"), wdir, 128, replace=(mode == "replace"))[0] for mode in ("append", "replace")}
        for v in summ["checks"]["calibrate_on_the_dump"].values():
            ok &= v["after"]["calibrated"] == 1 and v["after"]["calib_caller_cus"] >= 1
    for (name, q), s in sorted(logs.items()):
        an = logs[("wpp_anchor" if name.startswith("wpp_") and ("wpp_anchor", q) in logs else "anchor", q)]
        row = {"leg": name, "qp": q, "kbps": s["bitrate_kbps"], "psnr_y": s["psnr_y"], "enc_user_s": s["time_user_s"], "enc_elapsed_s": s["time_elapsed_s"],
               "time_saving_user_pct": round(100.0 * (an["time_user_s"] - s["time_user_s"]) / an["time_user_s"], 2),
               "time_saving_elapsed_pct": round(100.0 * (an["time_elapsed_s"] - s["time_elapsed_s"]) / an["time_elapsed_s"], 2),
               "cnn_calls": s.get("cnn_calls"), "cnn_seconds": s.get("cnn_seconds"),
               "cnn_share_of_elapsed_pct": round(100.0 * s["cnn_seconds"] / s["time_elapsed_s"], 3) if s.get("cnn_seconds") is not None else None,
               "us_per_call": round(1e6 * s["cnn_seconds"] / s["cnn_calls"], 1) if s.get("cnn_calls") else None,
               "cnn_calls_by_size": s.get("cnn_calls_by_size"), "cnn_init_seconds": s.get("cnn_init_seconds"),
               "cnn_seconds_by_entry": s.get("cnn_seconds_by_entry"), "sha256": s["sha256"][:16]}
        summ["rows"].append(row)
    if a.mode == "gpu" and len(qps) >= 4:
        bd = {}
        for test, anchor in (("serial", "anchor"), ("wpp_batch", "wpp_anchor")):
            try:
                bd[test] = {"bd_rate_y_pct": round(eh.bd_rate([(logs[(anchor, q)]["bitrate_kbps"], logs[(anchor, q)]["psnr_y"]) for q in qps],
                                                              [(logs[(test, q)]["bitrate_kbps"], logs[(test, q)]["psnr_y"]) for q in qps]), 3),
                            "time_saving_user_pct": round(eh.time_saving([logs[(anchor, q)]["time_user_s"] for q in qps], [logs[(test, q)]["time_user_s"] for q in qps]), 2),
                            "time_saving_elapsed_pct": round(eh.time_saving([logs[(anchor, q)]["time_elapsed_s"] for q in qps], [logs[(test, q)]["time_elapsed_s"] for q in qps]), 2)}
            except ValueError as e:
                bd[test] = {"error": str(e)}
        summ["bd"] = bd
    summ["wall_s"] = round(time.time() - t_all, 1)
    summ["ok"] = bool(ok)
    json.dump(summ, open(os.path.join(a.out, "summary.json"), "w"), indent=1, default=str)
    with open(os.path.join(a.out, "table.md"), "w") as f:
        f.write(f"RA-toolset encodes ({summ['cfg']}; {summ['clip']}; mode {a.mode}; {jobs} GPU-using encoder process{'es' if jobs > 1 else ''} at a time on {os.cpu_count()} CPUs)\n\n")
        f.write("| leg | QP | kbps | Y-PSNR | enc user s | enc elapsed s | saving user % | saving elapsed % | CNN calls | s inside predictor | share of elapsed % | us / call | s in mlt_init |\n|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
        for r in summ["rows"]:
            f.write(f"| {r['leg']} | {r['qp']} | {r['kbps']:.2f} | {r['psnr_y']:.3f} | {r['enc_user_s']:.1f} | {r['enc_elapsed_s']:.1f} | {r['time_saving_user_pct']} | {r['time_saving_elapsed_pct']} | "
                    f"{r['cnn_calls']} | {r['cnn_seconds']} | {r['cnn_share_of_elapsed_pct']} | {r['us_per_call']} | {r['cnn_init_seconds']} |\n")
        f.write("\nchecks: " + json.dumps(summ["checks"], default=str) + "\n")
        if "bd" in summ:
            f.write("\nBD-rate / time saving over the QPs (tools/eval_harness.py; SEEDED weights -- the harness is what is measured, not the method): " + json.dumps(summ["bd"]) + "\n")
    print(open(os.path.join(a.out, "table.md")).read())
    for leg in {l[0] for l in legs}:   # keep logs + bitstreams, drop the bulky YUVs
        for fn in os.listdir(os.path.join(a.out, leg)):
            if fn.endswith(".yuv"):
                os.remove(os.path.join(a.out, leg, fn))
    os.remove(yuv)
    for f in ("calls128.bin", "calls_all.bin"):
        if os.path.exists(os.path.join(a.out, f)) and os.path.getsize(os.path.join(a.out, f)) > (8 << 20):
            os.remove(os.path.join(a.out, f))
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
