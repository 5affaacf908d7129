#!/usr/bin/env python3
"""CPU baseline per SURVEY.md §8(d) -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (same import rules as oracle.py).

"The reference's CPU libtorch path" = the same graph on host cores (EncCu.cpp:894-909 with at::kCPU).  The reference
modules cannot travel to the GPU box, so this times oracle/torch_port.py -- the restatement on PyTorch's CPU operators
(oneDNN convolutions), i.e. the kernels LibTorch-CPU would dispatch -- the way §8(d) prescribes:
  * batch 1 (the encoder's call pattern) and batch 64 in ONE process with torch.set_num_threads(physical cores),
  * the batch-4096 workload sharded over P worker processes x T threads with P*T = physical cores (one process with
    256 threads oversubscribes oneDNN's per-primitive parallelism: measured 3.6 CU/s in round 1), on a bounded sample,
  * 3 warm-ups, median of >= 10 timed runs; cores, CPU model, P and T reported.
Prints one JSON object (last line of stdout) when --json is given.  Never touches the GPU.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def cpu_info():
    model, pairs = "unknown", set()
    phys = core = None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and model == "unknown":
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        pairs.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    logical = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    physical = min(len(pairs), logical) if pairs else max(logical // 2, 1)
    return model, physical, logical


def _worker(rank, T, size, first, count, runs, warm, bar, blob):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import mltcnn_pkg
    from oracle.torch_port import TorchPort
    pkg = mltcnn_pkg.load()
    torch.set_num_threads(T)
    port = TorchPort(blob)
    org, pred = pkg.synth.make_patches_bulk(size, count, 0xC0FFEE, first=first)
    poc, qp = pkg.synth.make_scalars(count, 0xC0FFEE, first=first)
    for i in range(warm + runs):
        bar.wait()
        port.forward(org, pred, poc, qp, chunk=64)
        bar.wait()


def sharded(size, blob, P, T, per_proc, runs, warm):
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    bar = ctx.Barrier(P + 1)
    procs = [ctx.Process(target=_worker, args=(r, T, size, r * per_proc, per_proc, runs, warm, bar, blob)) for r in range(P)]
    for p in procs:
        p.start()
    times = []
    for i in range(warm + runs):
        bar.wait(timeout=600)
        t0 = time.perf_counter()
        bar.wait(timeout=600)
        if i >= warm:
            times.append(time.perf_counter() - t0)
    for p in procs:
        p.join(timeout=60)
    return times


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=128)
    ap.add_argument("--threads", type=int, default=0, help="T: threads per worker process of the sharded run (default 1: measured on the "
                    "2 x 64-core EPYC 9575F of the GPU box, T = 1 / 2 / 4 / 8 / 16 -> 790 / 691 / 464 / 370 / 221 CU/s)")
    ap.add_argument("--procs", type=int, default=0, help="P: worker processes (default physical cores / T)")
    ap.add_argument("--per-proc", type=int, default=8, help="CUs each worker evaluates per run (sample = P * this)")
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--json", action="store_true")
    args = ap.parse_args()
    model, physical, logical = cpu_info()
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import mltcnn_pkg
    from oracle.torch_port import TorchPort
    pkg = mltcnn_pkg.load()
    size = args.size
    blob = pkg.weights.synthetic_blob(pkg.synth.arch_for_size(size), 10)
    rows = []
    # ---- one process: batch 1 (the encoder's call pattern, EncCu.cpp:894-909) and batch 64.  The thread count is SWEPT: one
    # 128x128 CU on all 128 cores of the GPU box is an oversubscription lottery (round 2: 5 CU/s in the driver's run, 160 in ours);
    # every T is reported and the best one is the row that counts ----
    port = TorchPort(blob)
    org, pred = pkg.synth.make_patches_bulk(size, 64, 0xC0FFEE)
    poc, qp = pkg.synth.make_scalars(64, 0xC0FFEE)
    sweep_T = sorted({t for t in (8, 16, 32, 64, 128) if t <= physical} | {physical})
    for b in (1, 64):
        best_row, sweep = None, {}
        for T in sweep_T:
            torch.set_num_threads(T)
            ts = []
            for i in range(args.warmup + (args.runs if b == 1 else max(args.runs // 2, 3))):
                t0 = time.perf_counter()
                port.forward(org[:b], pred[:b], poc[:b], qp[:b], chunk=64)
                if i >= args.warmup:
                    ts.append(time.perf_counter() - t0)
            med = statistics.median(ts)
            sweep[str(T)] = round(b / med, 2)
            row = {"impl": "torch-CPU port (oracle/torch_port.py, oneDNN fp32)", "batch": b, "procs": 1, "threads": T,
                   "value": round(b / med, 2), "median_s": round(med, 5), "runs": len(ts), "warmup": args.warmup}
            if best_row is None or row["value"] > best_row["value"]:
                best_row = row
        best_row["thread_sweep_cu_per_s"] = sweep
        rows.append(best_row)
    # ---- the batch workload sharded over P processes x T threads ----
    T = args.threads or 1
    P = args.procs or max(physical // T, 1)
    times = sharded(size, blob, P, T, args.per_proc, args.runs, args.warmup)
    med = statistics.median(times)
    n = P * args.per_proc
    rows.append({"impl": "torch-CPU port (oracle/torch_port.py, oneDNN fp32)", "batch": f"{n} (sample of the 4096-CU batch, {args.per_proc} per process)",
                 "procs": P, "threads": T, "value": round(n / med, 2), "median_s": round(med, 5), "runs": args.runs, "warmup": args.warmup})
    best = max(rows, key=lambda r: r["value"])
    # ---- optional second row of BASELINE.md section 3: the reference's CALL PATTERN -- the model file is re-read and re-materialised on every
    # call (EncCu.cpp:894-900: torch::jit::load + .eval() per CU) -- one CU per call, best thread count of the batch-1 sweep.  Here: the MLTW
    # blob re-read from a file and its tensors rebuilt per call (no TorchScript graph to parse: a LOWER bound of what the reference pays).
    # Reported, never the headline denominator (`value` above is chosen before this row exists).
    try:
        import tempfile
        T1 = rows[0]["threads"]
        torch.set_num_threads(T1)
        with tempfile.NamedTemporaryFile(suffix=".mltw") as tf:
            tf.write(blob)
            tf.flush()
            ts = []
            for i in range(2 + 8):
                t0 = time.perf_counter()
                with open(tf.name, "rb") as fh:
                    TorchPort(fh.read()).forward(org[:1], pred[:1], poc[:1], qp[:1], chunk=64)
                if i >= 2:
                    ts.append(time.perf_counter() - t0)
        med1 = statistics.median(ts)
        rows.append({"impl": "torch-CPU port, weights re-read and rebuilt on EVERY call (the reference's call pattern, EncCu.cpp:894-900; not a candidate for `value`)",
                     "batch": 1, "procs": 1, "threads": T1, "value": round(1.0 / med1, 2), "median_s": round(med1, 5), "runs": len(ts), "warmup": 2})
    except Exception as e:  # the optional row must never cost the baseline
        rows.append({"impl": "reload-per-call row failed", "error": str(e)[:200], "value": 0.0, "batch": 1, "procs": 1, "threads": 0})
    out = {"value": best["value"], "unit": "CU-inferences/s", "cores": best["procs"] * best["threads"], "kind": "port",
           "sample": f"{best['batch']} CUs per run, {best['procs']} processes x {best['threads']} threads, median of {args.runs} after {args.warmup} warm-ups",
           "cpu_model": model, "physical_cores": physical, "logical_cpus": logical, "rows": rows}
    if args.json:
        print(json.dumps(out))
    else:
        print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
