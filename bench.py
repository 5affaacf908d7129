#!/usr/bin/env python3
"""bench.py -- CU-inferences/s of the MI355X MLT-CNN split predictor (BASELINE.json metric).

One "step" = one pass of the hot path (raw int16 org/pred planes + poc/qp already resident in HBM
-> logits + split modes in HBM) over ONE batch of 4096 synthetic 128x128 CUs per GPU.
N > 1: one process per GPU (torch.distributed.run), weights broadcast once over RCCL, the batch is
sharded by rank with no hot-path collective (weak scaling: 4096 CUs per GPU).  `python bench.py --gpus N`
starts the N ranks itself (as a child process, before anything touches the GPU); under an existing
torchrun launch (WORLD_SIZE set) it is one of the ranks.

Prints ONE JSON line on rank 0.  See DESIGN.md "Measurement" for how roofline / cpu_baseline are defined.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 4096
SIZE = 128
# SURVEY.md §8(d), per CU: 2 x MACs (conv + FC) / fp16 layer-wise bytes / compulsory bytes
FLOP_PER_CU = {128: 1_134_562_340, 64: 224_007_036, 32: 56_005_500, 16: 17_150_844}
LAYERWISE_BYTES_PER_CU = {128: 8_061_854, 64: 1_967_214, 32: 492_654, 16: 127_086}
COMPULSORY_BYTES_PER_CU = {128: 65_580, 64: 16_452, 32: 4_164, 16: 1_092}
MFMA_PEAK_TFLOPS = 2500.0            # dense fp16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md (6.29 TB/s measured float4 copy)
LOGIT_TOL = 1e-3                     # BASELINE.json north_star


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 20 + 50 steps of ~5 ms.  The first ~15 steps of a process run 1-3 % slower (measured round 3, scripts/steps_ab.py: 10 steps
    # after 3 warm-ups 808-836 k CU/s, after 20 warm-ups 834 k, 50 after 5: 829-863 k on the same boxes), so a 3 + 10 default under-reported
    # the steady state the metric is about
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=BATCH, help="CUs per GPU per step (BASELINE: 4096)")
    ap.add_argument("--size", type=int, default=SIZE)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU legs (C oracle over the whole batch = full-batch parity, torch port)")
    ap.add_argument("--cpu-sample", type=int, default=0, help="CUs of the batch the C oracle evaluates (default: whole batch when the host has >= 64 cores)")
    ap.add_argument("--weight-seed", type=int, default=10, help="seed of the synthetic weight set (10 = the BASELINE workload; 13 / 24 land in the hi+lo-weights tier)")
    ap.add_argument("--weights-blob", default="", help="an MLTW file instead of the seeded set (e.g. tools/train_synth_weights.py's trained family): another WORKLOAD, never the driver's line")
    ap.add_argument("--flags", type=lambda s: int(s, 0), default=0, help="mlt_config.flags; 0 = the shipped configuration (calibrated arithmetic + flat-content guard + decision guard, what the encoder runs); "
                         "1 = exact arithmetic for 128, 2 = fast arithmetic for 64/32/16, 0x20 = without the decision guard (measurement only)")
    ap.add_argument("--sustain-s", type=float, default=None,
                    help="seconds of back-to-back steps AFTER the timed region for derived.sustained_cu_per_s (the 50-step region lasts 0.25 s on a cool chip; "
                         "the part is power-limited) and of fp16 GEMMs for derived.mfma_sustained (the box's own MFMA ceiling); 0 = skip both")
    ap.add_argument("--flat-frac", type=float, default=0.0,
                    help="fraction of the batch replaced by content the flat-content guard re-evaluates exactly (constant / dither / ramp / low contrast in turn); 0 = the BASELINE workload")
    ap.add_argument("--content", choices=("texture", "natural"), default="texture",
                    help="texture: the BASELINE workload (SURVEY 8d); natural: 1/f-spectrum scenes + motion-shifted prediction (synth.natural_patches) -- the class the guard's flag rate is quoted on")
    ap.add_argument("--latency", action="store_true", help="also time the synchronous one-CU-per-call path (mlt_predict)")
    ap.add_argument("--host-staged", action="store_true",
                    help="also time mlt_predict_batch from pinned HOST buffers (PCIe-inclusive rate; never `value`)")
    a = ap.parse_args()
    if a.sustain_s is None:
        # the sustained legs (16 s of full-power work per rank) belong to the HEADLINE run: the default workload in the shipped configuration;
        # sweeps, A/Bs and the other workloads skip them unless asked
        headline = (not a.weights_blob and a.weight_seed == 10 and a.flags == 0 and a.content == "texture" and a.flat_frac == 0 and a.size == 128
                    and not a.no_cpu_baseline)
        a.sustain_s = 8.0 if headline else 0.0
    return a


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(args):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks as a CHILD process -- this process has not
    touched the GPU (no torch.cuda / HIP call yet) and never exec()s -- relay rank 0's JSON line, exit with the child's rc."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in child.stdout.splitlines() if l.startswith("{")]
    for l in child.stdout.splitlines():
        if not l.startswith("{"):
            print(l, file=sys.stderr)
    if child.returncode == 0 and len(lines) == 1:
        print(lines[0])
        return 0
    print(f"bench.py: {args.gpus}-rank launch failed (rc {child.returncode}, {len(lines)} JSON lines)", file=sys.stderr)
    return child.returncode or 1


def source_signature():
    """sha256 over the kernel + runtime sources (every file of csrc/, sorted): profiles/pmc_traffic.json is only valid for the build it was
    measured on.  One definition, in the package's build recipe -- the library carries the same value (mlt_build_signature)."""
    import mltcnn_pkg
    return mltcnn_pkg.load().build.source_signature()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))

    import numpy as np
    import torch
    import mltcnn_pkg
    pkg = mltcnn_pkg.load()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    dist = None
    rccl = None
    # under a torchrun launch (RANK / WORLD_SIZE in the environment) the process group is initialised even for ONE rank, so that a
    # 1-GPU box exercises the RCCL path end to end (communicator set-up + weight-blob broadcast on device tensors)
    if world > 1 or ("RANK" in os.environ and "WORLD_SIZE" in os.environ):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # one rank per GPU over RCCL.  Only for exercising this code path on a box with fewer GPUs than ranks
        # (MLT_BENCH_OVERSUBSCRIBE=1): ranks share GPUs and the init-time collectives go over gloo on host tensors.
        ndev = torch.cuda.device_count()
        oversub = world > ndev and os.environ.get("MLT_BENCH_OVERSUBSCRIBE") == "1"
        if world > ndev and not oversub:
            print(f"bench.py: {world} ranks but {ndev} GPUs (set MLT_BENCH_OVERSUBSCRIBE=1 to share GPUs over gloo)", file=sys.stderr)
            sys.exit(2)
        torch.cuda.set_device(local_rank % max(ndev, 1))
        dist.init_process_group(backend="gloo" if oversub else "nccl", rank=rank, world_size=world)
        if dist.get_backend() != "nccl" and not oversub:
            print(f"bench.py: backend {dist.get_backend()} is not RCCL", file=sys.stderr)
            sys.exit(2)
    else:
        oversub = False
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    dev_index = local_rank % torch.cuda.device_count() if oversub else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    cdev = torch.device("cpu") if oversub else dev  # where collective payloads live
    if rank == 0:
        pkg.build.build_lib()  # no-op when the in-tree library is current; never let N ranks race hipcc on one output file
    if dist is not None:
        dist.barrier()

    size, B = args.size, args.batch
    arch = pkg.synth.arch_for_size(size)
    # ---- weights: rank 0 builds the blob, everyone else receives it over RCCL (xGMI) ----
    blob = (open(args.weights_blob, "rb").read() if args.weights_blob else pkg.weights.synthetic_blob(arch, args.weight_seed)) if rank == 0 else None
    if dist is not None:
        pkg.shard.broadcast_blob(blob, dist, cdev)  # first collective also sets up the communicator: time the second
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        blob = pkg.shard.broadcast_blob(blob, dist, cdev)
        torch.cuda.synchronize()
        bcast_ms = (time.perf_counter() - t0) * 1e3
        devs = [None] * world
        dist.all_gather_object(devs, {"rank": rank, "device": dev_index, "name": torch.cuda.get_device_name(dev_index)})
        rccl = {"backend": dist.get_backend(), "world": dist.get_world_size(), "devices": devs,
                "weight_blob_bytes": len(blob), "weight_broadcast_ms": round(bcast_ms, 3)}
    m = pkg.MltCnn(device=dev_index, sizes=(size,), blobs={size: blob}, max_batch=B, flags=args.flags)
    arith = m.arithmetic(size)  # fast or exact (load-time calibration), guards
    if dist is not None:
        # every rank calibrated its own copy of the weights: they must have landed on the SAME arithmetic (rank 0's is broadcast and compared;
        # a mismatch raises on every rank) -- an N-rank line never mixes tiers
        try:
            pkg.shard.agree_on_arithmetic(arith, dist)
        except RuntimeError as e:
            print(f"bench.py: {e}", file=sys.stderr)
            dist.destroy_process_group()
            sys.exit(4)
        if rccl is not None:
            rccl["arithmetic_agreed"] = True

    # ---- synthetic inputs: rank r owns CUs [r*B, (r+1)*B) of the global batch ----
    if args.flat_frac > 0 or args.content != "texture":
        org, pred, _ = pkg.synth.make_mix_bulk(size, B, 0xC0FFEE, args.flat_frac, args.content == "natural", first=rank * B)
    else:
        org, pred = pkg.synth.make_patches_bulk(size, B, 0xC0FFEE, first=rank * B)
    poc, qp = pkg.synth.make_scalars(B, 0xC0FFEE, first=rank * B)
    d_org = torch.from_numpy(org).to(dev)
    d_pred = torch.from_numpy(pred).to(dev)
    d_poc = torch.from_numpy(poc).to(dev)
    d_qp = torch.from_numpy(qp).to(dev)
    nl = m.num_logits(size)
    d_split = torch.full((B,), -1, dtype=torch.int32, device=dev)
    d_logits = torch.zeros((B, nl), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    m.set_stream(stream.cuda_stream)

    def step():
        m.predict_batch_device(B, size, d_org.data_ptr(), d_pred.data_ptr(), d_poc.data_ptr(), d_qp.data_ptr(),
                               d_split.data_ptr(), d_logits.data_ptr())

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    reruns_before = m.arithmetic(size)["guard_reruns"]
    if dist is not None:
        dist.barrier()
    # ---- the timed region: exactly `steps` passes, nothing else (no per-launch events) ----
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    reruns_per_step = (m.arithmetic(size)["guard_reruns"] - reruns_before) / max(args.steps, 1)   # CUs the guards re-evaluated exactly, per timed step
    per_rank = None
    if dist is not None:
        own = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        allt = [torch.zeros_like(own) for _ in range(world)]
        dist.all_gather(allt, own)  # every rank's own time: a straggler GPU is visible in the line
        te = own.clone()
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())
        rates = [B * args.steps / float(x.item()) for x in allt]
        per_rank = {"cu_per_s": [round(r, 1) for r in rates], "min": round(min(rates), 1), "max": round(max(rates), 1)}
    # ---- sustained figures (outside the timed region, every rank: the power envelope is per package) ----
    # (i) the same step back to back for --sustain-s seconds; (ii) the MFMA rate THIS box sustains over the same time on a plain fp16 GEMM
    # (hipBLASLt through torch.matmul, 8192^3, operands with post-ReLU-like statistics: |N(0,1)| activations, N(0, 1/K) weights): the part
    # runs into its power limit long before the 2.5 PFLOP/s nominal peak, and roofline.frac_of_sustained is quoted against this ceiling
    sustained = None
    if args.sustain_s > 0:
        n_sus = max(args.steps, int(args.sustain_s / max(elapsed / max(args.steps, 1), 1e-4)))
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for _ in range(n_sus):
            step()
        torch.cuda.synchronize()
        sus_s = time.perf_counter() - c0
        gm = 8192
        ga = torch.randn((gm, gm), device=dev, dtype=torch.float32).abs_().to(torch.float16)
        gb = (torch.randn((gm, gm), device=dev, dtype=torch.float32) / gm ** 0.5).to(torch.float16)
        gc = torch.empty((gm, gm), device=dev, dtype=torch.float16)
        for _ in range(5):
            torch.matmul(ga, gb, out=gc)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        n_gemm = 0
        while time.perf_counter() - c0 < args.sustain_s:
            for _ in range(50):
                torch.matmul(ga, gb, out=gc)
            torch.cuda.synchronize()
            n_gemm += 50
        gemm_s = time.perf_counter() - c0
        sustained = {"cu_per_s": B * n_sus / sus_s, "steps": n_sus, "seconds": sus_s,
                     "mfma_tflops": 2.0 * gm ** 3 * n_gemm / gemm_s / 1e12, "gemm_seconds": gemm_s, "gemm": f"{gm}^3 fp16 torch.matmul (hipBLASLt), fp32 accumulate"}
        del ga, gb, gc
    # ---- the same steps again with HIP events around every kernel launch (on the launch stream): per-kernel table ----
    m.profile_enable(True)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    prof = m.profile_read()
    m.profile_enable(False)
    got_logits = d_logits.cpu().numpy()
    got_split = d_split.cpu().numpy()
    if dist is not None:
        dist.barrier()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- encoder-realistic call: ONE CU per synchronous mlt_predict (strided host planes in, split mode out), as
    # EncCu.cpp:806-921 is used today; outside the timed region, informational ----
    batch1_us = None
    if args.latency:  # opt-in so that the default command and its rocprofv3 profile contain only batch launches
        m.set_stream(0)
        lat = []
        for i in range(60):
            c0 = time.perf_counter()
            m.predict(org[i % B], pred[i % B], int(poc[i % B]), int(qp[i % B]))
            lat.append(time.perf_counter() - c0)
        batch1_us = float(np.median(lat[10:]) * 1e6)

    # ---- PCIe-inclusive rate: the same batch from pinned host memory through mlt_predict_batch (H2D of both planes,
    # kernels, D2H of split + logits), informational (SURVEY.md §8d "device-resident vs staged") ----
    staged = None
    if args.host_staged:
        ho = torch.from_numpy(org).pin_memory().numpy()
        hp = torch.from_numpy(pred).pin_memory().numpy()
        m.predict_batch(ho, hp, poc, qp)
        ts = []
        for _ in range(5):
            c0 = time.perf_counter()
            m.predict_batch(ho, hp, poc, qp)
            ts.append(time.perf_counter() - c0)
        staged = B / float(np.median(ts))

    ms_per_step = elapsed / args.steps * 1e3
    value = world * B * args.steps / elapsed
    # ---- per-kernel table; roofline of the dominant kernel = the LONGEST launch of a step (largest average launch duration) ----
    kernels = []
    tot_ms = sum(r["total_ms"] for r in prof) or 1.0
    tot_flops = sum(r["flops"] for r in prof) or 1.0
    ridge = MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
    t_bound_ms = 0.0
    for r in prof:
        avg_ms = r["total_ms"] / max(r["launches"], 1)
        fl, by = r["flops"] / max(r["launches"], 1), r["bytes"] / max(r["launches"], 1)
        kb_ms = max(fl / (MFMA_PEAK_TFLOPS * 1e12), by / (HBM_PEAK_GBS * 1e9)) * 1e3  # this launch at its binding roof
        t_bound_ms += kb_ms * r["launches"] / args.steps
        kernels.append({"name": r["name"], "launches": r["launches"], "avg_ms": round(avg_ms, 4),
                        "share": round(r["total_ms"] / tot_ms, 4), "flop_share": round(r["flops"] / tot_flops, 4),
                        "tflops": round(r["flops"] / max(r["total_ms"], 1e-9) / 1e9, 1),
                        "algo_gbs": round(r["bytes"] / max(r["total_ms"], 1e-9) / 1e6, 1),
                        "bound": "mfma" if by > 0 and fl / by >= ridge else "hbm",
                        "roof_frac": round(kb_ms / max(avg_ms, 1e-9), 4)})
    # HBM traffic per launch from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (scripts/make_traffic_json.py); only valid
    # for the sources it was measured on: otherwise null.  Every kernel row carries it next to its algorithmic bytes.
    tj, traffic_valid, traffic_meta = {}, False, {}
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        traffic_meta = tj.get("_meta", {})
        traffic_valid = traffic_meta.get("source_sig") == source_signature()
    for k, r in zip(kernels, prof):
        t = tj.get(f"{r['name']}@{B}") if traffic_valid else None
        by = r["bytes"] / max(r["launches"], 1)
        k["algo_bytes_per_launch"] = by
        k["traffic"] = t["hbm_bytes_per_launch"] if t else None
        k["traffic_over_algorithmic"] = round(t["hbm_bytes_per_launch"] / by, 3) if t and by > 0 else None
    roofline = None
    # (deterministic, ADVICE r5: the launch with the largest algorithmic FLOPs; launches within 1 % of that -- layer2 and layer3 are whole stages
    # of identical FLOPs -- are ranked by their average duration, the longer one first, then by name; every MFMA-bound launch is listed beside it)
    top_flops = max((r["flops"] / max(r["launches"], 1) for r in prof), default=0.0)
    cands = [r for r in prof if r["flops"] / max(r["launches"], 1) >= 0.99 * top_flops] if top_flops > 0 else list(prof)
    dom = max(cands, key=lambda r: (round(r["total_ms"] / max(r["launches"], 1), 2), r["name"])) if cands else None
    if dom:
        avg_ms = dom["total_ms"] / dom["launches"]
        flops_l, bytes_l = dom["flops"] / dom["launches"], dom["bytes"] / dom["launches"]
        mfma_bound = bytes_l > 0 and flops_l / bytes_l >= ridge
        traffic, traffic_src = None, None
        if tj:
            t = tj.get(f"{dom['name']}@{B}")
            if t and traffic_valid:
                traffic = t["hbm_bytes_per_launch"]
                traffic_src = f"profiles/pmc_traffic.json ({traffic_meta.get('tag')}, sources {traffic_meta.get('source_sig')})"
            else:
                traffic_src = "null: profiles/pmc_traffic.json was measured on different kernel sources" if t else "null: kernel not in profiles/pmc_traffic.json"
        if mfma_bound:
            ach = flops_l / (avg_ms * 1e-3) / 1e12
            roofline = {"kernel": dom["name"], "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4),
                        "frac_of_sustained": round(ach / sustained["mfma_tflops"], 4) if sustained else None,
                        "sustained_peak": round(sustained["mfma_tflops"], 1) if sustained else None}
        else:
            ach = bytes_l / (avg_ms * 1e-3) / 1e9
            roofline = {"kernel": dom["name"], "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
        roofline.update({"traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": round(avg_ms, 4), "launches": dom["launches"],
                         "algo_flops_per_launch": flops_l, "algo_bytes_per_launch": bytes_l,
                         "flop_per_byte": round(flops_l / max(bytes_l, 1.0), 1),
                         "selection": "the launch with the most algorithmic FLOPs (among launches within 1 % of it: the longest)",
                         "others": [{"kernel": k["name"], "avg_ms": k["avg_ms"], "bound": k["bound"], "roof_frac": k["roof_frac"]} for k in kernels if k["name"] != dom["name"] and k["flop_share"] > 0.05]})

    # ---- CPU legs (rank 0, N = 1 only): the C oracle over the batch = parity of EVERY CU of the timed workload and the
    # second baseline row; the torch-CPU port timed per SURVEY.md §8(d) = cpu_baseline.value ----
    parity = None
    cpu_baseline = None
    import oracle
    cores = os.cpu_count() or 1
    if world == 1 and not args.no_cpu_baseline:
        sample = args.cpu_sample or (B if cores >= 64 else min(B, 16 * cores))
        orc = oracle.Oracle(blob)
        orc.forward(org[:cores], pred[:cores], poc[:cores], qp[:cores], threads=cores)  # warm-up
        c0 = time.perf_counter()
        ref, ref_split = orc.forward(org[:sample], pred[:sample], poc[:sample], qp[:sample], threads=cores)
        cs = time.perf_counter() - c0
        c_row = {"impl": "C oracle (oracle/mlt_oracle.c), fp32, OpenMP over CUs", "value": round(sample / cs, 2), "threads": cores,
                 "sample": f"first {sample} CUs of the same batch, one pass, {cs:.1f} s"}
        cpu_baseline = {"value": c_row["value"], "unit": "CU-inferences/s", "cores": cores, "kind": "port", "sample": c_row["sample"],
                        "rows": [c_row]}
        try:
            tb = subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "cpu_baseline.py"), "--size", str(size), "--json"],
                                capture_output=True, text=True, timeout=600)
            tl = [l for l in tb.stdout.splitlines() if l.startswith("{")]
            if tb.returncode == 0 and tl:
                cj = json.loads(tl[-1])
                cpu_baseline["rows"] = cj["rows"] + [c_row]
                cpu_baseline.update({"cpu_model": cj["cpu_model"], "physical_cores": cj["physical_cores"], "logical_cpus": cj["logical_cpus"]})
                if cj["value"] >= c_row["value"]:  # `value` = the best CPU row, C oracle included
                    cpu_baseline.update({"value": cj["value"], "cores": cj["cores"], "sample": cj["sample"]})
            else:
                cpu_baseline["torch_port_error"] = (tb.stderr or tb.stdout)[-400:]
        except subprocess.TimeoutExpired:
            cpu_baseline["torch_port_error"] = "timeout"
    else:
        sample = min(B, args.cpu_sample or (8 if args.no_cpu_baseline else 64))  # (--cpu-sample N: parity over N CUs even without the CPU baseline legs)
        ref, ref_split = oracle.Oracle(blob).forward(org[:sample], pred[:sample], poc[:sample], qp[:sample], threads=min(cores, sample))
    hs, lo = [], 0
    for c in pkg.synth.HEAD_CLASSES[arch]:
        hs.append(slice(lo, lo + c))
        lo += c
    dec = hs[2 if size == 128 else 0]
    srt = np.sort(ref[:, dec].astype(np.float64), axis=1)
    # the configuration guarantees the argmax above this reference margin: 2 x tolerance for an arithmetic with errors up to the tolerance,
    # 4e-5 (twice the exact arithmetic's own noise) when the decision guard re-evaluates every narrower CU exactly, or the arithmetic is exact
    guarded = bool(arith["decision_guard"]) or int(arith["exact"]) == 1
    decisive = (srt[:, -1] - srt[:, -2]) > (4e-5 if guarded else 2 * LOGIT_TOL)
    mism = got_split[:sample] != ref_split
    parity = {"checked_cus": int(sample), "of_batch": int(B), "max_abs_dlogit": float(np.abs(got_logits[:sample] - ref).max()),
              "tolerance": LOGIT_TOL, "within_tolerance": bool(np.abs(got_logits[:sample] - ref).max() <= LOGIT_TOL),
              "split_mismatch_decisive": int((mism & decisive).sum()), "non_decisive": int((~decisive).sum()),
              "split_mismatch_non_decisive": int((mism & ~decisive).sum()),
              "split_identical_decisive": bool(not (mism & decisive).any()), "split_identical": bool(not mism.any()),
              "decisive_margin": 4e-5 if guarded else 2 * LOGIT_TOL,
              "oracle": "oracle/mlt_oracle.c (fp32 restatement pinned to the reference fixtures)"}

    tier = int(arith["exact"])  # 0 fast, 1 exact, 2 hi+lo weights on fp16 activations in every stage, 3 in the stages of w2_stages only, 4 exact stages, 5 exact-lite
    exact = tier == 1
    arith = m.arithmetic(size)
    def _unit(i):  # a stage with hi+lo weights in ONE of its two launch units is named by that unit
        u = (int(arith["w2_units"]) >> (2 * i)) & 3
        return f"layer{i}" + ("" if u == 3 else (".0" if u == 1 else ".1") if i == 0 else (".s2" if u == 1 else ".s1"))
    stages = "+".join(_unit(i) for i in range(5) if (int(arith["w2_stages"]) >> i) & 1)
    def _xunit(i):
        u = (int(arith["x_units"]) >> (2 * i)) & 3
        return f"layer{i}" + ("" if u == 3 else (".0" if u == 1 else ".1") if i == 0 else (".s2" if u == 1 else ".s1"))
    xstages = "+".join(_xunit(i) for i in range(5) if (int(arith["x_stages"]) >> i) & 1)
    out = {
        "metric": f"CU-inferences/sec (batch {B}, {size}x{size})", "value": round(value, 1), "unit": "CU-inferences/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16x2 (hi+lo pairs, fp32 accumulate)" if exact else "f16x2 pairs, cross terms in scaled fp8 (exact-lite, fp32 accumulate)" if tier == 5 else "f16 weights hi+lo x f16 activations (fp32 accumulate)" if tier == 2 else f"f16, weights hi+lo in {stages} (fp32 accumulate)" if tier == 3 else f"f16x2 pairs in {xstages}, f16 with weights hi+lo in {stages or 'no stage'} (fp32 accumulate)" if tier == 4 else "f16 (fp32 accumulate)", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[{1 if size == 128 else 2}]: batch {B} synthetic {size}x{size} CU patches per GPU, fp16 MFMA / fp32 accumulate, "
                               "inputs (int16 org+pred, int32 poc/qp) resident in HBM, outputs logits+split in HBM",
                   "batch_per_gpu": B, "cu_size": size, "weights": (f"MLTW file {os.path.basename(args.weights_blob)}" if args.weights_blob else f"synthetic seed {args.weight_seed}") + " (no trained checkpoint is distributed)",
                   "parallelism": f"shard{world}",
                   "arithmetic": {"mode": "exact (fp16 hi+lo pairs, 3 MFMA passes)" if exact else "exact-lite (fp16 hi x hi + both cross terms in one scaled FP8 MFMA: 2 fp16-equivalent passes)" + (" + decision guard" if arith["decision_guard"] else "") if tier == 5 else ("hi+lo weights (2 MFMA passes)" if tier == 2 else f"hi+lo weights in {stages}, single pass in the other stages" if tier == 3 else f"exact in {xstages}, hi+lo weights in {stages or 'no stage'}, single pass in the others" if tier == 4 else "fast (single fp16 pass)") + (" + flat-content guard" if arith["flat_guard"] else "") + (" + decision guard" if arith["decision_guard"] else "") + (f" + magnitude guard (logit magnitude > {arith['mag_guard_thr']:.3g} re-run exactly)" if arith["mag_guard_kind"] == 2 else ""),
                                  "calibrated_at_load": bool(arith["calibrated"]), "calib_rms_dlogit": arith["calib_rms"], "calib_max_dlogit": arith["calib_max"],
                                  "w2_stages": int(arith["w2_stages"]), "w2_units": int(arith["w2_units"]), "x_stages": int(arith["x_stages"]), "x_units": int(arith["x_units"]), "weight_rounding": int(arith["rounding"]), "decision_guard_margin": arith["guard_margin"],
                                  "magnitude_guard_thr": arith["mag_guard_thr"], "magnitude_guard_flagged_at_calibration": arith["mag_guard_flagged"],
                                  "magnitude_guard_kind": {0: "none", 1: "range (1.5 x the largest calibrated magnitude)", 2: "admission (the tier was admitted behind it)"}[int(arith["mag_guard_kind"])],
                                  "guard_reruns_total": arith["guard_reruns"], "guard_reruns_per_step": round(reruns_per_step, 2),
                                  "guard_rerun_fraction": round(reruns_per_step / B, 5)},
                   "content": args.content if args.flat_frac == 0 else f"{args.content} + {args.flat_frac:g} flat / dither / ramp / low-contrast CUs"},
        "roofline": roofline,
        "cpu_baseline": cpu_baseline,
        "parity": parity,
        "rccl": rccl,
        "per_rank": per_rank,
        "derived": {"model_tflops": round(value * FLOP_PER_CU[size] / 1e12, 1),
                    "mfma_frac_whole_net": round(value / world * FLOP_PER_CU[size] / 1e12 / MFMA_PEAK_TFLOPS, 4),
                    "hbm_layerwise_roofline_frac": round(value / world * LAYERWISE_BYTES_PER_CU[size] / 1e9 / HBM_PEAK_GBS, 4),
                    "whole_path": {"t_bound_ms": round(t_bound_ms, 4), "t_measured_ms": round(ms_per_step, 4),
                                   "frac": round(t_bound_ms / max(ms_per_step, 1e-9), 4),
                                   "note": "sum over launches of max(algorithmic FLOPs / 2.5 PFLOP/s, algorithmic bytes / 8 TB/s) / measured step time"},
                    "sustained_cu_per_s": round(world * sustained["cu_per_s"], 1) if sustained else None,
                    "sustained": None if not sustained else {"steps": sustained["steps"], "seconds": round(sustained["seconds"], 2), "per_gpu_cu_per_s": round(sustained["cu_per_s"], 1),
                                                             "hbm_layerwise_roofline_frac": round(sustained["cu_per_s"] * LAYERWISE_BYTES_PER_CU[size] / 1e9 / HBM_PEAK_GBS, 4)},
                    "mfma_sustained": None if not sustained else {"tflops": round(sustained["mfma_tflops"], 1), "seconds": round(sustained["gemm_seconds"], 2), "gemm": sustained["gemm"],
                                                                  "whole_net_frac_of_sustained": round(value / world * FLOP_PER_CU[size] / 1e12 / sustained["mfma_tflops"], 4)},
                    "batch1_sync_call_us": None if batch1_us is None else round(batch1_us, 1),
                    "host_staged_cu_per_s": None if staged is None else round(staged, 1),
                    "source_sig": source_signature(),
                    "kernels": kernels},
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    if parity is not None and not (parity["within_tolerance"] and parity["split_identical_decisive"]):
        print("bench.py: PARITY FAILURE " + json.dumps(parity), file=sys.stderr)
        sys.exit(3)


if __name__ == "__main__":
    main()
