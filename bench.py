#!/usr/bin/env python3
"""bench.py -- CU-inferences/s of the MI355X MLT-CNN split predictor (BASELINE.json metric).

One "step" = one pass of the hot path (raw int16 org/pred planes + poc/qp already resident in HBM
-> logits + split modes in HBM) over ONE batch of 4096 synthetic 128x128 CUs per GPU.
N > 1: one process per GPU (torch.distributed.run), weights broadcast once over RCCL, the batch is
sharded by rank with no hot-path collective (weak scaling: 4096 CUs per GPU).

Prints ONE JSON line on rank 0.  See DESIGN.md "Measurement" for how roofline / cpu_baseline are defined.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BATCH = 4096
SIZE = 128
FLOP_PER_CU = 1_134_562_340          # 2 x MACs, conv + FC, S = 128
LAYERWISE_BYTES_PER_CU = 8_061_854   # fp16 activations, every layer reads/writes HBM once
COMPULSORY_BYTES_PER_CU = 65_580
MFMA_PEAK_TFLOPS = 2500.0            # dense fp16, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0                # MI355X_MICROARCH.md (6.29 TB/s measured float4 copy)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH, help="CUs per GPU per step (BASELINE: 4096)")
    ap.add_argument("--size", type=int, default=SIZE)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0)
    ap.add_argument("--flags", type=int, default=0, help="mlt_config.flags (1 = exact arithmetic for 128, 2 = fast arithmetic for 64/32/16)")
    ap.add_argument("--latency", action="store_true", help="also time the synchronous one-CU-per-call path (mlt_predict)")
    ap.add_argument("--host-staged", action="store_true",
                    help="also time mlt_predict_batch from pinned HOST buffers (PCIe-inclusive rate; never `value`)")
    args = ap.parse_args()

    import numpy as np
    import torch
    import mltcnn_pkg
    pkg = mltcnn_pkg.load()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # one rank per GPU over RCCL.  Only for exercising this code path on a box with fewer GPUs than ranks
        # (MLT_BENCH_OVERSUBSCRIBE=1): ranks share GPUs and the two init-time collectives go over gloo on host tensors.
        oversub = world > torch.cuda.device_count() and os.environ.get("MLT_BENCH_OVERSUBSCRIBE") == "1"
        torch.cuda.set_device(local_rank % max(torch.cuda.device_count(), 1))
        dist.init_process_group(backend="gloo" if oversub else "nccl", rank=rank, world_size=world)
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
    if rank == 0:
        blob = pkg.weights.synthetic_blob(arch, 10)
    if world > 1:
        blob = pkg.shard.broadcast_blob(blob if rank == 0 else None, dist, cdev)
    m = pkg.MltCnn(device=dev_index, sizes=(size,), blobs={size: blob}, max_batch=B, flags=args.flags)

    # ---- synthetic inputs: rank r owns CUs [r*B, (r+1)*B) of the global batch ----
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
    if dist is not None:
        dist.barrier()
    m.profile_enable(True)  # HIP events around every kernel launch, on the launch stream
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    prof = m.profile_read()
    m.profile_enable(False)
    if dist is not None:
        te = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(te, op=dist.ReduceOp.MAX)
        elapsed = float(te.item())

    # ---- parity spot-check (outside the timed region): first CUs of this rank vs the CPU oracle ----
    parity = None
    if rank == 0:
        import oracle
        k = 8
        ref, ref_split = oracle.Oracle(blob).forward(org[:k], pred[:k], poc[:k], qp[:k])
        got = d_logits[:k].cpu().numpy()
        parity = {"checked_cus": k, "max_abs_dlogit": float(np.abs(got - ref).max()),
                  "split_identical": bool(np.array_equal(d_split[:k].cpu().numpy(), ref_split)), "tolerance": 1e-3}

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
    # ---- roofline of the dominant kernel (largest total device time) ----
    dom = max(prof, key=lambda r: r["total_ms"]) if prof else None
    roofline = None
    kernels = []
    tot_ms = sum(r["total_ms"] for r in prof) or 1.0
    for r in prof:
        avg_ms = r["total_ms"] / max(r["launches"], 1)
        kernels.append({"name": r["name"], "launches": r["launches"], "avg_ms": round(avg_ms, 4),
                        "share": round(r["total_ms"] / tot_ms, 4),
                        "tflops": round(r["flops"] / max(r["total_ms"], 1e-9) / 1e9, 1),
                        "algo_gbs": round(r["bytes"] / max(r["total_ms"], 1e-9) / 1e6, 1)})
    if dom:
        avg_ms = dom["total_ms"] / dom["launches"]
        flops_l, bytes_l = dom["flops"] / dom["launches"], dom["bytes"] / dom["launches"]
        # which roof binds this kernel: arithmetic intensity against the MI355X ridge (2.5 PFLOP/s / 8 TB/s ~ 310 FLOP/B)
        mfma_bound = bytes_l > 0 and flops_l / bytes_l >= MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")  # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        if os.path.exists(tpath):
            t = json.load(open(tpath)).get(f"{dom['name']}@{B}")
            traffic = t["hbm_bytes_per_launch"] * (B / t["batch"]) if t else None
        if mfma_bound:
            ach = flops_l / (avg_ms * 1e-3) / 1e12
            roofline = {"kernel": dom["name"], "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4)}
        else:
            ach = bytes_l / (avg_ms * 1e-3) / 1e9
            roofline = {"kernel": dom["name"], "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
        roofline.update({"traffic": traffic, "avg_launch_ms": round(avg_ms, 4), "launches": dom["launches"],
                         "algo_flops_per_launch": flops_l, "algo_bytes_per_launch": bytes_l,
                         "flop_per_byte": round(flops_l / max(bytes_l, 1.0), 1)})

    cpu_baseline = None
    if world == 1 and not args.no_cpu_baseline:
        import oracle
        cores = os.cpu_count() or 1
        sample = args.cpu_sample or min(B, 24 * cores)
        orc = oracle.Oracle(blob)
        orc.forward(org[:cores], pred[:cores], poc[:cores], qp[:cores], threads=cores)  # warm-up
        c0 = time.perf_counter()
        orc.forward(org[:sample], pred[:sample], poc[:sample], qp[:sample], threads=cores)
        cs = time.perf_counter() - c0
        cpu_baseline = {"value": round(sample / cs, 2), "unit": "CU-inferences/s", "cores": cores, "kind": "port",
                        "sample": f"first {sample} CUs of the same batch, fp32 C oracle (oracle/mlt_oracle.c), OpenMP over CUs, {cs:.1f} s"}

    out = {
        "metric": "CU-inferences/sec (batch 4096, 128x128)", "value": round(value, 1), "unit": "CU-inferences/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16" if not (args.flags & 1 and size == 128) and (size == 128 or args.flags & 2) else "f16x2 (hi+lo pairs)", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: batch {B} synthetic {size}x{size} CU patches per GPU, fp16 MFMA / fp32 accumulate, "
                               "inputs (int16 org+pred, int32 poc/qp) resident in HBM, outputs logits+split in HBM",
                   "batch_per_gpu": B, "cu_size": size, "weights": "synthetic seed 10 (no trained checkpoint is distributed)",
                   "parallelism": f"shard{world}"},
        "roofline": roofline,
        "cpu_baseline": cpu_baseline,
        "parity": parity,
        "derived": {"model_tflops": round(value * FLOP_PER_CU / 1e12, 1),
                    "mfma_frac_whole_net": round(value / world * FLOP_PER_CU / 1e12 / MFMA_PEAK_TFLOPS, 4),
                    "hbm_layerwise_roofline_frac": round(value / world * LAYERWISE_BYTES_PER_CU / 1e9 / HBM_PEAK_GBS, 4),
                    "batch1_sync_call_us": None if batch1_us is None else round(batch1_us, 1),
                    "host_staged_cu_per_s": None if staged is None else round(staged, 1),
                    "kernels": kernels},
    }
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
