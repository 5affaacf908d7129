"""Debug: per-phase cycle breakdown of conv_mfma_kernel (build with -DMLT_PHASE_TIMING).  GPU box only.
usage: python scripts/phase_timing.py [batch] [exact | seed=<weight seed>]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mltcnn_pkg  # noqa: E402

pkg = mltcnn_pkg.load()
out = os.path.join(os.path.dirname(pkg.build.__file__), "_variants", "lib_phase.so")
if not os.path.exists(out) or os.environ.get("MLT_PHASE_REBUILD"):
    pkg.build.build_lib(force=True, defines=["MLT_PHASE_TIMING=1"], out=out)
os.environ["MLT_TUNING"] = "1"
os.environ["MLT_LIB_PATH"] = out
import torch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
seed = next((int(a[5:]) for a in sys.argv[2:] if a.startswith('seed=')), 10)
blob = pkg.weights.synthetic_blob(0, seed)
m = pkg.MltCnn(device=0, sizes=(128,), blobs={128: blob}, max_batch=n, flags=pkg.capi.FLAG_EXACT_128 if 'exact' in sys.argv[2:] else 0 if seed != 10 else pkg.capi.FLAG_NO_CALIBRATION)
org, pred = pkg.synth.make_patches_bulk(128, n, 3)
poc, qp = pkg.synth.make_scalars(n, 3)
dev = torch.device("cuda:0")
t = [torch.from_numpy(x).to(dev) for x in (org, pred, poc, qp)]
split = torch.empty(n, dtype=torch.int32, device=dev)
lg = torch.empty(n, 9, dtype=torch.float32, device=dev)
lib = C.CDLL(out)
buf = (C.c_ulonglong * 128)()
for it in range(3):
    m.predict_batch_device(n, 128, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), split.data_ptr(), lg.data_ptr())
    m.synchronize()
    lib.mlt_debug_phase_read(buf, 1)
names = ["wait/issue patch (DMA: residual issue)", "commit (DMA: dma issue)", "barrier+pf issue", "mfma loop", "step wait+barrier", "epilogue", "tile barrier", "tile setup"]
for kid in range(16):
    row = [buf[kid * 8 + i] for i in range(8)]
    tot = sum(row)
    if not tot:
        continue
    if kid == 13:
        print("block32_kernel: phases = commit patch | barrier | issue next patch | conv1 -> T | barrier | conv2 + residual + store | barrier | tile decode")
    if kid == 14:
        print("stem_block_kernel: phases = commit raw | barrier | issue next raw | phase 1 (composed 5x5 -> T) | barrier | phase 2 (conv2 + shortcut + store) | barrier | tile decode")
    print(f"kernel id {kid} (cin={32 * (kid & 7)}, stride={2 if kid & 8 else 1}): total {tot / 1e6:.1f} Mcycles (wave 0, all WGs, all launches)")
    for nm, v in zip(names, row):
        print(f"   {nm:40s} {100.0 * v / tot:5.1f} %")

cbuf = (C.c_ulonglong * 64)()
lib.mlt_debug_phase_read_chain(cbuf, 0)
cnames = ["tile top: wait for patches 0/1 + first weight step (64 chain: barrier after own vmcnt wait)", "S2 steps: reads + MFMA (64 chain: own wait for the input DMA at the tile top)", "S2 steps: end wait + barrier", "S2 epilogue (bias, t -> LDS, sc -> regs, barrier)",
          "conv 0 steps: reads + MFMA", "conv 0 steps: end wait + barrier", "conv 0 epilogue", "conv 1 steps: reads + MFMA", "conv 1 steps: end wait + barrier",
          "conv 1 epilogue", "conv 2 steps: reads + MFMA", "conv 2 steps: end wait + barrier", "conv 2 epilogue (HBM / GAP) + next tile's patch DMA issue"]
for cid, nm in enumerate(["64-channel chain (default) or chain 128@16 (MLT_NO_CHAIN_S2=1)", "stage 128@16 (S2)", "chain 256@8", "stage 256@8 (S2)"]):
    row = [cbuf[cid * 16 + i] for i in range(16)]
    tot = sum(row)
    if not tot:
        continue
    print(f"chain_kernel {nm}: total {tot / 1e6:.1f} Mcycles (wave 0 of every workgroup, the last launch loop)")
    for k, v in zip(cnames, row):
        print(f"   {k:70s} {100.0 * v / tot:5.1f} %   {v / 1e6:9.2f} Mcyc")

lbuf = (C.c_ulonglong * 32)()
if hasattr(lib, "mlt_debug_phase_read_l0") and lib.mlt_debug_phase_read_l0(lbuf, 0) == 0:
    lnames = ["top of the step (S1: raw commit + issue; S2: shortcut)", "fragment reads + MFMA issue", "epilogue (bias, activation, stores issued)", "step end: prefetch, wait for the stores, barrier"]
    for st in range(4):
        row = [lbuf[st * 8 + i] for i in range(4)]
        tot = sum(row)
        if tot:
            print(f"layer0_stream_kernel stage S{st + 1} (unit 0 of every workgroup, the last launch): total {tot / 1e6:.1f} Mcycles")
            for nm, v in zip(lnames, row):
                print(f"   {nm:64s} {100.0 * v / tot:5.1f} %   {v / 1e6:9.2f} Mcyc")
