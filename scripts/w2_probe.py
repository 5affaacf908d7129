import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
import mltcnn_pkg, oracle, torch
pkg = mltcnn_pkg.load()
dev = torch.device("cuda:0")
size, n = 128, 96
for seed in (10, 11, 13, 22, 24):
    blob = pkg.weights.synthetic_blob(0, seed)
    org, pred = pkg.synth.make_patches_bulk(size, n, 777)
    poc, qp = pkg.synth.make_scalars(n, 777)
    ref, rs = oracle.Oracle(blob).forward(org, pred, poc, qp, threads=16)
    m = pkg.MltCnn(device=0, sizes=(size,), blobs={size: blob}, max_batch=4096)
    a = m.arithmetic(size)
    s, l = m.predict_batch(org, pred, poc, qp)
    s1, l1 = m.predict_batch(org[:5], pred[:5], poc[:5], qp[:5])
    err = np.abs(l - ref)
    nb = 4096
    o2, p2 = pkg.synth.make_patches_bulk(size, nb, 5); c2, q2 = pkg.synth.make_scalars(nb, 5)
    t = [torch.from_numpy(x).to(dev) for x in (o2, p2, c2, q2)]
    sp = torch.empty(nb, dtype=torch.int32, device=dev); lg = torch.empty(nb, 9, dtype=torch.float32, device=dev)
    for _ in range(2): m.predict_batch_device(nb, size, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), sp.data_ptr(), lg.data_ptr())
    m.synchronize(); t0 = time.perf_counter()
    for _ in range(5): m.predict_batch_device(nb, size, t[0].data_ptr(), t[1].data_ptr(), t[2].data_ptr(), t[3].data_ptr(), sp.data_ptr(), lg.data_ptr())
    m.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"seed {seed}: tier {a['exact']} calib rms {a['calib_rms']:.2e} max {a['calib_max']:.2e} | vs oracle max {err.max():.2e} rms {np.sqrt((err**2).mean()):.2e} split mism {int((s != rs).sum())} | small==large {np.array_equal(l[:5], l1)} | {nb/dt/1e3:.0f} k CU/s")
    m.close()
