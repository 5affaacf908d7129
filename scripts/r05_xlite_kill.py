import sys, os
import numpy as np
sys.path.insert(0, '.')
import mltcnn_pkg, oracle
pkg = mltcnn_pkg.load()
F = pkg.capi
blob = pkg.weights.synthetic_blob(0, 10)
org, pred = pkg.synth.natural_patches(128, 128, 4242)
poc, qp = pkg.synth.make_scalars(128, 4242)
ref, _ = oracle.Oracle(blob).forward(org, pred, poc, qp, threads=64)
os.environ["MLT_TUNING"] = "1"
m = pkg.MltCnn(device=0, sizes=(128,), blobs={128: blob}, flags=F.FLAG_EXACT_128 | F.FLAG_EXACT_LITE)
_, l = m.predict_batch(org, pred, poc, qp)
d = np.abs(l - ref); print("kill", os.environ.get("MLT_XL_KILL", "0"), "max %.3e rms %.3e" % (d.max(), np.sqrt((d**2).mean())))
