#!/usr/bin/env python3
"""GPU box: is the chip power-limited under this workload?  The persistent grids (MLT_WG_CAP: block32 / 32->64 / chains and stages,
MLT_WG_CAP2: stem_block) at 256 (= all CUs) / 224 / 192 / 128 workgroups.  If a kernel's time does not grow as its CUs shrink, the chip,
not the kernel, is the limit."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for cap in (256, 240, 224, 192, 128, 256):
    env = dict(os.environ, MLT_TUNING="1", MLT_WG_CAP=str(cap), MLT_WG_CAP2=str(cap))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "40", "--warmup", "20"], env=env, capture_output=True, text=True).stdout
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    print(f"cap {cap:3d}: {d['value']:9.0f} CU/s  {d['ms_per_step']:.3f} ms  " + " ".join(f"{k['avg_ms']:.3f}" for k in d["derived"]["kernels"][:6]), flush=True)
