#!/usr/bin/env python3
"""GPU box: A/B of the three ways mlt_predict_batch_device waits for the guard's flagged-CU count (default: sleep for the expected batch
time, then poll; MLT_GUARD_SPIN_WAIT=1; MLT_GUARD_BLOCKING_WAIT=1): CU/s, step time and host CPU seconds of the benched process."""
import json
import os
import resource
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for mode in ("default", "spin", "block") * 2:
    env = dict(os.environ, MLT_TUNING="1")
    env.pop("MLT_GUARD_SPIN_WAIT", None)
    env.pop("MLT_GUARD_BLOCKING_WAIT", None)
    if mode == "spin":
        env["MLT_GUARD_SPIN_WAIT"] = "1"
    if mode == "block":
        env["MLT_GUARD_BLOCKING_WAIT"] = "1"
    r0 = resource.getrusage(resource.RUSAGE_CHILDREN)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", "200", "--warmup", "5"], env=env, capture_output=True, text=True).stdout
    r1 = resource.getrusage(resource.RUSAGE_CHILDREN)
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    print(f"{mode:8s} {d['value']:10.0f} CU/s  {d['ms_per_step']:.3f} ms/step  kernels {sum(k['avg_ms'] for k in d['derived']['kernels']):.3f} ms  "
          f"host CPU {r1.ru_utime - r0.ru_utime + r1.ru_stime - r0.ru_stime:.1f} s (whole process, 400 steps incl. the profiled repeat)", flush=True)
