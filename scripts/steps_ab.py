#!/usr/bin/env python3
"""GPU box: does the timed region carry a fixed cost?  bench.py at several step counts and guard-wait modes, back to back."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for mode, steps, warm in (("default", 10, 3), ("spin", 10, 3), ("block", 10, 3), ("default", 10, 20), ("spin", 10, 20), ("default", 50, 5), ("spin", 50, 5), ("noguard", 10, 3), ("noguard", 50, 5)):
    env = dict(os.environ, MLT_TUNING="1")
    for k in ("MLT_GUARD_SPIN_WAIT", "MLT_GUARD_BLOCKING_WAIT"):
        env.pop(k, None)
    if mode == "spin":
        env["MLT_GUARD_SPIN_WAIT"] = "1"
    if mode == "block":
        env["MLT_GUARD_BLOCKING_WAIT"] = "1"
    flags = "24" if mode == "noguard" else "0"   # NO_CALIBRATION | NO_FLAT_GUARD: the device entry never synchronises
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--steps", str(steps), "--warmup", str(warm), "--flags", flags], env=env, capture_output=True, text=True).stdout
    d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
    ks = sum(k['avg_ms'] for k in d['derived']['kernels'])
    print(f"{mode:8s} steps {steps:4d} warmup {warm:2d}: {d['value']:10.0f} CU/s  {d['ms_per_step']:.3f} ms/step  kernels {ks:.3f} ms  gap x steps = {(d['ms_per_step'] - ks) * steps:.2f} ms", flush=True)
