#!/usr/bin/env python3
"""Build box: turn the rocprofv3 --kernel-trace CSV of scripts/latency_run.py into a per-launch timeline of ONE mlt_predict call
(kernel, duration, gap to the previous kernel's end), averaged over the last calls.  usage: latency_timeline.py <kernel_trace.csv> [out.csv]"""
import csv
import re
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(.*", "", r["Kernel_Name"])[:70] for r in rows]
# a call = the kernels from one stem kernel (first kernel of the network) to the next
starts = [i for i, nme in enumerate(names) if nme.startswith("void stem_block_kernel") or nme.startswith("stem_block_kernel") or "stem5_kernel" in nme]
calls = [(starts[k], starts[k + 1]) for k in range(len(starts) - 1)]
calls = [c for c in calls if c[1] - c[0] == calls[-1][1] - calls[-1][0]][-16:]   # the steady-state (graph replay) calls
acc = defaultdict(lambda: [0.0, 0.0, 0])
order = []
for a, b in calls:
    prev_end = None
    for j in range(a, b):
        key = (j - a, names[j])
        if key not in acc:
            order.append(key)
        s, e = int(rows[j]["Start_Timestamp"]), int(rows[j]["End_Timestamp"])
        acc[key][0] += (e - s) / 1e3
        acc[key][1] += 0.0 if prev_end is None else (s - prev_end) / 1e3
        acc[key][2] += 1
        prev_end = e
out = open(sys.argv[2], "w") if len(sys.argv) > 2 else sys.stdout
print("index,kernel,avg_duration_us,avg_gap_before_us,samples", file=out)
tot_d = tot_g = 0.0
for key in order:
    d, g, c = acc[key]
    tot_d += d / c; tot_g += g / c
    print(f"{key[0]},{key[1]},{d / c:.2f},{g / c:.2f},{c}", file=out)
print(f"total,,{tot_d:.2f},{tot_g:.2f},{len(calls)}", file=out)
