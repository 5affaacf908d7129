#!/usr/bin/env python3
"""Build box: the figures of DESIGN.md's "(d) Current results" table from the files of one evidence run (profiles/<tag>_*).
usage: python scripts/results_table.py r04s     (prints `key: value` lines; the prose of the table stays hand-written)"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
P = os.path.join(ROOT, "profiles")


def line(name):
    with open(os.path.join(P, f"{tag}_{name}.json")) as f:
        return [json.loads(l) for l in f if l.startswith("{")][-1]


d = line("bench_line")
r, dv = d["roofline"], d["derived"]
print(f"headline: {d['value'] / 1e3:.0f} k CU/s, {d['ms_per_step']:.3f} ms/step, {dv['model_tflops']:.0f} TFLOP/s whole net = {dv['mfma_frac_whole_net']:.2f} of 2.5 PF")
print(f"parity: max |dlogit| {d['parity']['max_abs_dlogit']:.1e} over {d['parity']['checked_cus']} CUs, decisive mismatches {d['parity']['split_mismatch_decisive']}, non-decisive {d['parity']['non_decisive']}")
print(f"roofline kernel {r['kernel']}: {r['avg_launch_ms']:.3f} ms, {r['achieved']:.0f} {r['unit']} = {r['frac']:.3f}; traffic {r['traffic']}")
with open(os.path.join(P, f"{tag}_bench_kernel_batch_launches.csv")) as f:
    for row in csv.reader(l for l in f if not l.startswith("#")):
        if row and "chain_kernel<128, 4, 0, 2, 2, 2, 4, 3" in row[0]:
            print(f"  rocprofv3 batch launches of that kernel: {float(row[3]) / 1e6:.3f} ms")
for k in dv["kernels"]:
    print(f"  {k['name'][:44]:44s} {k['avg_ms']:.3f} ms  roof {k.get('roof_frac')}")
print(f"whole path: {dv['whole_path']['t_bound_ms']:.3f} / {dv['whole_path']['t_measured_ms']:.3f} = {dv['whole_path']['frac']:.2f}; HBM layer-wise fraction {dv['hbm_layerwise_roofline_frac']:.2f}")
cb = d["cpu_baseline"]
print("cpu baseline:", f"{cb['value']:.0f} CU/s;", "; ".join(f"{x['impl'][:14]} batch {x.get('batch')}: {x['value']:.0f}" for x in cb["rows"]))
for f in sorted(glob.glob(os.path.join(P, f"{tag}_bench_*.json"))):
    name = os.path.basename(f)[len(tag) + 7:-5]
    if name in ("line",):
        continue
    x = line("bench_" + name)
    a = x["config"]["arithmetic"]
    extra = f" cpu {x['cpu_baseline']['value']:.0f}" if x.get("cpu_baseline") else ""
    print(f"{name:24s} {x['value'] / 1e3:8.0f} k  {x['ms_per_step']:7.3f} ms  err {x['parity']['max_abs_dlogit']:.1e}  reruns {a['guard_reruns_per_step']:6.1f}  {x['dtype'][:90]}{extra}")
e = line("bench_latency_hoststaged")["derived"]
print(f"latency {e.get('batch1_sync_call_us')} us; host-staged {e.get('host_staged_cu_per_s', 0) / 1e3:.0f} k CU/s")
print(open(os.path.join(P, f"{tag}_latency_modes.txt")).read().strip())
print("pmc_traffic.json:", json.load(open(os.path.join(P, "pmc_traffic.json")))["_meta"])
