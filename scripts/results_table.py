#!/usr/bin/env python3
"""Build box: the figures of DESIGN.md's "(d) Current results" table from the files of one evidence run (profiles/<tag>_*).
usage: python scripts/results_table.py r04s [--markdown]    (key: value lines, or the table itself with the figures filled in)"""
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
# the device kernel behind the roofline launch (the profile name of the launch -> a prefix of its rocprofv3 kernel name)
ROOF = {"stage_128": "chain_kernel<128, 4, 0, 2, 2, 2, 4, 3", "stage_256": "chain_kernel<256, 3, 1, 2, 1, 2, 4, 3", "chain3_s1_64": "chain_kernel<64, 5, 0, 2, 4, 1, 8",
        "layer0_stream": "layer0_stream_kernel", "layer1_stream": "layer1_stream_kernel"}
roof_kernel = next((v for k, v in ROOF.items() if r["kernel"].startswith(k)), "chain_kernel<128, 4, 0, 2, 2, 2, 4, 3")
print(f"headline: {d['value'] / 1e3:.0f} k CU/s, {d['ms_per_step']:.3f} ms/step, {dv['model_tflops']:.0f} TFLOP/s whole net = {dv['mfma_frac_whole_net']:.2f} of 2.5 PF")
print(f"parity: max |dlogit| {d['parity']['max_abs_dlogit']:.1e} over {d['parity']['checked_cus']} CUs, decisive mismatches {d['parity']['split_mismatch_decisive']}, non-decisive {d['parity']['non_decisive']}")
print(f"roofline kernel {r['kernel']}: {r['avg_launch_ms']:.3f} ms, {r['achieved']:.0f} {r['unit']} = {r['frac']:.3f}; traffic {r['traffic']}")
with open(os.path.join(P, f"{tag}_bench_kernel_batch_launches.csv")) as f:
    for row in csv.reader(l for l in f if not l.startswith("#")):
        if row and roof_kernel in row[0]:
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
    try:
        x = line("bench_" + name)
    except (IndexError, OSError):
        print(f"{name:24s} (no JSON line)")
        continue
    a = x["config"]["arithmetic"]
    extra = f" cpu {x['cpu_baseline']['value']:.0f}" if x.get("cpu_baseline") else ""
    print(f"{name:24s} {x['value'] / 1e3:8.0f} k  {x['ms_per_step']:7.3f} ms  err {x['parity']['max_abs_dlogit']:.1e}  reruns {a['guard_reruns_per_step']:6.1f}  {x['dtype'][:90]}{extra}")
e = line("bench_latency_hoststaged")["derived"]
print(f"latency {e.get('batch1_sync_call_us')} us; host-staged {e.get('host_staged_cu_per_s', 0) / 1e3:.0f} k CU/s")
print(open(os.path.join(P, f"{tag}_latency_modes.txt")).read().strip())
print("pmc_traffic.json:", json.load(open(os.path.join(P, "pmc_traffic.json")))["_meta"])


if "--markdown" in sys.argv:
    def b(name):
        return line("bench_" + name)
    k = {x["name"].split("(")[0].split("_h")[0]: x for x in dv["kernels"]}
    km = {x["name"]: x for x in dv["kernels"]}
    def ms(prefix):
        return next(v["avg_ms"] for n, v in km.items() if n.startswith(prefix))
    def rf(prefix):
        return next(v.get("roof_frac") for n, v in km.items() if n.startswith(prefix))
    sq = {}
    with open(os.path.join(P, f"{tag}_pmc_sq_summary.txt")) as f:
        rows = [l for l in f if not l.startswith("#")]
    hdr = rows[0].strip().split(",")
    for l in csv.reader(rows[1:]):
        if l and roof_kernel in l[0]:
            sq = dict(zip(hdr, l))
    batch_ms = None
    with open(os.path.join(P, f"{tag}_bench_kernel_batch_launches.csv")) as f:
        for row in csv.reader(l for l in f if not l.startswith("#")):
            if row and roof_kernel in row[0]:
                batch_ms = float(row[3]) / 1e6
    lat = {l.split()[1]: float(l.split()[4]) for l in open(os.path.join(P, f"{tag}_latency_modes.txt")) if l.startswith("tier")}
    seeds = {n: b(f"seed{n}") for n in (11, 12, 13, 21, 22, 23, 24, 25) if os.path.exists(os.path.join(P, f"{tag}_bench_seed{n}.json"))}
    sz = {n: b(f"s{n}") for n in (64, 32, 16)}
    nat, flat, ex = b("natural"), b("flat25"), b("exact")
    dg = b("no_decision_guard") if os.path.exists(os.path.join(P, f"{tag}_bench_no_decision_guard.json")) else b("decision_guard")  # (round 5: the guard is the default; the extra leg is the run WITHOUT it)
    rows_cpu = {str(x.get("batch"))[:4]: x["value"] for x in cb["rows"] if "EVERY call" not in x.get("impl", "")}
    reload_row = next((x["value"] for x in cb["rows"] if "EVERY call" in x.get("impl", "")), None)
    print(f"| **CU-inferences/s, batch 4096 x 128x128**, seed 10 | **{d['value'] / 1e3:.0f} k** ({d['ms_per_step']:.2f} ms per batch); {dv['model_tflops']:.0f} TFLOP/s over the whole net = {dv['mfma_frac_whole_net']:.2f} of 2.5 PFLOP/s |")
    print(f"| parity over the whole timed batch | max |dlogit| {d['parity']['max_abs_dlogit']:.1e}; {d['parity']['split_mismatch_decisive']} split mismatches; {d['parity']['non_decisive']} CUs below the decidable margin |")
    print(f"| roofline kernel {r['kernel']} | {r['avg_launch_ms']:.3f} ms (HIP events; rocprofv3 batch launches {batch_ms:.3f} ms) = {r['achieved']:.0f} TFLOP/s = {r['frac']:.3f}; SQ_VALU_MFMA_BUSY {100 * float(sq.get('mfma_util', 0)):.1f} %, bank conflicts {100 * float(sq.get('lds_conflict_frac', 0)):.1f} %; PMC traffic {next(v['hbm_bytes_per_launch'] for n, v in json.load(open(os.path.join(P, f'{tag}_pmc_traffic.json'))).items() if n.startswith(r['kernel'].split('(')[0])) / 1e9:.2f} GB vs {r['algo_bytes_per_launch'] / 1e9:.2f} GB |")
    front = (f"layer0 + layer1.0.conv1 streamed {ms('layer0_stream'):.3f} ({rf('layer0_stream'):.2f})" if any(n.startswith("layer0_stream") for n in km) else
             f"layer0.0 {ms('stem+block'):.3f} ({rf('stem+block'):.2f}) . layer0.1 {ms('block_s1'):.3f} ({rf('block_s1'):.2f}) . 32->64 s2 {ms('conv3x3_s2_32to64'):.3f} ({rf('conv3x3_s2_32to64'):.2f})")
    print(f"| other launches | {front} . 64-channel stage {ms('layer1_stream' if any(n.startswith('layer1_stream') for n in km) else 'chain3_s1_64'):.3f} ({rf('layer1_stream' if any(n.startswith('layer1_stream') for n in km) else 'chain3_s1_64'):.2f}) . layer3 {ms('stage_256'):.3f} ({rf('stage_256'):.2f}) . heads {ms('heads'):.3f}" + (f" . guard select {ms('guard_select'):.3f}" if any(n.startswith('guard_select') for n in km) else " (with the guards' selection)") + " |")
    print(f"| whole path | {dv['whole_path']['t_bound_ms']:.3f} / {dv['whole_path']['t_measured_ms']:.3f} = {dv['whole_path']['frac']:.2f}; HBM layer-wise fraction {dv['hbm_layerwise_roofline_frac']:.2f} |")
    print(f"| decision guard | {dg['value'] / 1e3:.0f} k |")
    print("| other weight sets | " + " . ".join(f"{n}: {seeds[n]['value'] / 1e3:.0f} k ({seeds[n]['parity']['max_abs_dlogit']:.1e})" for n in (23, 24, 13, 11, 21, 25, 12, 22) if n in seeds) + f" . exact: {ex['value'] / 1e3:.0f} k |")
    print(f"| content | natural {nat['value'] / 1e3:.0f} k ({nat['config']['arithmetic']['guard_rerun_fraction'] * 100:.2f} % flagged, {nat['parity']['max_abs_dlogit']:.1e}) . 25 % flat {flat['value'] / 1e3:.0f} k |")
    print(f"| CPU baseline | {cb['value']:.0f} CU/s; one CU at a time {rows_cpu.get('1', 0):.0f}; C oracle {rows_cpu.get('None', 0):.0f}" + (f"; weights re-read on every call (the reference's call pattern) {reload_row:.0f}" if reload_row else "") + " |")
    print(f"| latency / host-staged | {e.get('batch1_sync_call_us'):.0f} us (tier 3: {lat.get('3', 0):.0f}, tier 4: {lat.get('4', 0):.0f}) / {e.get('host_staged_cu_per_s', 0) / 1e3:.0f} k |")
    print("| 64 / 32 / 16 | " + " / ".join(f"{sz[n]['value'] / 1e6:.2f} M" for n in (64, 32, 16)) + "; max |dlogit| " + " / ".join(f"{sz[n]['parity']['max_abs_dlogit']:.1e}" for n in (64, 32, 16)) + "; CPU " + " / ".join(f"{sz[n]['cpu_baseline']['value'] / 1e3:.1f} k" for n in (64, 32, 16)) + " |")
