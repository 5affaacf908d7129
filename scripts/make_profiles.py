#!/usr/bin/env python3
"""Turn one evidence run (gpurun_out/<run>/, see DESIGN.md "(d) Measurement" for the commands) into the committed
profiles/<tag>_* files: bench line, rocprofv3 kernel stats, HBM traffic per kernel (pmc_traffic.json) and the SQ
MFMA / LDS counter summary.  usage: make_profiles.py gpurun_out/r01g r01g"""
import collections
import csv
import json
import glob
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
run, tag = sys.argv[1], sys.argv[2]
STEPS = 2  # scripts/prof_run.py 4096 2
prof = os.path.join(ROOT, "profiles")
shutil.copy(os.path.join(run, "stats", "out_kernel_stats.csv"), os.path.join(prof, f"{tag}_bench_kernel_stats.csv"))
shutil.copy(os.path.join(run, "bench_line.json"), os.path.join(prof, f"{tag}_bench_line.json"))
if os.path.exists(os.path.join(run, "bench_extras.json")):
    shutil.copy(os.path.join(run, "bench_extras.json"), os.path.join(prof, f"{tag}_bench_latency_hoststaged.json"))
# rocprofv3's own --stats averages EVERY dispatch of a kernel name, the load-time calibration's 96-CU sub-batches included (6 of the 126
# dispatches of the 128 stage): a second table averages the batch launches only (dispatches longer than half the longest of their name)
trace = os.path.join(run, "stats", "out_kernel_trace.csv")
if os.path.exists(trace):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(trace)):
        per[r["Kernel_Name"]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    with open(os.path.join(prof, f"{tag}_bench_kernel_batch_launches.csv"), "w") as o:
        o.write("# from the same rocprofv3 --kernel-trace --stats run as %s_bench_kernel_stats.csv: per kernel, the dispatches longer than half the longest one (= the 4096-CU batch launches)\n" % tag)
        o.write("kernel,batch_dispatches,all_dispatches,avg_ns_batch_dispatches,avg_ns_all_dispatches\n")
        for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
            big = [x for x in v if x > 0.5 * max(v)]
            o.write('"%s",%d,%d,%.1f,%.1f\n' % (k[:110], len(big), len(v), sum(big) / len(big), sum(v) / len(v)))
for f in sorted(glob.glob(os.path.join(run, "bench_*.json"))):  # round 4: the other tiers / content mixes / sizes of the same run
    base = os.path.basename(f)
    if base not in ("bench_line.json", "bench_extras.json"):
        shutil.copy(f, os.path.join(prof, f"{tag}_{base}"))
if os.path.exists(os.path.join(run, "lat", "out_kernel_trace.csv")):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "latency_timeline.py"), os.path.join(run, "lat", "out_kernel_trace.csv"),
                           os.path.join(prof, f"{tag}_single_cu_timeline.csv")])
if os.path.exists(os.path.join(run, "flush_latency.txt")):
    open(os.path.join(prof, f"{tag}_flush_latency.txt"), "w").write("".join(l for l in open(os.path.join(run, "flush_latency.txt")) if "amdgpu.ids" not in l))
if os.path.exists(os.path.join(run, "latency_modes.txt")):
    open(os.path.join(prof, f"{tag}_latency_modes.txt"), "w").write("".join(l for l in open(os.path.join(run, "latency_modes.txt")) if "amdgpu.ids" not in l))
sig = json.load(open(os.path.join(run, "bench_line.json")))["derived"]["source_sig"]  # sources the evidence run was built from
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "make_traffic_json.py"),
                       os.path.join(run, "fetch", "out_counter_collection.csv"), os.path.join(run, "write", "out_counter_collection.csv"),
                       "4096", os.path.join(prof, "pmc_traffic.json"), tag, sig], stdout=subprocess.DEVNULL)
shutil.copy(os.path.join(prof, "pmc_traffic.json"), os.path.join(prof, f"{tag}_pmc_traffic.json"))
def sq_summary(pmc_dir, out_path, header, last):
    """One line per kernel from the per-group counter passes under pmc_dir; last = dispatches per kernel to average (None: all of them)."""
    rows = collections.defaultdict(dict)
    for d in sorted(glob.glob(os.path.join(pmc_dir, "*/"))):
        f = os.path.join(d, "out_counter_collection.csv")
        if not os.path.exists(f):
            continue
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id", 0) or 0)):
            k = r["Kernel_Name"]
            if "conv_" in k or "block" in k or "heads" in k or "chain_" in k or "guard_" in k or "flat_stat" in k or "layer0_" in k or "layer1_" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            for c, x in v.items():
                x = x[-last:] if last else x  # the batch launches come last; the load-time calibration runs the same kernels on 16-CU sub-batches first
                rows[k][c] = sum(x) / len(x)
    if not rows:
        return
    cols = ["GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS",
            "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY"]
    with open(out_path, "w") as o:
        o.write(header)
        o.write("# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); lds_busy = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE/8 * 256 CUs)\n")
        o.write("kernel," + ",".join(cols) + ",mfma_util,lds_busy_frac,lds_conflict_frac\n")
        for k, v in rows.items():
            g = v.get("GRBM_GUI_ACTIVE", 0) / 8
            mu = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g * 1024) if g else 0
            lb = v.get("SQ_LDS_IDX_ACTIVE", 0) / (g * 256) if g else 0
            lc = v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"] if v.get("SQ_LDS_IDX_ACTIVE") else 0
            short = k.replace("void ", "").split("(")[0][:96]
            o.write('"%s",' % short + ",".join("%.4g" % v.get(c, 0) for c in cols) + ",%.3f,%.3f,%.3f\n" % (mu, lb, lc))


sq_summary(os.path.join(run, "pmc"), os.path.join(prof, f"{tag}_pmc_sq_summary.txt"),
           "# rocprofv3 --kernel-trace --pmc <counters> -- python3 scripts/prof_run.py 4096 2  (one pass per counter group, MI355X)\n"
           "# per-kernel average over the LAST 2 dispatches (the 4096-CU batches; calibration launches excluded); SQ_* summed over the chip; GRBM_GUI_ACTIVE summed over the 8 XCDs\n", STEPS)
# round 6: the EXACT arithmetic's per-conv launches (configured exact: no calibration, every dispatch is a 2048-CU batch launch; a kernel runs several layers of one
# shape per step -- the line is the average over all of them)
sq_summary(os.path.join(run, "pmc_exact"), os.path.join(prof, f"{tag}_pmc_sq_summary_exact.txt"),
           "# rocprofv3 --kernel-trace --pmc <counters> -- python3 scripts/prof_run.py 2048 2 1  (MLT_FLAG_EXACT_128; one pass per counter group, MI355X)\n"
           "# per-kernel average over ALL dispatches (no calibration in this configuration); SQ_* summed over the chip; GRBM_GUI_ACTIVE summed over the 8 XCDs\n", None)
print("wrote profiles/%s_*" % tag)
