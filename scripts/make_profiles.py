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
sig = json.load(open(os.path.join(run, "bench_line.json")))["derived"]["source_sig"]  # sources the evidence run was built from
subprocess.check_call([sys.executable, os.path.join(ROOT, "scripts", "make_traffic_json.py"),
                       os.path.join(run, "fetch", "out_counter_collection.csv"), os.path.join(run, "write", "out_counter_collection.csv"),
                       "4096", os.path.join(prof, "pmc_traffic.json"), tag, sig], stdout=subprocess.DEVNULL)
shutil.copy(os.path.join(prof, "pmc_traffic.json"), os.path.join(prof, f"{tag}_pmc_traffic.json"))
rows = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(run, "pmc", "*/"))):
    f = os.path.join(d, "out_counter_collection.csv")
    if not os.path.exists(f):
        continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id", 0) or 0)):
        k = r["Kernel_Name"]
        if "conv_" in k or "block" in k or "heads" in k or "chain_" in k or "guard_" in k or "flat_stat" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        for c, x in v.items():
            x = x[-STEPS:]  # the batch launches come last; the load-time calibration runs the same kernels on 16-CU sub-batches first
            rows[k][c] = sum(x) / len(x)
if rows:
    cols = ["GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS", "SQ_WAIT_INST_LDS",
            "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY"]
    with open(os.path.join(prof, f"{tag}_pmc_sq_summary.txt"), "w") as o:
        o.write("# rocprofv3 --kernel-trace --pmc <counters> -- python3 scripts/prof_run.py 4096 2  (one pass per counter group, MI355X)\n")
        o.write("# per-kernel average over the LAST 2 dispatches (the 4096-CU batches; calibration launches excluded); SQ_* summed over the chip; GRBM_GUI_ACTIVE summed over the 8 XCDs\n")
        o.write("# mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); lds_busy = SQ_LDS_IDX_ACTIVE / (GRBM_GUI_ACTIVE/8 * 256 CUs)\n")
        o.write("kernel," + ",".join(cols) + ",mfma_util,lds_busy_frac,lds_conflict_frac\n")
        for k, v in rows.items():
            g = v.get("GRBM_GUI_ACTIVE", 0) / 8
            mu = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (g * 1024) if g else 0
            lb = v.get("SQ_LDS_IDX_ACTIVE", 0) / (g * 256) if g else 0
            lc = v.get("SQ_LDS_BANK_CONFLICT", 0) / v["SQ_LDS_IDX_ACTIVE"] if v.get("SQ_LDS_IDX_ACTIVE") else 0
            short = k.replace("void ", "").split("(")[0][:72]
            o.write('"%s",' % short + ",".join("%.4g" % v.get(c, 0) for c in cols) + ",%.3f,%.3f,%.3f\n" % (mu, lb, lc))
print("wrote profiles/%s_*" % tag)
