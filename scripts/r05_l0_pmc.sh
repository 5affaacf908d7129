#!/bin/bash
# round 5: SQ counters of layer0_stream_kernel (one --pmc pass per group, never combined with traces other than --kernel-trace)
out=gpurun_out/${1:-r05m}
mkdir -p $out
export TMPDIR=/tmp
for grp in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_BUSY_CYCLES SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_MFMA SQ_ACTIVE_INST_ANY" "SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  name=$(echo $grp | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc/$name -o out -- python3 scripts/prof_run.py 4096 2 > $out/pmc_$name.log 2>&1
done
python3 - $out <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(dict)
for d in sorted(glob.glob(os.path.join(out, "pmc", "*/"))):
    f = os.path.join(d, "out_counter_collection.csv")
    if not os.path.exists(f):
        print("no counters in", d); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r.get("Dispatch_Id", 0) or 0)):
        k = r["Kernel_Name"]
        if "layer0" in k or "layer1" in k or "block32" in k or "stem_block" in k or "chain_kernel<128" in k:
            acc[k[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        for c, x in v.items():
            rows[k][c] = sum(x[-2:]) / len(x[-2:])
with open(os.path.join(out, "l0_pmc.txt"), "w") as o:
    for k, v in rows.items():
        line = k + "  " + "  ".join(f"{c}={x:.4g}" for c, x in sorted(v.items()))
        if "GRBM_GUI_ACTIVE" in v:
            cyc = v["GRBM_GUI_ACTIVE"] / 8
            line += f"  | cycles/launch {cyc:.4g}"
            if "SQ_VALU_MFMA_BUSY_CYCLES" in v: line += f" mfma_util {v['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):.3f}"
            if "SQ_LDS_IDX_ACTIVE" in v: line += f" lds_busy {v['SQ_LDS_IDX_ACTIVE'] / (cyc * 256):.3f}"
            if "SQ_LDS_BANK_CONFLICT" in v and v.get("SQ_LDS_IDX_ACTIVE"): line += f" lds_conflict {v['SQ_LDS_BANK_CONFLICT'] / v['SQ_LDS_IDX_ACTIVE']:.3f}"
        print(line); o.write(line + "\n")
PY
