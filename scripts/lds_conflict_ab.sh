export TMPDIR=/tmp
for v in base swz_old ko_mask; do
  MLT_LIB_PATH=$PWD/fastintercu-vvc_amd/_variants/lib_$v.so rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d gpurun_out/conf/$v -o out -- python3 scripts/prof_run.py 4096 2 > /dev/null 2>&1
  python3 - <<PY
import csv,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("gpurun_out/conf/$v/out_counter_collection.csv")):
    if "chain_kernel" in r["Kernel_Name"]: acc[r["Kernel_Name"][:24]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    c=sum(v["SQ_LDS_BANK_CONFLICT"])/len(v["SQ_LDS_BANK_CONFLICT"]); a=sum(v["SQ_LDS_IDX_ACTIVE"])/len(v["SQ_LDS_IDX_ACTIVE"])
    print("$v", k, "conflict %.3g active %.3g frac %.3f"%(c,a,c/a))
PY
done
SWEEP_FLAGS=16 python scripts/sweep_cfg.py run 2>&1 | tail -5
