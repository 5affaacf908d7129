#!/bin/bash
# round 6, last evidence call (sources with the guards' selection inside the heads kernel): the whole parity suite, a same-box A/B of the two forms of the
# selection, the evidence run (bench line, rocprofv3 stats, PMC traffic + SQ counters, side legs with all 4096 CUs checked, trained families, latencies),
# the exact-lite range probe, and the SQ counters of the EXACT arithmetic's per-conv launches (what the guards' re-runs and the exact tiers run on)
tag=${1:-r06C}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log; tail -4 $out/pytest_gpu.log
for i in 1 2; do
  python3 bench.py --no-cpu-baseline --sustain-s 0 > $out/ab_select_fused_$i.json 2>> $out/ab.err
  MLT_TUNING=1 MLT_GUARD_SELECT_KERNEL=1 python3 bench.py --no-cpu-baseline --sustain-s 0 > $out/ab_select_kernel_$i.json 2>> $out/ab.err
  python3 bench.py --no-cpu-baseline --sustain-s 0 --content natural > $out/ab_select_fused_natural_$i.json 2>> $out/ab.err
  MLT_TUNING=1 MLT_GUARD_SELECT_KERNEL=1 python3 bench.py --no-cpu-baseline --sustain-s 0 --content natural > $out/ab_select_kernel_natural_$i.json 2>> $out/ab.err
done
python3 - $out <<'PY' | tee $out/ab_select.txt
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "ab_select_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{os.path.basename(f):40s} {d['value']:10.0f} CU/s  {d['ms_per_step']:.3f} ms  max {d['parity']['max_abs_dlogit']:.2e} mism {d['parity']['split_mismatch_decisive']}")
    except Exception as e:
        print(f, "FAILED", e)
PY
bash scripts/evidence_run.sh $tag > $out/evidence.log 2>&1; tail -3 $out/evidence.log | cut -c1-400
timeout 600 python scripts/r06_lite_range_probe.py > $out/lite_range_probe.txt 2>&1; tail -30 $out/lite_range_probe.txt | cut -c1-300
# the tail probe (>= 300 k logits per weight set over six content classes + natural scenes) of the shipped tiers on the final sources: the seeded set of the headline, the
# seeded set admitted behind the magnitude guard, the two trained families
blobs=$(ls tests/data/_blobs/*.mltw 2>/dev/null | tr '\n' ',' | sed 's/,$//')
timeout 2400 python scripts/tail_probe.py --seeds "10,21" --blobs "$blobs" --natural 4096 > $out/tail_probe.txt 2>&1; echo "tail probe rc $?"
grep -E "^seed|=>|natural|texture" $out/tail_probe.txt | cut -c1-330 | tail -40
for grp in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_BUSY_CYCLES SQ_INSTS_LDS"; do
  name=$(echo $grp | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc_exact/$name -o out -- python3 scripts/prof_run.py 2048 2 1 > $out/pmc_exact_$name.log 2>&1
done
du -sh $out
