#!/bin/bash
# round 6: MLT_FLAT_RANGE 8 -> 4.  The whole parity suite, the need probe (shipped tier with / without the flat guard against the exact arithmetic, per content class) on nine
# seeded sets + the trained families, the tail probes with the classes around the near-flat rule added, and the natural-content / flat-mix bench legs
tag=${1:-r06j}
out=gpurun_out/$tag
mkdir -p $out
timeout 1800 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log; tail -4 $out/pytest_gpu.log
blobs=$(ls tests/data/_blobs/*.mltw 2>/dev/null | tr '\n' ',' | sed 's/,$//')
timeout 900 python scripts/r06_flat_guard_need_probe.py "10,23,24,13,11,21,25,12,22,$blobs" 4096 > $out/flat_guard_need.txt 2>&1
grep -E "weight set|SHIPPED" $out/flat_guard_need.txt | grep -E "weight set|natural|low_contrast|dither|partial_near" | sed 's/unguarded single pass on the FLAGGED CUs/FLAGGED/' | cut -c1-330
timeout 2400 python scripts/tail_probe.py --seeds "10,23,24,13,11,21,25,12,22" --blobs "$blobs" --natural 4096 --near-flat 4096 > $out/tail_probe.txt 2>&1; echo "tail probe rc $?"
grep -E "^seed|=>|low_contrast|dither|natural" $out/tail_probe.txt | cut -c1-230
for c in "--content natural" "--flat-frac 0.25 --steps 20 --warmup 5" ""; do
  python3 bench.py --no-cpu-baseline --cpu-sample 4096 --sustain-s 0 $c > "$out/bench_$(echo $c | tr -d ' -' | cut -c1-14)_.json" 2>> $out/bench.err
done
python3 - $out <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); a = d['config']['arithmetic']
        print(f"{os.path.basename(f):34s} {d['value']:10.0f} CU/s {d['ms_per_step']:.3f} ms reruns/step {a['guard_reruns_per_step']} max {d['parity']['max_abs_dlogit']:.2e} checked {d['parity'].get('checked_cus')} mism {d['parity']['split_mismatch_decisive']}")
    except Exception as e:
        print(f, "FAILED", e)
PY
