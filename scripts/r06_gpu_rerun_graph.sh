#!/bin/bash
# round 6: guard re-runs of <= 8 CUs replayed from a hipGraph (guard_fixup_async).  Parity tests that exercise the re-run through every entry point, then the same-box A/B:
# the small models' batches (the decision guard re-runs ~1 CU per 4096-CU step of the 16 x 16 model) with the graphs and with MLT_NO_GRAPH=1 (eager launches)
tag=${1:-r06h}
out=gpurun_out/$tag
mkdir -p $out
timeout 1500 python -m pytest tests/test_hip_parity.py -m gpu -q -k "rerun or guard or flat or content or deferred or magnitude or one_context" > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log; tail -4 $out/pytest_gpu.log
for i in 1 2; do
  for s in 16 32 64; do
    python3 bench.py --no-cpu-baseline --sustain-s 0 --size $s > $out/ab_graph_s${s}_$i.json 2>> $out/ab.err
    MLT_TUNING=1 MLT_NO_GRAPH=1 python3 bench.py --no-cpu-baseline --sustain-s 0 --size $s > $out/ab_eager_s${s}_$i.json 2>> $out/ab.err
  done
  python3 bench.py --no-cpu-baseline --sustain-s 0 --size 16 --content natural --batch 256 > $out/ab_graph_s16_natural256_$i.json 2>> $out/ab.err
  MLT_TUNING=1 MLT_NO_GRAPH=1 python3 bench.py --no-cpu-baseline --sustain-s 0 --size 16 --content natural --batch 256 > $out/ab_eager_s16_natural256_$i.json 2>> $out/ab.err
done
python3 - $out <<'PY' | tee $out/ab_rerun_graph.txt
import json, sys, glob, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "ab_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        a = d['config']['arithmetic']
        print(f"{os.path.basename(f):40s} {d['value']:10.0f} CU/s  {d['ms_per_step']:.3f} ms  reruns/step {a['guard_reruns_per_step']}  max {d['parity']['max_abs_dlogit']:.2e} mism {d['parity']['split_mismatch_decisive']}")
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -5 $out/ab.err
