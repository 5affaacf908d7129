#!/bin/bash
# round 6, fourth GPU call: magnitude-guard threshold = the largest magnitude whose CUs meet the refined rule (instead of tolerance / worst relative error):
# seeded sets and the trained families again (tiers, bench, tail probes), the whole parity suite, deferred-batch latency at n = 1 .. 32
out=gpurun_out/r06d
mkdir -p $out
timeout 900 python - > $out/seed_tiers.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, '.')
import mltcnn_pkg
pkg = mltcnn_pkg.load()
sets = [(f"seed {s}", pkg.weights.synthetic_blob(0, s)) for s in (10, 23, 11, 24, 13, 25, 21, 12, 22)] + [("amplifying(10)", pkg.weights.amplifying_blob(0, 10))]
for name, blob in sets:
    t0 = time.time()
    m = pkg.MltCnn(device=0, sizes=(128,), blobs={128: blob})
    a = m.arithmetic(128)
    t1 = time.time() - t0
    m.close()
    f = lambda x: f"tier {x['exact']} w2 units 0x{x['w2_units']:x} exact units 0x{x['x_units']:x} rounding {x['rounding']} calib rms {x['calib_rms']:.2e} max {x['calib_max']:.2e} mag thr {x['mag_guard_thr']:.3g} flagged {100 * x['mag_guard_flagged']:.1f} %"
    print(f"{name}: {f(a)}  ({t1:.2f} s)", flush=True)
PY
cat $out/seed_tiers.txt | grep -E "seed|ampl"
bash scripts/r06_trained_probe.sh r06d > $out/trained.log 2>&1; grep -E "ARITH|CU/s|=>|partial_flat|natural  " $out/trained.log | cut -c1-360
for s in 21 12 25 13 11; do timeout 300 python bench.py --no-cpu-baseline --cpu-sample 4096 --weight-seed $s > $out/bench_seed$s.json 2>> $out/bench.err; done
python - $out <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + '/bench_seed*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1]); a = d['config']['arithmetic']
        print(f.split('/')[-1], round(d['value']), 'CU/s |', a['mode'][:170], '| reruns/step', a['guard_reruns_per_step'], '| parity', d['parity']['checked_cus'], 'max', '%.2e' % d['parity']['max_abs_dlogit'], 'mism', d['parity']['split_mismatch_decisive'])
    except Exception as e:
        print(f, 'FAILED', e)
PY
timeout 1500 python scripts/tail_probe.py --seeds 21,12,25,13,11 --natural 4096 > $out/tail_probe_seeds.txt 2>&1; echo "tail probe seeds rc $?"
grep -E "^seed|=>" $out/tail_probe_seeds.txt | cut -c1-300
timeout 300 python scripts/flush_latency.py 10 40 > $out/flush_latency.txt 2>&1; cat $out/flush_latency.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log
tail -5 $out/pytest_gpu.log
