#!/bin/bash
# round 6, third GPU call: the tier every seeded weight set reaches under the round-6 search (plain rule, then the same rule behind the magnitude guard) with its
# bench rate and tail probe; the trained families again (exact-lite tier behind the flat guard at 1/16); MFMA shape bit probe; streaming launches from 128 CUs
out=gpurun_out/r06c
mkdir -p $out
timeout 120 scripts/probes/mfma_shape_bits_probe.bin > $out/mfma_shape_bits_probe.txt 2>&1; cat $out/mfma_shape_bits_probe.txt
timeout 900 python - > $out/seed_tiers.txt 2>&1 <<'PY'
import sys, time
sys.path.insert(0, '.')
import mltcnn_pkg
pkg = mltcnn_pkg.load()
for seed in (10, 23, 11, 24, 13, 25, 21, 12, 22):
    blob = pkg.weights.synthetic_blob(0, seed)
    t0 = time.time()
    m = pkg.MltCnn(device=0, sizes=(128,), blobs={128: blob})
    a = m.arithmetic(128)
    t1 = time.time() - t0
    m.close()
    p = pkg.MltCnn(device=0, sizes=(128,), blobs={128: blob}, flags=pkg.capi.FLAG_NO_MAGNITUDE_GUARD)
    b = p.arithmetic(128)
    p.close()
    f = lambda x: f"tier {x['exact']} w2 units 0x{x['w2_units']:x} exact units 0x{x['x_units']:x} rounding {x['rounding']} calib rms {x['calib_rms']:.2e} max {x['calib_max']:.2e} mag thr {x['mag_guard_thr']:.3g} flagged {100 * x['mag_guard_flagged']:.1f} %"
    print(f"seed {seed}: {f(a)}  ({t1:.2f} s) | plain rule: {f(b)}", flush=True)
PY
cat $out/seed_tiers.txt | grep seed
for s in 23 11 24 13 25 21 12 22; do timeout 300 python bench.py --no-cpu-baseline --cpu-sample 4096 --weight-seed $s > $out/bench_seed$s.json 2>> $out/bench.err; done
python - $out <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + '/bench_seed*.json')):
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1]); a = d['config']['arithmetic']
        print(f.split('/')[-1], round(d['value']), 'CU/s |', a['mode'][:150], '| reruns/step', a['guard_reruns_per_step'], '| parity', d['parity']['checked_cus'], 'max', '%.2e' % d['parity']['max_abs_dlogit'], 'mism', d['parity']['split_mismatch_decisive'])
    except Exception as e:
        print(f, 'FAILED', e)
PY
timeout 1500 python scripts/tail_probe.py --seeds 10,23,11,24,13,25,21,12,22 --natural 4096 > $out/tail_probe_seeds.txt 2>&1; echo "tail probe seeds rc $?"
grep -E "^seed|=>" $out/tail_probe_seeds.txt | cut -c1-260
bash scripts/r06_trained_probe.sh r06c > $out/trained.log 2>&1; grep -E "ARITH|CU/s|=>|partial_flat|natural  " $out/trained.log | cut -c1-330
for n in 96 128 160 192 256; do timeout 200 python bench.py --no-cpu-baseline --cpu-sample 64 --batch $n --steps 200 --warmup 50 > $out/bench_batch$n.json 2>> $out/bench.err; python -c "
import json,sys
d=json.loads([l for l in open('$out/bench_batch$n.json') if l.startswith('{')][-1]); print('batch $n', round(d['value']), 'CU/s', [k['name'][:24] for k in d['derived']['kernels']][:2])"; done
