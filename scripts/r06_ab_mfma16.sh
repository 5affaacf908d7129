#!/bin/bash
# round 6: same-box A/B of the v_mfma_f32_16x16x32_f16 forms against round 5's 32x32x16 forms (MLT_TUNING=1 MLT_L1_MFMA32=1 / MLT_L0_MFMA32=1 / MLT_CHAIN_MFMA32=1 select the
# old form of one kernel family), alternating, + the bit-identity tests of the streaming launches against the tiled / chain forms
out=gpurun_out/${1:-r06e}
mkdir -p $out
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "layer0_stream or full_batch_4096_properties" > $out/pytest_bits.log 2>&1; echo "pytest (bit identity) rc $?"; tail -3 $out/pytest_bits.log
line() { python - "$1" "$2" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[2], round(d['value']), 'CU/s', d['ms_per_step'], 'ms |', ' | '.join(f"{k['name'][:14]} {k['avg_ms']:.3f}" for k in d['derived']['kernels'][:4]), '| parity', '%.2e' % d['parity']['max_abs_dlogit'])
PY
}
for rep in 1 2 3; do
  for v in ${AB_VARS:-MLT_L1_MFMA32}; do
    MLT_TUNING=1 env $v=1 python bench.py --no-cpu-baseline --cpu-sample 64 --steps 100 --warmup 30 > $out/ab_${v}_$rep.json 2>> $out/ab.err; line $out/ab_${v}_$rep.json "$v=1 (32x32x16)"
  done
  python bench.py --no-cpu-baseline --cpu-sample 64 --steps 100 --warmup 30 > $out/ab_default_$rep.json 2>> $out/ab.err; line $out/ab_default_$rep.json "default (16x16x32)"
done
