#!/bin/bash
# round 6, second GPU call: the magnitude guard -- its GPU test + the calibration test, the trained families, then the whole parity suite and the bench line
out=gpurun_out/r06b
mkdir -p $out
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -s -k "magnitude_guard or calibrate_on_caller" > $out/pytest_mag.log 2>&1; echo "pytest (magnitude guard) rc $?"
grep -E "behind the magnitude|plain rule|max \|dlogit\||passed|failed|Error|error" $out/pytest_mag.log | cut -c1-400 | tail -12
bash scripts/r06_trained_probe.sh r06b
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log
tail -4 $out/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline --cpu-sample 4096 > $out/bench_seed10.json 2> $out/bench_err.log; echo "bench rc $?"
cut -c1-300 $out/bench_seed10.json
