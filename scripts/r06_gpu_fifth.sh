#!/bin/bash
# round 6, fifth GPU call: guarded tiers behind the flat guard at 1/16 (trained families: tiers, bench, tail probes); the 16x16x32 forms of the two streaming kernels
# (bit identity against the tiled / chain forms, same-box A/B against round 5's MFMA shape); whole parity suite
out=gpurun_out/r06e
mkdir -p $out
AB_VARS="MLT_L1_MFMA32 MLT_L0_MFMA32" bash scripts/r06_ab_mfma16.sh r06e
bash scripts/r06_trained_probe.sh r06e > $out/trained.log 2>&1; grep -E "ARITH|CU/s|=>|partial_flat|natural  " $out/trained.log | cut -c1-360
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log
tail -5 $out/pytest_gpu.log
