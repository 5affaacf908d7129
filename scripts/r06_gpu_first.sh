#!/bin/bash
# round 6, first GPU call: MFMA shape probe, the parity suite on the rebuilt library (source signature embedded), the driver's bench line
out=gpurun_out/r06a
mkdir -p $out
nproc > $out/nproc.txt
timeout 300 scripts/probes/mfma_shape_probe.bin 2.0 > $out/mfma_shape_probe.txt 2>&1; echo "probe rc $?"
cat $out/mfma_shape_probe.txt | cut -c1-220
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log
tail -5 $out/pytest_gpu.log
timeout 600 python bench.py > $out/bench_line.json 2> $out/bench_err.log; echo "bench rc $?"
cut -c1-700 $out/bench_line.json
