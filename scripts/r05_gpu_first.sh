#!/bin/bash
# round 5, first GPU call: parity suite, the driver's bench line, the RA-toolset evaluation (tools/run_ra_eval.py --mode gpu)
mkdir -p gpurun_out/r05a
nproc > gpurun_out/r05a/nproc.txt
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r05a/pytest_gpu.log 2>&1; echo "pytest rc $?" >> gpurun_out/r05a/pytest_gpu.log
tail -5 gpurun_out/r05a/pytest_gpu.log
timeout 900 python bench.py > gpurun_out/r05a/bench_line.json 2> gpurun_out/r05a/bench_err.log; echo "bench rc $?"
timeout 3000 python tools/run_ra_eval.py --mode gpu --out gpurun_out/r05a/ra_eval > gpurun_out/r05a/ra_eval.log 2>&1; echo "ra_eval rc $?"
tail -40 gpurun_out/r05a/ra_eval.log | cut -c1-300
rm -rf gpurun_out/r05a/ra_eval/*/*.bin gpurun_out/r05a/ra_eval/torch_model
du -sh gpurun_out/r05a
