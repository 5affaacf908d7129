#!/bin/bash
# round 6, final evidence call: the whole parity suite on the final sources, the evidence run (bench line, rocprofv3 stats, PMC traffic + SQ counters, side legs with all
# 4096 CUs checked, trained families, latencies), the exact-lite range probe
tag=${1:-r06A}
out=gpurun_out/$tag
mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log; tail -4 $out/pytest_gpu.log
bash scripts/evidence_run.sh $tag > $out/evidence.log 2>&1; tail -3 $out/evidence.log | cut -c1-400
timeout 600 python scripts/r06_lite_range_probe.py > $out/lite_range_probe.txt 2>&1; cat $out/lite_range_probe.txt | cut -c1-420
du -sh $out
