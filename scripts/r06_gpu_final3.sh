#!/bin/bash
# round 6, evidence call on the FINAL sources (MLT_FLAT_RANGE = 6): the whole parity suite, the evidence run (bench line, rocprofv3 stats, PMC traffic + SQ counters, side legs
# with all 4096 CUs checked, trained families, latencies), the exact-lite range probe, the need probe of the flat guard, and the tail probes (479,232 logits per weight set:
# six content classes + natural scenes + the classes around the near-flat rule) of all nine seeded sets and both trained families
tag=${1:-r06E}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log; tail -4 $out/pytest_gpu.log
python3 -c "import mltcnn_pkg; p = mltcnn_pkg.load(); print('source signature', p.build.source_signature(), 'stale', p.build.stale())" >> $out/pytest_gpu.log 2>&1
bash scripts/evidence_run.sh $tag > $out/evidence.log 2>&1; tail -3 $out/evidence.log | cut -c1-400
timeout 600 python scripts/r06_lite_range_probe.py > $out/lite_range_probe.txt 2>&1
blobs=$(ls tests/data/_blobs/*.mltw 2>/dev/null | tr '\n' ',' | sed 's/,$//')
timeout 900 python scripts/r06_flat_guard_need_probe.py "10,23,24,13,11,21,25,12,22,$blobs" 4096 > $out/flat_guard_need.txt 2>&1
timeout 2400 python scripts/tail_probe.py --seeds "10,23,24,13,11,21,25,12,22" --blobs "$blobs" --natural 4096 --near-flat 4096 > $out/tail_probe.txt 2>&1; echo "tail probe rc $?"
grep -E "^seed|=>" $out/tail_probe.txt | cut -c1-160
du -sh $out
