#!/bin/bash
# round 6, last GPU call: the whole parity suite on the final sources (signature = evidence run r06C's) and the tail probe (331,776 logits per weight set: six content
# classes + natural scenes) of all nine seeded weight sets on those sources
tag=${1:-r06D}
out=gpurun_out/$tag
mkdir -p $out
timeout 1800 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; echo "pytest rc $?" >> $out/pytest_gpu.log; tail -4 $out/pytest_gpu.log
python3 -c "import mltcnn_pkg; p = mltcnn_pkg.load(); print('source signature', p.build.source_signature(), 'stale', p.build.stale())" > $out/signature.txt 2>&1; cat $out/signature.txt
timeout 3000 python scripts/tail_probe.py --seeds "10,23,24,13,11,21,25,12,22" --natural 4096 > $out/tail_probe_seeds.txt 2>&1; echo "tail probe rc $?"
grep -E "^seed|=>" $out/tail_probe_seeds.txt | cut -c1-300
