# one gpurun call: tiers after the refinement rule, bit-identity of the entry points (incl. the exact latency variants), and the tail probe
python scripts/tier_probe.py 11 13 23 24 25 12 21 22 2>&1 | grep -E "^seed"
python scripts/w2_check.py 11 13 24 12 22 2>&1 | grep "^seed"
python scripts/tail_probe.py --seeds 11,13,23,24,25,12,21,22 2>&1 | grep -v amdgpu.ids > gpurun_out/r04m_tail_probe_units2.txt; grep -E "=>" gpurun_out/r04m_tail_probe_units2.txt
