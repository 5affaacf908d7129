#!/bin/bash
# round 5: layer0_stream_kernel against the two-launch form -- bit identity (n = 301: workgroups with one and with two CUs; n = 4096), then timings
out=gpurun_out/${1:-r05j}
mkdir -p $out
timeout 600 python scripts/ab_env.py MLT_NO_L0_STREAM 301 > $out/ab301.txt 2>&1; tail -3 $out/ab301.txt | cut -c1-600
timeout 600 python scripts/ab_env.py MLT_NO_L0_STREAM 4096 > $out/ab4096.txt 2>&1; tail -3 $out/ab4096.txt | cut -c1-600
