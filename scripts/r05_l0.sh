#!/bin/bash
# round 5: layer0_stream_kernel against the tiled launches -- bit identity (n = 301: workgroups with one and with two CUs; n = 4096), then timings.
#   $2 = the environment switch to A/B (default MLT_NO_L0_STREAM; MLT_NO_L0_S5 = the fifth stage alone)
out=gpurun_out/${1:-r05j}
var=${2:-MLT_NO_L0_STREAM}
mkdir -p $out
timeout 600 python scripts/ab_env.py $var 301 > $out/ab301_$var.txt 2>&1; tail -3 $out/ab301_$var.txt | cut -c1-600
timeout 600 python scripts/ab_env.py $var 4096 > $out/ab4096_$var.txt 2>&1; tail -3 $out/ab4096_$var.txt | cut -c1-600
