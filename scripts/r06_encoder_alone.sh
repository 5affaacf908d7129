#!/bin/bash
# round 6: the encoder's own call, alone on the GPU -- tools/run_ra_eval.py --mode alone: the GPU-using legs one after the other, anchors beside them
# on other cores.  832 x 480 x 17 frames at QP 32 (round 5's clip) and 1920 x 1080 x 9 frames at QP 37 (the shape of the reference's script_128/BasketballDrive_enc_50.sh:
# 135 CTUs of 128 x 128 per picture, WPP diagonals of up to 8 CTUs)
out=gpurun_out/${1:-r06f}
mkdir -p $out
timeout 2400 python tools/run_ra_eval.py --mode alone --out $out/alone_480p --qps 32 > $out/alone_480p.log 2>&1; echo "480p rc $?"
tail -12 $out/alone_480p.log | cut -c1-330
timeout 3000 python tools/run_ra_eval.py --mode alone --out $out/alone_1080p --width 1920 --height 1080 --frames 9 --qps 37 > $out/alone_1080p.log 2>&1; echo "1080p rc $?"
tail -12 $out/alone_1080p.log | cut -c1-330
rm -rf $out/alone_*/torch_model $out/alone_*/*/*.bin
du -sh $out
