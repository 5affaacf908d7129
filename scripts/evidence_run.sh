#!/bin/bash
# GPU box: one evidence run -> gpurun_out/<tag>/ (turned into profiles/<tag>_* by scripts/make_profiles.py on the build box).
#   bash scripts/evidence_run.sh r04k
# 1. the default bench command (the line the driver records)          -> bench_line.json
# 2. rocprofv3 --kernel-trace --stats of the same workload            -> stats/out_kernel_stats.csv
# 3. HBM traffic: separate --pmc FETCH_SIZE / WRITE_SIZE passes       -> fetch/, write/  (MI355X_MICROARCH.md HBM section)
# 4. MFMA / LDS / wait counters, one --pmc pass per counter group     -> pmc/<group>/
# 5. round 4: the other arithmetic tiers (weight seeds 11 / 13 / 23 / 24: hi+lo weights in some stages; 12 / 21 / 22: exact stages; --flags 1: exact), the encoder's
#    configuration (decision guard: the default since ABI 4; --flags 0x20 = without), exact-lite (--flags 0x41), the trained weight families staged under
#    tests/data/_blobs/ (tools/train_synth_weights.py; round 6: with and without the magnitude guard), content mixes (25 % flat-guard content, natural statistics),
#    the small CU sizes, the one-CU timeline, the deferred-batch latency at n = 1 .. 32
# Counter passes never combine --pmc with --stats / trace domains other than --kernel-trace.
set -u
tag=${1:-r04k}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/bench_line.json 2> $out/bench.err
python3 bench.py --no-cpu-baseline --latency --host-staged > $out/bench_extras.json 2>> $out/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o out -- python3 bench.py --no-cpu-baseline > $out/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -o out -- python3 scripts/prof_run.py 4096 2 > $out/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -o out -- python3 scripts/prof_run.py 4096 2 > $out/write.log 2>&1
for grp in "GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_BUSY_CYCLES SQ_INSTS_LDS"; do
  name=$(echo $grp | tr ' ' '+')
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/pmc/$name -o out -- python3 scripts/prof_run.py 4096 2 > $out/pmc_$name.log 2>&1
done
# round 6: every side leg checks ALL 4096 CUs of its batch against the C oracle (--cpu-sample 4096; round 5's legs checked 8) and skips the sustained legs (bench.py's default outside the headline run)
S="--no-cpu-baseline --cpu-sample 4096"
for s in 11 13 23 24 25 12 21 22; do python3 bench.py $S --weight-seed $s > $out/bench_seed$s.json 2>> $out/bench.err; done
python3 bench.py $S --flags 1 --steps 20 --warmup 5 > $out/bench_exact.json 2>> $out/bench.err
python3 bench.py $S --flags 0x41 --steps 20 --warmup 5 > $out/bench_exact_lite.json 2>> $out/bench.err   # round 5: exact-lite forced (tier 5)
python3 bench.py $S --flags 0x20 > $out/bench_no_decision_guard.json 2>> $out/bench.err                     # ABI 4: the guard is the default; this is the round-4 headline configuration
# round 6: the trained weight families (tools/train_synth_weights.py; blobs staged, git-ignored, under tests/data/_blobs/) on the bench content and on natural scenes; the same without the magnitude guard
for f in tests/data/_blobs/*.mltw; do
  [ -f "$f" ] || continue
  b=$(basename $f .mltw)
  python3 bench.py $S --weights-blob $f --steps 20 --warmup 5 > $out/bench_$b.json 2>> $out/bench.err
  python3 bench.py $S --weights-blob $f --steps 20 --warmup 5 --content natural > $out/bench_${b}_natural.json 2>> $out/bench.err
  python3 bench.py $S --weights-blob $f --steps 20 --warmup 5 --flags 0x80 > $out/bench_${b}_no_magnitude_guard.json 2>> $out/bench.err
done
python3 bench.py $S --flat-frac 0.25 --steps 20 --warmup 5 > $out/bench_flat25.json 2>> $out/bench.err
python3 bench.py $S --content natural > $out/bench_natural.json 2>> $out/bench.err
python3 bench.py $S --content natural --flags 0x20 > $out/bench_natural_no_decision_guard.json 2>> $out/bench.err
python3 scripts/flush_latency.py 10 40 > $out/flush_latency.txt 2>&1
for s in 64 32 16; do python3 bench.py --size $s > $out/bench_s$s.json 2>> $out/bench.err; done
rocprofv3 --kernel-trace --output-format csv -d $out/lat -o out -- python3 scripts/latency_run.py 60 10 0 > $out/lat.log 2>&1
python3 scripts/latency_run.py 80 10 4 > $out/latency_modes.txt 2>&1
python3 scripts/latency_run.py 80 13 4 >> $out/latency_modes.txt 2>&1
python3 scripts/latency_run.py 80 22 4 >> $out/latency_modes.txt 2>&1
find $out -name '*.csv' | head -40
cut -c1-600 $out/bench_line.json
