#!/bin/bash
# GPU box: one evidence run -> gpurun_out/<tag>/ (turned into profiles/<tag>_* by scripts/make_profiles.py on the build box).
#   bash scripts/evidence_run.sh r02a
# 1. the default bench command (the line the driver records)          -> bench_line.json
# 2. rocprofv3 --kernel-trace --stats of the same workload            -> stats/out_kernel_stats.csv
# 3. HBM traffic: separate --pmc FETCH_SIZE / WRITE_SIZE passes       -> fetch/, write/  (MI355X_MICROARCH.md HBM section)
# 4. MFMA / LDS / wait counters, one --pmc pass per counter group     -> pmc/<group>/
# Counter passes never combine --pmc with --stats / trace domains other than --kernel-trace.
set -u
tag=${1:-r02a}
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
find $out -name '*.csv' | head -40
cut -c1-600 $out/bench_line.json
