# one gpurun call: the exact per-conv kernels' patch prefetch depth (sweep_cfg.py variants) on the exact 128 model and on the 64 model,
# and the two-samples-per-tile patch budget of the 128 -> 256 stride-2 layer against the old 64 KiB rule
SWEEP_NAMES=1 SWEEP_FLAGS=1 python scripts/sweep_cfg.py run 2>&1 | grep -v amdgpu | tee gpurun_out/sweep_une_exact128.txt
MLT_TUNING=1 MLT_EXACT_PATCH_64K=1 python bench.py --flags 1 --no-cpu-baseline --steps 20 --warmup 10 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('patch64k', round(d['value']), ' '.join('%.3f'%k['avg_ms'] for k in d['derived']['kernels']))
" | tee -a gpurun_out/sweep_une_exact128.txt
SWEEP_ARGS="--size 64" python scripts/sweep_cfg.py run 2>&1 | grep -v amdgpu | tee gpurun_out/sweep_une_64.txt
SWEEP_ARGS="--weight-seed 13" python scripts/sweep_cfg.py run 2>&1 | grep -v amdgpu | tee gpurun_out/sweep_une_seed13.txt
