SWEEP_NAMES=1 SWEEP_FLAGS=1 python scripts/sweep_cfg.py run 2>&1 | grep -v amdgpu
SWEEP_ARGS="--size 64" python scripts/sweep_cfg.py run 2>&1 | grep -v amdgpu
SWEEP_FLAGS=1 python scripts/sweep_cfg.py run 2>&1 | grep -v amdgpu
