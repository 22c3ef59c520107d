#!/bin/bash
cd "$GRAFT_REPO_ROOT"
m() { FR_REDO_SLICES=$1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('slices=$1', 'value %.0f'%d['value'], 'ms_per_step %.3f'%d['ms_per_step'], 'iso %.3f'%d['roofline']['avg_launch_ms'], 'e2e %.0f'%d['e2e_evals_per_s'], d['e2e']['model_sha1'][:10], 'exact ms/step %.3f'%d['verify']['exact_kernel_ms_per_step'])"; }
m 1; m 0; m 1; m 0; m 1; m 0
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
FR_REDO_GRID=64 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "verify or ties or dup" 2>&1 | tail -3
