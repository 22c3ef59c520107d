#!/bin/bash
cd "$GRAFT_REPO_ROOT"
m() { env $1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', 'value %.0f'%d['value'], 'ms_per_step %.3f'%d['ms_per_step'], 'iso %.3f'%d['roofline']['avg_launch_ms'])"; }
for r in 1 2; do
m "FR_X=0"
m "FR_REDO_GRID=128"
m "FR_REDO_GRID=512"
m "FR_REDO_GRID=128 FR_LS_FIFO=1"
m "FR_REDO_GRID=2048"
done
