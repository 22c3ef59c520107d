#!/bin/bash
# FIFO verify stream x size of the redo kernel's fixed grid: with the verify kernels in one queue, a finished set's tail
# competes for wave slots with the next set's verify grid -- a 32 768-block redo grid (8192 pairs x 4 slices) of mostly empty
# blocks then takes a millisecond to dispatch
cd "$GRAFT_REPO_ROOT"
m() { python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', 'ms_per_step', round(d['ms_per_step'],4), 'isolated', round(d['roofline']['avg_launch_ms'],4), 'value', round(d['value']), {k: round(v,1) for k,v in d['kernels_ms'].items()})"; }
m base
for grid in 2048 512 128 64; do
  FR_REDO_GRID=$grid m "grid$grid"
  FR_LS_FIFO=1 FR_REDO_GRID=$grid m "fifo+grid$grid"
done
FR_LS_FIFO=1 FR_REDO_GRID=128 FR_LS_PIPELINE=2 m "fifo+grid128+2sets"
FR_LS_FIFO=1 FR_REDO_GRID=128 FR_LS_PIPELINE=4 m "fifo+grid128+4sets"
FR_LS_FIFO=1 m fifo
m base
