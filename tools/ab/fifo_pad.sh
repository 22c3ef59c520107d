#!/bin/bash
# FIFO verify stream x occupancy cap of the verify kernel (unused dynamic LDS): do the tails of a set get wave slots while
# the next set's verify grid is dispatching?
cd "$GRAFT_REPO_ROOT"
m() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', 'ms_per_step', round(d['ms_per_step'],4), 'isolated', round(d['roofline']['avg_launch_ms'],4), 'value', round(d['value']), {k: round(v,1) for k,v in d['kernels_ms'].items()})"; }
m base
FR_LS_FIFO=1 m fifo
for pad in 5000 5700 6400 7300; do
  FR_VERIFY_LDS_PAD=$pad m "pad$pad"
  FR_LS_FIFO=1 FR_VERIFY_LDS_PAD=$pad m "fifo+pad$pad"
done
FR_LS_FIFO=1 FR_VERIFY_LDS_PAD=5700 FR_LS_PIPELINE=2 m "fifo+pad5700+2sets"
FR_LS_FIFO=1 FR_VERIFY_LDS_PAD=5700 FR_LS_PIPELINE=4 m "fifo+pad5700+4sets"
m base
