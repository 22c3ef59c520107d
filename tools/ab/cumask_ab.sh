#!/bin/bash
# First-in-first-out verify stream restricted to all but n compute units per XCD (FR_LS_CUMASK=n): do the tails of a finished
# set then overlap the next set's verify kernel?
cd "$GRAFT_REPO_ROOT"
m() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', 'ms_per_step', round(d['ms_per_step'],4), 'isolated', round(d['roofline']['avg_launch_ms'],4), 'value', round(d['value']), {k: round(v,1) for k,v in d['kernels_ms'].items()})"; }
m base
FR_LS_FIFO=1 m fifo
for n in 1 2 4; do
  FR_LS_CUMASK=$n m "cumask$n"
  FR_LS_CUMASK=$n FR_LS_PIPELINE=2 m "cumask$n+2sets"
  FR_LS_CUMASK=$n FR_LS_PIPELINE=4 m "cumask$n+4sets"
done
m base
