#!/bin/bash
# per-kernel HIP-event durations of a lock-step tick (nothing overlaps): what the tail of a tick costs in isolation
cd "$GRAFT_REPO_ROOT"
FR_LS_PIPELINE=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); n=d['instrumented_steps']
print('lockstep ms_per_step', round(d['ms_per_step'],4), 'per step:', {k: round(v/n,4) for k,v in d['kernels_ms'].items()}, 'sum', round(sum(d['kernels_ms'].values())/n,4))"
