#!/bin/bash
# What would a tick cost with the host out of the loop?  FR_LS_REPEAT=n queues every tick's kernels n times back to back
# (idempotent), so (ms_per_step(n) - ms_per_step(1)) / (n - 1) is the device-side cost of one more tick per set with nothing
# waiting for the host -- the floor a device-side replay of the accept logic could reach.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04c
for n in 1 4 1 4; do
  FR_LS_REPEAT=$n python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('repeat $n ms_per_step', round(d['ms_per_step'],4), 'isolated', round(d['roofline']['avg_launch_ms'],4), 'value', round(d['value']))"
done
for n in 1 4; do
  FR_LS_PIPELINE=0 FR_LS_REPEAT=$n python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('lockstep repeat $n ms_per_step', round(d['ms_per_step'],4), 'launch', round(d['roofline']['avg_launch_ms'],4))"
done
