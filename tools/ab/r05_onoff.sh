#!/bin/bash
# round 5: visiting order on / off on one box by the pipelined bench value (40 steps + 2 repeats), for the given data kinds
cd "$GRAFT_REPO_ROOT"
m() { local kind=$1; shift; env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --repeats 2 --data $kind 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kind $*', 'value', round(d['value']), 'runs', [round(x) for x in d['value_runs']], 'ms/step %.3f iso %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']), 'kernels', {k: round(v,1) for k,v in d['kernels_ms'].items()})"; }
for k in ${@:-mslr hard}; do
  m $k FR_VERIFY_ORDER=1
  m $k FR_VERIFY_ORDER=0
done
