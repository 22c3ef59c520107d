#!/bin/bash
# visiting-order tables on / off (FR_VERIFY_ORDER) per data kind on one box, by the pipelined bench value
cd "$GRAFT_REPO_ROOT"
m() { k=$1; shift; env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --repeats 2 --data $k 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$k $*', 'value', round(d['value']), 'runs', [round(x) for x in d['value_runs']], 'ms/step %.3f' % d['ms_per_step'], 'kernels_ms', {a:round(b,1) for a,b in d['kernels_ms'].items()})"; }
for k in ${KINDS:-mslr hard ties tiesmix hardties}; do
  m $k FR_VERIFY_ORDER=1; m $k FR_VERIFY_ORDER=0; m $k FR_VERIFY_ORDER=1; m $k FR_VERIFY_ORDER=0
done
