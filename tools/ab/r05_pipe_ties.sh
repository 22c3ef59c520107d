#!/bin/bash
cd "$GRAFT_REPO_ROOT"
m() { local kind=$1; shift; env "$@" python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-e2e --repeats 1 --data $kind 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kind $*', 'value', round(d['value']), 'runs', [round(x) for x in d['value_runs']], 'ms/step %.3f' % d['ms_per_step'])"; }
for k in ties hardties; do m $k FR_LS_PIPELINE=3; m $k FR_LS_PIPELINE=4; m $k FR_LS_PIPELINE=2; done
