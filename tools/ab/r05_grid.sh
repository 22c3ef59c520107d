#!/bin/bash
cd "$GRAFT_REPO_ROOT"
m() { env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --repeats 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'value', round(d['value']), 'runs', [round(x) for x in d['value_runs']], 'ms/step %.3f iso %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
m A=1; m FR_REDO_GRID=128; m FR_REDO_GRID=64; m A=1; m FR_REDO_GRID=128
