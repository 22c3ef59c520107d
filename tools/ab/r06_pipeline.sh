#!/bin/bash
# sets of restarts kept in flight (FR_LS_PIPELINE) in the driver's form of the timed region; one box
cd "$GRAFT_REPO_ROOT"
m() { env $1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-side --no-power --repeats 5 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-20s' % ('$1' or 'default (3)'), 'value', round(d['value']), 'min/median/max', [round(x) for x in d['value_runs_min_median_max']], 'ms/step %.3f iso %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
m ""; m FR_LS_PIPELINE=2; m FR_LS_PIPELINE=4; m ""; m FR_LS_PIPELINE=2; m FR_LS_PIPELINE=4
