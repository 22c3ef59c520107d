#!/bin/bash
cd "$GRAFT_REPO_ROOT"
m() { FR_BENCH_PROFILE_TIMED=$1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('profile_timed=$1', 'value %.0f'%d['value'], 'ms_per_step %.3f'%d['ms_per_step'], 'iso %.3f'%d['roofline']['avg_launch_ms'], 'launches/step', d['config']['launches_per_step'])"; }
m 0; m 1; m 0; m 1; m 0; m 1
