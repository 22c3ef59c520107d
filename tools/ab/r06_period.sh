#!/bin/bash
# line searches between two refreshes of a restart's R ranks (FR_RANK_PERIOD, pricing build), driver's form of the timed region; one box
cd "$GRAFT_REPO_ROOT"
export FR_BUILD_FLAGS=-DFR_PRICING; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
m() { env $2 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-side --no-power --repeats 5 --data $1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-9s %-20s' % ('$1', '$2' or 'default (8)'), 'value', round(d['value']), 'min/median/max', [round(x) for x in d['value_runs_min_median_max']], 'ms/step %.3f' % d['ms_per_step'], 'chain/visit %.3f' % d['verify']['chain_runs_per_visit'])"; }
for k in ${KINDS:-mslr hardties}; do m $k ""; m $k FR_RANK_PERIOD=4; m $k FR_RANK_PERIOD=16; m $k ""; m $k FR_RANK_PERIOD=4; m $k FR_RANK_PERIOD=16; done
