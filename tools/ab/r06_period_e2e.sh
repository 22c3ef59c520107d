#!/bin/bash
# the same sweep over a 40-step region and whole jobs (bench.py's e2e leg); pricing build, one box
cd "$GRAFT_REPO_ROOT"
export FR_BUILD_FLAGS=-DFR_PRICING; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
m() { env $2 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-side --no-power --repeats 1 --data $1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['e2e']; print('%-9s %-20s' % ('$1', '$2' or 'default (8)'), 'timed(40)', [round(x) for x in d['value_runs']], 'e2e evals/s', round(e['e2e_evals_per_s']), 'wall %.3f s' % e['wall_s'])"; }
for k in ${KINDS:-mslr hard hardties}; do m $k ""; m $k FR_RANK_PERIOD=16; m $k ""; m $k FR_RANK_PERIOD=16; done
