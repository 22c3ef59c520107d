#!/bin/bash
# Time-to-model (bench.py's e2e leg: 32 restarts to convergence) and the timed steps under the R-rank upkeep policy (default),
# with the ranks always refreshed (pricing build, FR_RANK_OFF_BELOW=0) and in storage order (FR_VERIFY_ORDER=0); one box.
cd "$GRAFT_REPO_ROOT"
export FR_BUILD_FLAGS=-DFR_PRICING; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
m() { env $2 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-side --no-power --repeats 1 --data $1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['verify']; e=d['e2e']; print('%-9s %-22s' % ('$1', '$2' or 'default'), 'timed', round(d['value']), 'e2e evals/s', round(e['e2e_evals_per_s']), 'wall %.3f s ticks %d' % (e['wall_s'], e['per_rank_ticks'][0]), 'oracle_check', e['oracle_check']['ok'], 'chain/visit %.4f' % v['chain_runs_per_visit'], 'switched on/off', v.get('rank_slots_on_off'))"; }
for k in ${KINDS:-mslr hardties}; do m $k ""; m $k FR_RANK_OFF_BELOW=0; m $k FR_VERIFY_ORDER=0; m $k ""; m $k FR_RANK_OFF_BELOW=0; m $k FR_VERIFY_ORDER=0; done
