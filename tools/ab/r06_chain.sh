#!/bin/bash
# Device counter of insertion-chain runs per (document, group) visit of the verify kernel, with the visiting-order tables
# (default) and in storage order (FR_VERIFY_ORDER=0), per data kind; one box.
cd "$GRAFT_REPO_ROOT"
IFS=";" read -ra ENVS <<< "${ENVLIST:-;FR_VERIFY_ORDER=0}"
m() { env $2 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --no-side --no-power --repeats 1 --data $1 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['verify']; print('%-9s %-20s' % ('$1', '$2' or 'default'), 'value', round(d['value']), 'runs', [round(x) for x in d['value_runs']], 'ms/step %.3f iso %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']), 'chain runs per visit %.4f' % v['chain_runs_per_visit'], 'redo %.2e' % (v['redo_fraction'] or 0), 'rslot ms/step %.3f' % (d['kernels_ms'].get('rslot_kernel', 0.0) / d['instrumented_steps']), 'ranked on/off', v.get('rank_slots_on_off'))"; }
for k in ${KINDS:-mslr hard hardties}; do for e in "${ENVS[@]:-}"; do m $k "$e"; done; done
