#!/bin/bash
# round 5: the two constants of the visiting order on one box (pipelined bench value, 40 steps + 2 repeats, and the whole job)
# (round 6: the tuning / ablation switches this script sets exist only in a pricing build -- csrc/device.hpp pricing_env)
export FR_BUILD_FLAGS="${FR_BUILD_FLAGS:--DFR_PRICING}"; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
m() { local kind=$1; shift; env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --repeats 2 --data $kind 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$kind $*', 'value', round(d['value']), 'runs', [round(x) for x in d['value_runs']], 'ms/step %.3f' % d['ms_per_step'], 'e2e %.0f evals/s %.3f s' % (d['e2e']['e2e_evals_per_s'], d['e2e']['wall_s']))"; }
for k in mslr hard; do
  m $k A=0
  m $k FR_RANK_PERIOD=4
  m $k FR_RANK_PERIOD=16
  m $k FR_ORDER_KAPPA=0.5
  m $k FR_ORDER_KAPPA=2
done
