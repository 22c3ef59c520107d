#!/bin/bash
# round 5: sets of restarts in flight (FR_LS_PIPELINE) on one box, by the pipelined bench value
# (round 6: the tuning / ablation switches this script sets exist only in a pricing build -- csrc/device.hpp pricing_env)
export FR_BUILD_FLAGS="${FR_BUILD_FLAGS:--DFR_PRICING}"; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
m() { env "$@" python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --repeats 2 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', 'value', round(d['value']), 'runs', [round(x) for x in d['value_runs']], 'ms/step %.3f iso %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
m FR_LS_PIPELINE=3
m FR_LS_PIPELINE=2
m FR_LS_PIPELINE=4
m FR_LS_PIPELINE=3 FR_RANK_PERIOD=16
m FR_LS_PIPELINE=3 FR_VERIFY_ORDER=0
