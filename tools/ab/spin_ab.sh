#!/bin/bash
# (record of a withdrawn experiment: the FR_SPIN_WAIT_US switch it drives was removed again after this A/B, DESIGN.md section 4.9 (round 3: git history))
cd "$GRAFT_REPO_ROOT"
m() { env $1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', 'value %.0f'%d['value'], 'ms_per_step %.3f'%d['ms_per_step'], 'iso %.3f'%d['roofline']['avg_launch_ms'], 'e2e %.0f'%d['e2e_evals_per_s'])"; }
for r in 1 2 3; do
m "FR_SPIN_WAIT_US=2000"
m "FR_SPIN_WAIT_US=0"
done
for f in 2000 0; do FR_SPIN_WAIT_US=$f python tools/train_e2e.py --measure ndcg --shape 30k --restarts 32 --max-ticks 272 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('spin=$f', d['measure'], 'evals/s %.0f' % d['useful_evals_per_s'])"; done
