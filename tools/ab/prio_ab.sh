#!/bin/bash
# (record of a withdrawn experiment: the FR_LS_PRIO switch it drives was removed again after this A/B, DESIGN.md section 4.9 (round 3: git history))
# stream priorities for the tick's tail: FR_LS_FIFO x FR_LS_PRIO on one box
cd "$GRAFT_REPO_ROOT"
m() { FR_LS_FIFO=$1 FR_LS_PRIO=$2 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('fifo=$1 prio=$2', 'value %.0f'%d['value'], 'ms_per_step %.3f'%d['ms_per_step'], 'iso %.3f'%d['roofline']['avg_launch_ms'], 'e2e %.0f'%d['e2e_evals_per_s'], d['e2e']['model_sha1'][:10])"; }
for r in a b; do m 0 0; m 1 1; m 0 1; m 1 0; done
for mm in ndcg map; do for c in "0 0" "1 1" "0 1"; do set -- $c; FR_LS_FIFO=$1 FR_LS_PRIO=$2 python tools/train_e2e.py --measure $mm --shape 30k --restarts 32 --max-ticks 272 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('fifo=$1 prio=$2', d['measure'], 'evals/s %.0f' % d['useful_evals_per_s'], d['restarts_sha1'][:10])"; done; done
