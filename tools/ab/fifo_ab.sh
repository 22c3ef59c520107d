#!/bin/bash
cd "$GRAFT_REPO_ROOT"
m() { FR_LS_FIFO=$1 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('fifo=$1 $2', 'value %.0f'%d['value'], 'ms_per_step %.3f'%d['ms_per_step'], 'iso %.3f'%d['roofline']['avg_launch_ms'], 'e2e %.0f'%d['e2e_evals_per_s'], d['e2e']['model_sha1'][:10])"; }
m 1 a; m 0 a; m 1 b; m 0 b
for mm in ndcg map mrr; do for f in 1 0; do FR_LS_FIFO=$f python tools/train_e2e.py --measure $mm --shape 30k --restarts 32 --max-ticks 272 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('fifo=$f', d['measure'], 'evals/s %.0f' % d['useful_evals_per_s'], d['restarts_sha1'][:10])"; done; done
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pipelined or trajectory or fullrank" 2>&1 | tail -3
