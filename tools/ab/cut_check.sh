#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python tools/fuzz_parity.py --iters 300 --seed 11 --measures ndcg@21,ndcg@30,ndcg@50,ndcg@100,ndcg@64 --long 2>&1 | tail -3
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fullrank or depth" 2>&1 | tail -2
for m in ndcg@50 ndcg@100 ndcg; do python tools/train_e2e.py --measure $m --shape 30k --restarts 32 --max-ticks 136 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(d['measure'], 'evals/s %.0f' % d['useful_evals_per_s'], d['restarts_sha1'][:10], d['redo_fraction'])"; done
