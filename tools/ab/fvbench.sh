#!/bin/bash
# correctness of the full-ranking paths, then per-class kernel times (lock step) and pipelined training rates
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fullrank or default_measure or mrr_training" 2>&1 | tail -4
bash tools/ab/fvprof.sh ndcg | grep "verify_\|evals_per_s"
for m in ndcg map; do python tools/train_e2e.py --measure $m --shape 30k --restarts 32 --max-ticks 136 2>&1 | tail -1; done
