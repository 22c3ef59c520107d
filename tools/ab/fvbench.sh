python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fullrank or default_measure or mrr_training" 2>&1 | tail -8
FR_FV_PROFILE=1 python tools/lsbench.py --measure ndcg --reps 3 2>&1 | tail -14
python tools/train_e2e.py --measure ndcg --shape 30k --restarts 32 --max-ticks 136 2>&1 | tail -2
