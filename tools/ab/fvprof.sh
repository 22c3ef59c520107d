#!/bin/bash
# per-kernel / per-size-class HIP-event times of full-ranking training in lock step (resident sums)
export FR_LS_PIPELINE=0 FR_FV_PROFILE=1
python tools/train_e2e.py --measure ${1:-ndcg} --shape 30k --restarts 32 --max-ticks 40 --profile 2>&1 | tail -22
