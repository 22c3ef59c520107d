#!/bin/bash
# PMC breakdown of one size class of fullrank_verify_kernel inside the trainer (resident sums, lock step)
export FR_LS_PIPELINE=0
bash tools/pmc_wave.sh gpurun_out/fv_pmc "${1:-fullrank_verify_kernel<64, 1, 0>}" -- python tools/train_e2e.py --measure ndcg --shape 30k --restarts 32 --max-ticks 6
