#!/bin/bash
# PMC breakdown of one size class of fullrank_verify_kernel inside the trainer (resident sums, lock step)
export FR_LS_PIPELINE=0
K="${1:-fullrank_verify_kernel<64, 2, 0>}"
bash tools/pmc_wave.sh gpurun_out/fv_pmc_wave "$K" -- python tools/train_e2e.py --measure ndcg --shape 30k --restarts 32 --max-ticks 6 | tail -20
bash tools/pmc_cmd.sh gpurun_out/fv_pmc_cmd "$K" -- python tools/train_e2e.py --measure ndcg --shape 30k --restarts 32 --max-ticks 6 | tail -24
