#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/fv
FR_LS_PIPELINE=0 bash tools/pmc_kernels.sh gpurun_out/fv/pmc_ndcg_after "fullrank_verify_kernel" -- python tools/train_e2e.py --measure ndcg --shape 30k --restarts 32 --max-ticks 6 > gpurun_out/fv/pmc_ndcg_after.log 2>&1
grep -E "^fullrank|derived: VALU|SQ_INSTS_VALU |SQ_INSTS_LDS |SQ_WAIT_INST_LDS|SQ_ACTIVE_INST_LDS|SQ_LDS_BANK|SQ_LDS_IDX|SQ_WAVE_CYCLES|SQ_WAIT_ANY |SQ_WAIT_INST_ANY|SQ_BUSY_CYCLES|ICACHE|IFETCH|instruction cache|shader clock" gpurun_out/fv/pmc_ndcg_after/summary.txt
