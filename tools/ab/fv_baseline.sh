#!/bin/bash
# full-ranking line search (depth-less NDCG, MAP): pipelined training rate, per-kernel stats, PMC per size class
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/fv
for m in ndcg map; do python tools/train_e2e.py --measure $m --shape 30k --restarts 32 --max-ticks 272 2>&1 | tail -1; done | tee gpurun_out/fv/train_fullrank_30k.json
( cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fv/stats -o s -- python tools/train_e2e.py --measure ndcg --shape 30k --restarts 32 --max-ticks 136 > gpurun_out/fv/stats.log 2>&1 )
cp gpurun_out/fv/stats/*kernel_stats.csv gpurun_out/fv/kernel_stats.csv 2>/dev/null
head -12 gpurun_out/fv/kernel_stats.csv
FR_LS_PIPELINE=0 bash tools/pmc_kernels.sh gpurun_out/fv/pmc_ndcg "fullrank_verify_kernel" -- python tools/train_e2e.py --measure ndcg --shape 30k --restarts 32 --max-ticks 6 | tail -150
FR_LS_PIPELINE=0 bash tools/pmc_kernels.sh gpurun_out/fv/pmc_map "fullrank_verify_kernel" -- python tools/train_e2e.py --measure map --shape 30k --restarts 32 --max-ticks 6 | tail -5
