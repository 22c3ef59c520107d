#!/bin/bash
# correctness of the full-ranking paths, then per-class kernel times (lock step) and pipelined training rates
# (round 6: the tuning / ablation switches this script sets exist only in a pricing build -- csrc/device.hpp pricing_env)
export FR_BUILD_FLAGS="${FR_BUILD_FLAGS:--DFR_PRICING}"; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/fv
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fullrank or default_measure or mrr_training or fullrank_verify or views" 2>&1 | tail -15
FR_LS_PIPELINE=0 FR_FV_PROFILE=1 timeout 600 python tools/train_e2e.py --measure ndcg --shape 30k --restarts 32 --max-ticks 40 --profile 2>&1 | tail -32
for m in ndcg map; do timeout 600 python tools/train_e2e.py --measure $m --shape 30k --restarts 32 --max-ticks 272 2>&1 | tail -1; done | tee gpurun_out/fv/train_fullrank_30k_new.json
