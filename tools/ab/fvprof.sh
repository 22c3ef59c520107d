#!/bin/bash
# per-kernel / per-size-class HIP-event times of full-ranking training in lock step (resident sums)
# (round 6: the tuning / ablation switches this script sets exist only in a pricing build -- csrc/device.hpp pricing_env)
export FR_BUILD_FLAGS="${FR_BUILD_FLAGS:--DFR_PRICING}"; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
export FR_LS_PIPELINE=0 FR_FV_PROFILE=1
python tools/train_e2e.py --measure ${1:-ndcg} --shape 30k --restarts 32 --max-ticks 40 --profile 2>&1 | tail -22
