#!/bin/bash
# FR_VERIFY_AUDIT=1: every value the bound-and-verify NDCG@k line search publishes is recomputed by the exact kernel and
# compared bit for bit, over whole training runs on the five kinds of data (resident sums never refreshed)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/audit; TAG=${1:-r05}
for kind in mslr hard ties tiesmix hardties; do
  FR_VERIFY_AUDIT=1 FR_RESIDENT_REFRESH=100000 python tools/train_e2e.py --shape 30k --restarts 32 --data $kind --max-ticks 500 2>&1 | tail -1
done | tee gpurun_out/audit/${TAG}_audit_30k.json
