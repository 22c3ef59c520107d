#!/bin/bash
# The trainer's randomised parity soak on the current sources -> gpurun_out/r06_fuzz_trainer.txt (device trainer vs oracle, bit for bit).
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r06_fuzz_trainer.txt
{
  echo "# tools/fuzz_parity.py (device trainer vs oracle, bit for bit), final round-6 sources"
  timeout 400 python tools/fuzz_parity.py --iters 1500 2>&1 | tail -1
  echo "# full-ranking measures only (depth-less NDCG, MAP, NDCG@25/30/60: plain and duplicate-group instantiations), query lengths spread over all size classes (--long)"
  timeout 500 python tools/fuzz_parity.py --iters 700 --measures ndcg,map,ndcg@30,ndcg,map,ndcg@25,ndcg@60 --long 2>&1 | tail -1
  echo "# NDCG@k only (the verify kernel's variants and the exact kernel's redo launches)"
  timeout 400 python tools/fuzz_parity.py --iters 1500 --measures ndcg@1,ndcg@3,ndcg@5,ndcg@10,ndcg@20,ndcg@10 2>&1 | tail -1
  echo "# the same with FR_LS_EXACT=1 (the exact kernel alone: narrow and wide instantiations), long queries"
  FR_LS_EXACT=1 timeout 300 python tools/fuzz_parity.py --iters 300 --measures ndcg@3,ndcg@5,ndcg@10,ndcg@20 --long 2>&1 | tail -1
} > $O 2>&1
cat $O
