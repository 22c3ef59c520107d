#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/fuzz; TAG=${1:-r05}
{
echo "# tools/fuzz_parity.py (device trainer vs oracle, bit for bit), round-6 sources"
python tools/fuzz_parity.py --iters 1500 --seed 501 2>&1 | tail -2
echo "# full-ranking measures only, query lengths spread over all size classes (--long)"
python tools/fuzz_parity.py --iters 800 --seed 502 --measures ndcg,map,ndcg@30,ndcg@100,ndcg,map --long 2>&1 | tail -2
echo "# NDCG@k only (the verify kernel's variants), and the same under FR_LS_PIPELINE=0 / FR_VERIFY_ORDER=0"
python tools/fuzz_parity.py --iters 1500 --seed 503 --measures ndcg@10,ndcg@5,ndcg@20,ndcg@3,ndcg@1 2>&1 | tail -1
FR_LS_PIPELINE=0 python tools/fuzz_parity.py --iters 300 --seed 504 --long 2>&1 | tail -1
FR_VERIFY_ORDER=0 python tools/fuzz_parity.py --iters 300 --seed 505 2>&1 | tail -1
echo "# tools/fuzz_rf.py (random-forest training vs oracle)"
python tools/fuzz_rf.py --iters 800 --seed 506 2>&1 | tail -2
echo "# tools/fuzz_trees.py (forest scoring vs oracle)"
python tools/fuzz_trees.py --iters 300 2>&1 | tail -1
} | tee gpurun_out/fuzz/${TAG}_fuzz.txt
