#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/fuzz
{
echo "# tools/fuzz_parity.py (device trainer vs oracle, bit for bit), round-4 sources"
python tools/fuzz_parity.py --iters 1500 --seed 301 2>&1 | tail -2
echo "# full-ranking measures only, query lengths spread over all size classes (--long)"
python tools/fuzz_parity.py --iters 1200 --seed 302 --measures ndcg,map,ndcg@30,ndcg@100,ndcg,map --long 2>&1 | tail -2
echo "# the same under FR_LS_PIPELINE=0 and FR_LS_FIFO=1"
FR_LS_PIPELINE=0 python tools/fuzz_parity.py --iters 300 --seed 303 --long 2>&1 | tail -1
FR_LS_FIFO=1 python tools/fuzz_parity.py --iters 300 --seed 304 --long 2>&1 | tail -1
echo "# tools/fuzz_rf.py (random-forest training vs oracle)"
python tools/fuzz_rf.py --iters 1500 --seed 305 2>&1 | tail -2
echo "# tools/fuzz_trees.py (forest scoring vs oracle)"
python tools/fuzz_trees.py --iters 300 2>&1 | tail -1
} | tee gpurun_out/fuzz/r04_fuzz.txt
