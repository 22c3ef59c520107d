#!/bin/bash
# random-forest training: parity tests, then wall / kernel times; FR_RF_PARTITION=0: children's segments by rekey + radix sort
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/rf
python -m pytest tests/test_gpu_rf_train.py tests/test_gpu_multidevice.py -x -q -m gpu 2>&1 | tail -3
python tools/fuzz_rf.py --iters 100 2>&1 | tail -2
run() { python tools/rfbench.py --shape 30k --trees 100 --cpu-trees 1 --check 2>/dev/null | tail -1 | tee gpurun_out/rf/rfbench_30k_$1.json | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('$1: trees/s %.2f wall %.2f' % (d['value'], d['wall_s']), d['kernels_ms'], d.get('first_trees_identical_to_oracle'))"; }
run default
FR_RF_PARTITION=0 run sort
python tools/rfbench.py --shape 10k --trees 30 --split-candidates 32 --cpu-trees 1 --check 2>/dev/null | tail -1 | tee gpurun_out/rf/rfbench_10k_k32.json | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('k32: trees/s %.2f wall %.2f' % (d['value'], d['wall_s']), d['kernels_ms'], d.get('first_trees_identical_to_oracle'))"
