#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tree or forest or ensemble or config5" 2>&1 | tail -2
python bench.py --measure trees --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('trees value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel ms', d['roofline']['avg_launch_ms'], 'parity', d['parity_first_docs_bit_exact'])"
FR_TREE_CACHE=0 python bench.py --measure trees --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('nocache value', d['value'], 'ms_per_step', d['ms_per_step'])"
python tools/fuzz_trees.py --iters 300 2>&1 | tail -2
