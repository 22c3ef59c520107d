#!/bin/bash
# (round 6: the tuning / ablation switches this script sets exist only in a pricing build -- csrc/device.hpp pricing_env)
export FR_BUILD_FLAGS="${FR_BUILD_FLAGS:--DFR_PRICING}"; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tree or forest or ensemble or config5" 2>&1 | tail -2
python bench.py --measure trees --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('trees value', d['value'], 'ms_per_step', d['ms_per_step'], 'kernel ms', d['roofline']['avg_launch_ms'], 'parity', d['parity_first_docs_bit_exact'])"
FR_TREE_CACHE=0 python bench.py --measure trees --steps 20 --warmup 3 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print('nocache value', d['value'], 'ms_per_step', d['ms_per_step'])"
python tools/fuzz_trees.py --iters 300 2>&1 | tail -2
