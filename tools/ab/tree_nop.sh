#!/bin/bash
# A/B: the s_nop between v_cmp and the SDWA add-with-carry of a tree step (is it needed? tests decide; is it slower?)
run() { python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "tree" 2>&1 | tail -1; for i in 1 2; do python tools/treebench.py --reps 3 --check 20000 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1 kernel_ms', round(d['kernel_avg_ms'],3), 'parity', d['parity_first_docs_bit_exact'])"; done; }
run no_nop
FR_BUILD_FLAGS="-DTREE_WITH_NOP" python -c "from fastrank_amd import _build; _build.build(force=True)"
run with_nop
