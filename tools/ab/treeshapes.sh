#!/bin/bash
# tree-ensemble kernel: parity tests, then the 500-tree pass at the 30K shape for block shapes x walks per thread
# (round 6: the tuning / ablation switches this script sets exist only in a pricing build -- csrc/device.hpp pricing_env)
export FR_BUILD_FLAGS="${FR_BUILD_FLAGS:--DFR_PRICING}"; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -x -q -m gpu -k "tree" 2>&1 | tail -3
for s in ${SHAPES:-256,2 192,4 192,2 128,4}; do for n in ${NS:-16 8 4 2}; do
  FR_TREE_NMAX=$n FR_TREE_SHAPE=$s python tools/treebench.py --reps 3 --check 2000 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('shape=%-6s nmax=%-2s' % ('$s', '$n'), 'kernel_ms', round(d['kernel_avg_ms'],3), 'parity', d['parity_first_docs_bit_exact'])"
done; done
