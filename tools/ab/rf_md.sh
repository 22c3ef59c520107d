#!/bin/bash
# random-forest fan-out + redo-wave A/B on one box
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/rfmd
python -m pytest tests/test_gpu_multidevice.py tests/test_gpu_rf_train.py -x -q -m gpu 2>&1 | tail -4
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fullrank or full_rank or map or ndcg" 2>&1 | tail -3
for f in "" "-DFV_REDO_WAVES=4"; do
  echo "=== [$f]"
  FR_BUILD_FLAGS="$f" python -c "from fastrank_amd import _build; _build.build()" || exit 1
  for m in ndcg map; do for i in 1 2; do FR_BUILD_FLAGS="$f" timeout 600 python tools/train_e2e.py --measure $m --shape 30k --restarts 32 --max-ticks 272 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(d['measure'], 'evals/s %.0f' % d['useful_evals_per_s'], 'wall %.3f' % d['train_wall_s'], d['restarts_sha1'][:10])"; done; done
done
