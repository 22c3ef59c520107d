#!/bin/bash
# A/B of full-ranking kernel builds on one box: tools/ab/fv_ab.sh "<flags A>" "<flags B>" ...   ("" = as committed)
# (round 6: the tuning / ablation switches this script sets exist only in a pricing build -- csrc/device.hpp pricing_env)
export FR_BUILD_FLAGS="${FR_BUILD_FLAGS:--DFR_PRICING}"; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/fv
run() {
  echo "=== build flags: [$1]"
  FR_BUILD_FLAGS="$1" python -c "from fastrank_amd import _build; _build.build()" || exit 1
  FR_BUILD_FLAGS="$1" FR_LS_PIPELINE=0 FR_FV_PROFILE=1 timeout 600 python tools/train_e2e.py --measure ndcg --shape 30k --restarts 32 --max-ticks 40 --profile 2>&1 | grep "fullrank_verify\|rank_metric\|scores_kernel"
  for m in ndcg map; do FR_BUILD_FLAGS="$1" timeout 600 python tools/train_e2e.py --measure $m --shape 30k --restarts 32 --max-ticks 272 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.readline()); print(d['measure'], 'evals/s %.0f' % d['useful_evals_per_s'], 'wall %.3f' % d['train_wall_s'], d['restarts_sha1'][:10])"; done
}
for f in "$@"; do run "$f"; done
