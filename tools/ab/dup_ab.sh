#!/bin/bash
cd "$GRAFT_REPO_ROOT"
for w in 8 6 5; do
  FR_BUILD_FLAGS="-DVERIFY_DUP_WAVES=$w" python -c "from fastrank_amd import _build; _build.build()" || exit 1
  for r in 1 2; do FR_BUILD_FLAGS="-DVERIFY_DUP_WAVES=$w" python bench.py --steps 30 --warmup 5 --data tiesmix --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('dup_waves=$w', 'value %.0f'%d['value'], 'ms_per_step %.3f'%d['ms_per_step'], 'iso %.3f'%d['roofline']['avg_launch_ms'], 'redo', d['verify']['redo_fraction'])"; done
done
