#!/bin/bash
# Does a tick's tail overlap other sets' verify kernels when nothing waits for the host (FR_LS_REPEAT) AND the sets are out of
# step (unequal shares, FR_LS_SPLIT)?  (t(4) - t(1)) / 3 per configuration.
cd "$GRAFT_REPO_ROOT"
m() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', 'ms_per_step', round(d['ms_per_step'],4), 'isolated', round(d['roofline']['avg_launch_ms'],4))"; }
for split in "" "14,11,7" "16,10,6" "20,8,4"; do
  for n in 1 4; do FR_LS_SPLIT=$split FR_LS_REPEAT=$n m "split[$split] repeat$n"; done
done
for split in "20,12" "24,8"; do
  for n in 1 4; do FR_LS_PIPELINE=2 FR_LS_SPLIT=$split FR_LS_REPEAT=$n m "2sets split[$split] repeat$n"; done
done
