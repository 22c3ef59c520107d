#!/bin/bash
# Per-instantiation durations of fullrank_verify_kernel on mslr (plain rule) and tiesmix (duplicate-group rule) data.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/fv_dup_prof; mkdir -p $O
for kind in mslr tiesmix; do
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$kind -o p -- python $R/tools/ms_by_tick.py ndcg $kind 6 > $O/$kind.log 2>&1
  f=$(find /tmp/prof_$kind -name '*kernel_stats.csv' | head -1)
  head -30 "$f" | cut -c1-260 > $O/${kind}_kernel_stats.csv
done
