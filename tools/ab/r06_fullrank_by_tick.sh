#!/bin/bash
# The full-ranking measures tick by tick on the data kinds, with and without the duplicate-group rule (tools/ms_by_tick.py).
cd "$GRAFT_REPO_ROOT"; TAG=${1:-r06}
{
  echo "# tools/ms_by_tick.py (30K shape, 32 restarts, single-stepped ticks from the start of a job): the full-ranking measures by data kind"
  for mk in "ndcg mslr" "map mslr" "ndcg@30 mslr" "ndcg ties" "ndcg tiesmix" "map tiesmix" "ndcg@30 tiesmix" "ndcg hardties" "map hardties"; do
    python tools/ms_by_tick.py $mk 10 2>/dev/null | grep -v kernels | cut -c1-220
  done
  echo "# the same with FR_NO_DUP_GROUPS=1 (no duplicate-group ids in the keys: the rule as it was before this round's DUP instantiations)"
  for mk in "ndcg tiesmix" "ndcg@30 tiesmix"; do
    FR_NO_DUP_GROUPS=1 python tools/ms_by_tick.py $mk 10 2>/dev/null | grep -v kernels | cut -c1-220
  done
} > gpurun_out/${TAG}_fullrank_by_tick.txt; cat gpurun_out/${TAG}_fullrank_by_tick.txt
