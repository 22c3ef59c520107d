#!/bin/bash
# What the duplicate-group slow path of fullrank_verify_kernel costs: ms per tick on tiesmix data with parts of it compiled out (wrong answers, timing only).
cd "$GRAFT_REPO_ROOT"
b() { FR_BUILD_FLAGS="$1" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1; }
for F in "" "-DFVX_NOSCAN"; do
  export FR_BUILD_FLAGS="$F"; b "$F"
  echo "== flags '$F'"; python tools/ms_by_tick.py ndcg tiesmix 4 2>/dev/null | cut -c1-300
done
unset FR_BUILD_FLAGS; b ""
