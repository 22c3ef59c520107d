#!/bin/bash
# A/B of a compile-time switch on ONE box by the pipelined bench value: B = the tree as built, A = rebuilt with FR_BUILD_FLAGS="$FLAGS".
# usage: FLAGS=-DVERIFY_PREFETCH [KIND=hard] tools/ab/r05_flags_ab.sh      (alternates B A B A, leaves the tree's own build behind)
cd "$GRAFT_REPO_ROOT"
m() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --repeats 2 --data ${KIND:-mslr} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value', round(d['value']), 'runs', [round(x) for x in d['value_runs']], 'ms/step %.3f iso %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
b() { FR_BUILD_FLAGS="$1" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1; }
m B1; export FR_BUILD_FLAGS="$FLAGS"; b "$FLAGS"; m A1; unset FR_BUILD_FLAGS; b ""; m B2; export FR_BUILD_FLAGS="$FLAGS"; b "$FLAGS"; m A2; unset FR_BUILD_FLAGS; b ""
