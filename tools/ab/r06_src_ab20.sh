#!/bin/bash
# A/B of two source states on ONE box in the DRIVER's form of the timed region (--steps 20 --warmup 5: ticks 5..24 of a job, five
# repeats on fresh jobs): B = the tree, A = the files under tools/ab/old/.  Alternates B A B A.
cd "$GRAFT_REPO_ROOT"; mkdir -p /tmp/cur
m() { python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-side --no-power --repeats 5 --data ${KIND:-mslr} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value', round(d['value']), 'min/median/max', [round(x) for x in d['value_runs_min_median_max']], 'ms/step %.3f iso %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']), 'chain/visit %.3f' % d['verify']['chain_runs_per_visit'])"; }
swap_in() { for f in tools/ab/old/*; do b=$(basename $f); cp fastrank_amd/csrc/$b /tmp/cur/$b; cp $f fastrank_amd/csrc/$b; done; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1; }
swap_out() { for f in tools/ab/old/*; do b=$(basename $f); cp /tmp/cur/$b fastrank_amd/csrc/$b; done; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1; }
m B1; swap_in; m A1; swap_out; m B2; swap_in; m A2; swap_out
