#!/bin/bash
# A/B of two source states on ONE box by the pipelined bench value: B = the tree, A = the files under tools/ab/old/ (git-ignored;
# put the other variant of any csrc file there).  Alternates B A B A.
cd "$GRAFT_REPO_ROOT"; mkdir -p /tmp/cur
m() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --repeats 2 --data ${KIND:-mslr} --measure ${MEASURE:-ndcg@10} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'value', round(d['value']), 'runs', [round(x) for x in d['value_runs']], 'ms/step %.3f iso %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
swap_in() { for f in tools/ab/old/*; do b=$(basename $f); cp fastrank_amd/csrc/$b /tmp/cur/$b; cp $f fastrank_amd/csrc/$b; done; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1; }
swap_out() { for f in tools/ab/old/*; do b=$(basename $f); cp /tmp/cur/$b fastrank_amd/csrc/$b; done; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1; }
m B1; swap_in; m A1; swap_out; m B2; swap_in; m A2; swap_out
