#!/bin/bash
# waves per SIMD of the resident verify variants (compile-time), per data kind / measure, one box: B = the tree, A = FLAGS
cd "$GRAFT_REPO_ROOT"
m() { python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e --repeats 2 --data ${KIND:-mslr} --measure ${MEASURE:-ndcg@10} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${KIND:-mslr} ${MEASURE:-ndcg@10} $1', 'value', round(d['value']), 'runs', [round(x) for x in d['value_runs']], 'ms/step %.3f iso %.3f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"; }
b() { FR_BUILD_FLAGS="$1" python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1; }
m "B"; export FR_BUILD_FLAGS="$FLAGS"; b "$FLAGS"; m "A($FLAGS)"; unset FR_BUILD_FLAGS; b ""; m "B"; export FR_BUILD_FLAGS="$FLAGS"; b "$FLAGS"; m "A($FLAGS)"; unset FR_BUILD_FLAGS; b ""
