#!/bin/bash
# What the exact kernel's redo launches cost on hardties data: rocprofv3 kernel trace of the bench command (durations, grids),
# pipelined (three sets in flight: durations include waiting for wave slots) and lock step (FR_LS_PIPELINE=0: nothing else on the device).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/exact_hardties; mkdir -p $O
for mode in 1 0; do
  rm -rf /tmp/prof_x
  FR_LS_PIPELINE=$mode rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_x -o p -- python $R/bench.py --data ${KIND:-hardties} --steps 20 --warmup 5 --no-cpu-baseline --no-e2e --no-side --repeats 0 > $O/pipe$mode.log 2>&1
  f=$(find /tmp/prof_x -name '*kernel_stats.csv' | head -1); head -8 "$f" | cut -c1-200 > $O/pipe${mode}_kernel_stats.csv
  t=$(find /tmp/prof_x -name '*kernel_trace.csv' | head -1)
  python3 - "$t" > $O/pipe${mode}_exact_launches.txt <<'PY'
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
for name in ('linesearch_ndcg_kernel', 'linesearch_verify_kernel', 'segment_sum_kernel', 'final_mean_kernel', 'rslot_kernel'):
    ex = [r for r in rows if name in r['Kernel_Name']][-60:]
    us = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in ex]
    print(name, 'last', len(ex), 'launches: grid', sorted(set(r['Grid_Size_X'] for r in ex)), 'us min/median/mean/max %.1f %.1f %.1f %.1f' % (min(us), statistics.median(us), statistics.mean(us), max(us)))
PY
done
