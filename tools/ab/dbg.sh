cd "$GRAFT_REPO_ROOT"
m() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"; }
for r in 1 2; do
FR_LS_PIPELINE=2 m P2
FR_LS_PIPELINE=3 m P3
FR_LS_PIPELINE=4 m P4
done
