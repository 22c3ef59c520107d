cd "$GRAFT_REPO_ROOT"
m() { FR_LS_PIPELINE=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['roofline']['avg_launch_ms'])"; }
setw() { sed -i "s/^#define VERIFY_WPB [0-9]*/#define VERIFY_WPB $1/" fastrank_amd/csrc/device_dataset.inc; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1; }
m WPB4; m WPB4
setw 1; m WPB1; m WPB1
setw 2; m WPB2; m WPB2
setw 8; m WPB8; m WPB8
setw 4; m WPB4
python -m pytest tests -x -q -m gpu -p no:cacheprovider 2>&1 | tail -2
