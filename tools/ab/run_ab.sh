#!/bin/bash
# A/B of two kernels_verify.inc variants on one box (development aid): put the other variant at tools/ab/kernels_verify_A.inc
cd "$GRAFT_REPO_ROOT"
m() { FR_LS_PIPELINE=0 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', d['ms_per_step'], d['roofline']['avg_launch_ms'])"; }
m B; m B
cp fastrank_amd/csrc/kernels_verify.inc /tmp/B.inc
cp tools/ab/kernels_verify_A.inc fastrank_amd/csrc/kernels_verify.inc
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
m A; m A
cp /tmp/B.inc fastrank_amd/csrc/kernels_verify.inc
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
m B
