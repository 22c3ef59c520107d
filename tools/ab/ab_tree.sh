#!/bin/bash
# A/B on ONE box: the working tree (B) against csrc files saved under tools/ab/A (A), e.g.
#   mkdir -p tools/ab/A; git show HEAD:fastrank_amd/csrc/kernels_verify.inc > tools/ab/A/kernels_verify.inc
# bench.py: value, ms per step, isolated launch.
cd "$GRAFT_REPO_ROOT"
m() { python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$1', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],3))"; }
m B; m B
mkdir -p /tmp/Bsave
for f in tools/ab/A/*; do b=$(basename $f); cp fastrank_amd/csrc/$b /tmp/Bsave/$b; cp $f fastrank_amd/csrc/$b; done
python -c "from fastrank_amd import _build; _build.build(force=True)" > /dev/null 2>&1
m A; m A
for f in /tmp/Bsave/*; do cp $f fastrank_amd/csrc/$(basename $f); done
python -c "from fastrank_amd import _build; _build.build(force=True)" > /dev/null 2>&1
m B
