#!/bin/bash
# round 4, first GPU call: the multi-device tests (restart queue), the whole GPU suite, the tick-vs-groups sweep
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_multidevice.py -x -q -m gpu 2>&1 | tail -15 | tee gpurun_out/r04/md.log
python tools/tick_sweep.py > gpurun_out/r04/tick_vs_groups.json 2> gpurun_out/r04/tick_vs_groups.err; tail -20 gpurun_out/r04/tick_vs_groups.err
python -m pytest tests -x -q -m gpu 2>&1 | tail -8 | tee gpurun_out/r04/gputest.log
