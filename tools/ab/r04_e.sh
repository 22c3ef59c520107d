#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04d
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
bash tools/ab/visit_ab2.sh
