#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04d
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4
bash tools/ab/run_ab.sh 2>&1 | tee gpurun_out/r04d/visit_ab.txt
bash tools/ab/tail_lat.sh
