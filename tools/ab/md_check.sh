#!/bin/bash
cd "$GRAFT_REPO_ROOT"
python -m pytest tests/test_gpu_multidevice.py tests/test_gpu_rf_train.py -x -q -m gpu 2>&1 | tail -15
