#!/bin/bash
# round 5: stage times of the one-time upload at the 30K shape (FR_UPLOAD_TIMING), next to the raw host-link figures
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r05c}; mkdir -p $O
FR_UPLOAD_TIMING=1 python bench.py --steps 2 --warmup 1 --no-e2e --no-cpu-baseline --repeats 0 > $O/upload_bench.json 2> $O/upload_stages.txt
grep -v amdgpu.ids $O/upload_stages.txt
python -c "
import json,sys; d=json.loads(open('$O/upload_bench.json').read().strip().splitlines()[-1]); print('setup', d['setup'])"
