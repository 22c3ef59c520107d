#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04c; mkdir -p $O
FR_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_torch2_nccl.json 2> $O/bench_torch2_nccl.err; echo "rc=$? torchrun2 nccl->fallback"; grep "bench.py\]" $O/bench_torch2_nccl.err | head -3
python -c "
import json
d=json.loads(open('$O/bench_torch2_nccl.json').read().strip().splitlines()[-1]); print(d['value'], d['n_gpus'], d['config']['launcher'], d['config']['collective_note'], d['e2e']['model_sha1'])"
bash tools/ab/repeat_price.sh 2>&1 | tee $O/repeat_price.txt
bash tools/ab/fv_ab.sh "" "-DFV_WPS96=1" 2>&1 | tee $O/fv_wps96.txt
