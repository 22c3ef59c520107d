#!/bin/bash
# round 4: bench.py in every way it can be started on a one-GPU box
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r04b; mkdir -p $O
python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "rc=$? single"; tail -c 600 $O/bench.err
FR_BENCH_DEVICE=0 python bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_threads2.json 2> $O/bench_threads2.err; echo "rc=$? threads2"; tail -c 800 $O/bench_threads2.err
FR_BENCH_DEVICE=0 python bench.py --gpus 8 --steps 10 --warmup 3 --restarts-per-gpu 8 > $O/bench_threads8.json 2> $O/bench_threads8.err; echo "rc=$? threads8"; tail -c 800 $O/bench_threads8.err
FR_BENCH_DEVICE=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --steps 10 --warmup 3 > $O/bench_torch2_nccl.json 2> $O/bench_torch2_nccl.err; echo "rc=$? torchrun2 nccl->fallback"; tail -c 1200 $O/bench_torch2_nccl.err
FR_BENCH_DEVICE=0 FR_BENCH_BACKEND=gloo python bench.py --gpus 2 --launcher torch --steps 10 --warmup 3 > $O/bench_torch2_gloo.json 2> $O/bench_torch2_gloo.err; echo "rc=$? self-launched torch gloo"; tail -c 800 $O/bench_torch2_gloo.err
python bench.py --measure trees --steps 10 --warmup 2 > $O/bench_trees.json 2> $O/bench_trees.err; echo "rc=$? trees"; tail -c 400 $O/bench_trees.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r04b/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "NO LINE", e); continue
    print(f, d["value"], d["unit"], "n_gpus", d["n_gpus"], "ms", round(d["ms_per_step"],3), d["config"].get("launcher"), d["config"].get("collective_note"))
    if d.get("e2e"): print("   e2e", d["e2e"]["wall_s"], d["e2e"]["per_rank_ticks"], d["e2e"]["model_sha1"], d["e2e"]["what"][:90])
    if d.get("inprocess"): print("   inproc", json.dumps(d["inprocess"]["second_call"]), d["inprocess"]["same_model_as_e2e_leg"])
PY
