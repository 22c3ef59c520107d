#!/bin/bash
# Regenerates the round's measured artefacts on the GPU box into gpurun_out/profiles_new/ (copy the ones to be
# judged into profiles/ afterwards).  Usage (from the repo root):
#   gpurun --timeout 1500 -- 'bash tools/refresh_profiles.sh r01'
set -u
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_new
mkdir -p "$OUT"
# 1. the bench line (HIP-event roofline inside)
python bench.py --steps 20 --warmup 3 > "$OUT/${TAG}_bench.json" 2> "$OUT/bench.err"
# 2. rocprofv3 kernel trace + stats of the same command
# (without the pre-heating throwaway job and the hardties side line, so that the launches below sort into the legs by their order alone)
FR_BENCH_NO_PREHEAT=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o p -- python bench.py --steps 20 --warmup 3 --no-side > "$OUT/kt.log" 2>&1
cp "$OUT"/kt/*kernel_stats.csv "$OUT/${TAG}_kernel_stats.csv" 2>/dev/null || cp "$OUT"/kt/*/*kernel_stats.csv "$OUT/${TAG}_kernel_stats.csv"
# the dominant kernel's launches by kind (the trainer keeps three launches in flight, bench.py appends isolated ones)
python - "$OUT" > "$OUT/${TAG}_verify_launches.txt" <<'PY'
import csv, glob, os, sys
out = sys.argv[1]
f = (glob.glob(os.path.join(out, "kt", "*kernel_trace.csv")) + glob.glob(os.path.join(out, "kt", "*", "*kernel_trace.csv")))[0]
rows = [r for r in csv.DictReader(open(f)) if "linesearch_verify_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
gmax = max(int(r["Grid_Size_X"]) for r in rows)
dur = lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
iso = [r for r in rows if int(r["Grid_Size_X"]) == gmax]
first_iso, last_iso = rows.index(iso[0]), rows.index(iso[-1])
rest = rows[:first_iso]                 # warm-up, timed, then the instrumented pipelined steps (bench.py's legs in order)
e2e = rows[last_iso + 1:]               # the train-to-convergence leg
warm, timed, instr = rest[:9], rest[9:69], rest[69:]
print("rocprofv3 --kernel-trace of `FR_BENCH_NO_PREHEAT=1 python bench.py --steps 20 --warmup 3 --no-side`: linesearch_verify_kernel launches by kind")
for name, sel in (("warm-up (3 ticks x 3 sets)", warm), ("timed (20 ticks x 3 sets, overlapping)", timed), ("repeats of the timed region, instrumented pipelined steps", instr),
                  ("isolated lock-step (32 groups)", iso), ("train-to-convergence leg (408 ticks x 3 sets, fewer groups as restarts converge)", e2e)):
    if sel:
        d = [dur(r) for r in sel]
        print("%-42s n=%-4d avg %.4f ms  min %.4f  max %.4f" % (name, len(d), sum(d) / len(d), min(d), max(d)))
if timed:
    t0, t1 = int(timed[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in timed)
    print("timed launches span %.3f ms -> %.4f ms per tick" % ((t1 - t0) / 1e6, (t1 - t0) / 1e6 / (len(timed) / 3.0)))
PY
# 3. secondary paths: full-ranking measures, tree ensemble, end-to-end training
{
  for m in ndcg map mrr; do python tools/lsbench.py --reps 3 --measure $m 2>&1 | tail -6; done
} > "$OUT/${TAG}_fullrank_lsbench.txt"
python tools/treebench.py --reps 5 2>&1 | tail -1 > "$OUT/${TAG}_treebench.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_tree" -o p -- python tools/treebench.py --reps 5 --check 0 > "$OUT/kt_tree.log" 2>&1
cp "$OUT"/kt_tree/*kernel_stats.csv "$OUT/${TAG}_tree_kernel_stats.csv" 2>/dev/null || cp "$OUT"/kt_tree/*/*kernel_stats.csv "$OUT/${TAG}_tree_kernel_stats.csv"
# the reference's default measure (depth-less NDCG) and MAP: sort-and-verify on resident sums
{ for m in ndcg map; do python tools/train_e2e.py --measure $m --shape 30k --restarts 32 --max-ticks 272 2>&1 | tail -1; done; } > "$OUT/${TAG}_train_fullrank_30k.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_full" -o p -- python tools/train_e2e.py --measure ndcg --shape 30k --restarts 32 --max-ticks 136 > "$OUT/kt_full.log" 2>&1
cp "$OUT"/kt_full/*kernel_stats.csv "$OUT/${TAG}_fullrank_kernel_stats.csv" 2>/dev/null || cp "$OUT"/kt_full/*/*kernel_stats.csv "$OUT/${TAG}_fullrank_kernel_stats.csv"
# side measurements on tie-heavy and on hard data: bench line + a whole run each
for kind in ties tiesmix hard hardties; do
  python bench.py --steps 20 --warmup 3 --data $kind --no-cpu-baseline 2> "$OUT/bench_$kind.err" | tail -1 > "$OUT/${TAG}_bench_$kind.json"
done
# the N>1 path on one GPU (gloo, both ranks on device 0): strong scaling of a 32-restart job, static and work stealing
for steal in 0 4; do
  FR_BENCH_DEVICE=0 FR_BENCH_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
    bench.py --gpus 2 --steps 10 --warmup 2 --shape 10k --restarts-total 32 --steal-block $steal 2> "$OUT/bench_2rank_$steal.err" | tail -1 > "$OUT/${TAG}_bench_2rank_1gpu_steal$steal.json"
done
# random-forest training and sampled views
python tools/rfbench.py --shape 30k --trees 100 --cpu-trees 1 --check 2>&1 | tail -1 > "$OUT/${TAG}_rfbench_30k.json"
python tools/rfbench.py --shape 10k --trees 30 --split-candidates 32 --cpu-trees 1 --check 2>&1 | tail -1 > "$OUT/${TAG}_rfbench_10k_k32.json"
python tools/viewbench.py 2>&1 | tail -1 > "$OUT/${TAG}_views_30k.json"
# round 3: config 5 as a bench line, the in-process fan-out (two contexts on the one GPU of this box), NDCG with a depth beyond 20
python bench.py --measure trees --steps 20 --warmup 3 2> "$OUT/bench_trees.err" | tail -1 > "$OUT/${TAG}_bench_trees.json"
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --inprocess-devices 0,0 2> "$OUT/bench_inproc.err" | tail -1 > "$OUT/${TAG}_bench_inprocess_2ctx_1gpu.json"
{ for m in ndcg@50 ndcg@100; do python tools/train_e2e.py --measure $m --shape 30k --restarts 32 --max-ticks 136 2>&1 | tail -1; done; } > "$OUT/${TAG}_train_ndcg_cut_30k.json"
FR_UPLOAD_TIMING=1 python tools/train_e2e.py --shape 30k --restarts 32 --max-ticks 3 2>&1 | grep "upload" > "$OUT/${TAG}_upload_stages.txt"
python tools/train_e2e.py 2>&1 | tail -1 > "$OUT/${TAG}_train_e2e_10k.json"
python tools/train_e2e.py --measure mrr --shape 30k --restarts 32 --max-ticks 272 2>&1 | tail -1 > "$OUT/${TAG}_train_mrr_30k.json"
{ python tools/train_e2e.py --shape 30k --restarts 32 2>&1 | tail -1; FR_LS_EXACT=1 python tools/train_e2e.py --shape 30k --restarts 32 2>&1 | tail -1; } > "$OUT/${TAG}_train_e2e_30k.json"
rm -rf "$OUT/kt" "$OUT/kt_tree" "$OUT/kt_full"
ls -la "$OUT"
