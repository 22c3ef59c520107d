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
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt" -o p -- python bench.py --steps 20 --warmup 3 > "$OUT/kt.log" 2>&1
cp "$OUT"/kt/*kernel_stats.csv "$OUT/${TAG}_kernel_stats.csv" 2>/dev/null || cp "$OUT"/kt/*/*kernel_stats.csv "$OUT/${TAG}_kernel_stats.csv"
# 3. secondary paths: full-ranking measures, tree ensemble, end-to-end training
{
  for m in ndcg map mrr; do python tools/lsbench.py --reps 3 --measure $m 2>&1 | tail -6; done
} > "$OUT/${TAG}_fullrank_lsbench.txt"
python tools/treebench.py --reps 5 2>&1 | tail -1 > "$OUT/${TAG}_treebench.json"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/kt_tree" -o p -- python tools/treebench.py --reps 5 --check 0 > "$OUT/kt_tree.log" 2>&1
cp "$OUT"/kt_tree/*kernel_stats.csv "$OUT/${TAG}_tree_kernel_stats.csv" 2>/dev/null || cp "$OUT"/kt_tree/*/*kernel_stats.csv "$OUT/${TAG}_tree_kernel_stats.csv"
python tools/train_e2e.py 2>&1 | tail -1 > "$OUT/${TAG}_train_e2e_10k.json"
python tools/train_e2e.py --measure mrr --shape 30k --restarts 32 --max-ticks 272 2>&1 | tail -1 > "$OUT/${TAG}_train_mrr_30k.json"
{ python tools/train_e2e.py --shape 30k --restarts 32 2>&1 | tail -1; FR_LS_EXACT=1 python tools/train_e2e.py --shape 30k --restarts 32 2>&1 | tail -1; } > "$OUT/${TAG}_train_e2e_30k.json"
rm -rf "$OUT/kt" "$OUT/kt_tree"
ls -la "$OUT"
