#!/bin/bash
# round 5: the headline and the four side lines (hard, ties, tiesmix, hardties) on ONE box, same command each
# usage: tools/ab/r05_kinds.sh <outdir-tag> [kinds...]
cd "$GRAFT_REPO_ROOT"; TAG=${1:-r05a}; shift; O=gpurun_out/$TAG; mkdir -p $O
KINDS=${@:-"mslr hard ties tiesmix hardties"}
for k in $KINDS; do
  python bench.py --steps 20 --warmup 5 --data $k --no-cpu-baseline > $O/bench_$k.json 2> $O/bench_$k.err; echo "rc=$? $k"; tail -c 300 $O/bench_$k.err
done
python - "$O" <<'PY'
import json,glob,sys
base=None
for k in ("mslr","hard","ties","tiesmix","hardties"):
    f="%s/bench_%s.json"%(sys.argv[1],k)
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(k,"NO LINE",e); continue
    if k=="mslr": base=d["value"]
    v=d["verify"]; e=d.get("e2e") or {}
    print("%-9s value %8.0f (%.2f of headline) median %8.0f ms/step %.3f iso %.3f redo %.2e slices/pair %s exact_groups %.3f exact_ls %.3f | e2e %.0f evals/s wall %.2fs exact_groups %.3f best %.4f" % (
        k, d["value"], d["value"]/base if base else 0, d["value_runs_min_median_max"][1], d["ms_per_step"], d["roofline"]["avg_launch_ms"], v["redo_fraction"] or 0,
        v["redo_slices_per_pair"], v["exact_group_share"], v["exact_line_search_share"], e.get("e2e_evals_per_s",0), e.get("wall_s",0), e.get("exact_group_share",0), e.get("best_score",0)))
    print("          runs", [round(x) for x in d["value_runs"]], "kernels_ms", {a:round(b,2) for a,b in d["kernels_ms"].items()})
PY
