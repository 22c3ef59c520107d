#!/bin/bash
# One or two quick PMC passes over the bench command: per-launch averages of the given counters for the ISOLATED launches
# of the dominant kernel (largest grid), and their duration.  usage (GPU box): bash tools/pmc_quick.sh <outdir> "<counters>" ["<counters2>"]
set -u
OUT=${1:-gpurun_out/pmc_quick}; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
CMD=(python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-side --repeats 0 --no-power)
i=0
for C in "$@"; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/p$i" -o p -- "${CMD[@]}" > "$OUT/p$i.log" 2>&1
  i=$((i+1))
done
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "p[0-9]*"))):
    if not os.path.isdir(d): continue
    per = collections.defaultdict(dict); grid = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "linesearch_verify_kernel" not in r["Kernel_Name"]: continue
            k = int(r["Dispatch_Id"])
            per[k][r["Counter_Name"]] = per[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            per[k]["_dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
            grid[k] = int(r["Grid_Size"])
            per[k]["_lds"] = float(r.get("LDS_Block_Size", 0) or 0); per[k]["_vgpr"] = float(r.get("VGPR_Count", 0) or 0)
    if not per: print(d, "no rows"); continue
    gmax = max(grid.values()); iso = [k for k in per if grid[k] == gmax]
    print(os.path.basename(d), "isolated launches", len(iso), "grid", gmax)
    for c in sorted(per[iso[0]]):
        print("   %-28s %.6g" % (c, sum(per[k].get(c, 0.0) for k in iso) / len(iso)))
PY
