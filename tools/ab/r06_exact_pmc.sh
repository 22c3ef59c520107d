#!/bin/bash
# Counters of the exact kernel's redo launches on hardties data (the isolated lock-step launches bench.py appends: largest grid).
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
O=$R/gpurun_out/exact_pmc; mkdir -p $O
CMD=(python bench.py --data ${KIND:-hardties} --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --no-side --repeats 0 --no-power)
i=0
for C in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY GRBM_GUI_ACTIVE" "FETCH_SIZE"; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/xp$i -o p -- "${CMD[@]}" > $O/p$i.log 2>&1
  i=$((i+1))
done
python3 - <<'PY' > $O/summary.txt
import csv, glob, os, collections
for d in sorted(glob.glob("/tmp/xp[0-9]*")):
    per = collections.defaultdict(dict); grid = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "linesearch_ndcg_kernel" not in r["Kernel_Name"]: continue
            k = int(r["Dispatch_Id"])
            per[k][r["Counter_Name"]] = per[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            per[k]["_dur_us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
            grid[k] = int(r["Grid_Size"])
    if not per: print(d, "no rows"); continue
    gmax = max(grid.values()); iso = [k for k in per if grid[k] == gmax]
    keys = sorted(set().union(*[per[k].keys() for k in iso]))
    print(d, "isolated launches", len(iso), "grid", gmax, {kk: round(sum(per[k].get(kk, 0) for k in iso) / len(iso), 1) for kk in keys})
PY
cat $O/summary.txt
