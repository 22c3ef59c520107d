#!/bin/bash
# kernel timeline of a tick's kernels queued back to back with no host in between (FR_LS_REPEAT=4), lock step and pipelined:
# where do the gaps between dependent kernels of one stream come from?
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/trace
for mode in 0 3; do
rm -rf gpurun_out/trace/kr$mode
FR_LS_PIPELINE=$mode FR_LS_REPEAT=4 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/trace/kr$mode -o b -- python bench.py --steps 12 --warmup 3 --no-cpu-baseline --no-e2e > gpurun_out/trace/bench_r$mode.json 2> gpurun_out/trace/bench_r$mode.err
f=$(ls gpurun_out/trace/kr$mode/*/*kernel_trace.csv gpurun_out/trace/kr$mode/*kernel_trace.csv 2>/dev/null | head -1)
echo "=== FR_LS_PIPELINE=$mode  $f"
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# the timed region: take a window in the middle third
k=int(len(rows)*0.45); rows=rows[k:k+70]
t0=int(rows[0]["Start_Timestamp"])
for r in rows:
    n=r["Kernel_Name"].split("(")[0].replace("void frdev::","").replace("frdev::","")[:40]
    print("%9.1f %9.1f us dur %8.1f q=%-3s grid=%-8s %s"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3,r.get("Queue_Id","?"),r.get("Grid_Size","?"),n))
PY
done
