#!/bin/bash
# kernel timeline of the pipelined headline bench: where the device idles between verify launches
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/trace
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace/kt -o b -- python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-e2e > gpurun_out/trace/bench.json 2> gpurun_out/trace/bench.err
f=$(ls gpurun_out/trace/kt/*/*kernel_trace.csv gpurun_out/trace/kt/*kernel_trace.csv 2>/dev/null | head -1)
python tools/kernel_gaps.py "$f" 0.25 0
python - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1]))); rows.sort(key=lambda r:int(r["Start_Timestamp"]))
rows=rows[int(len(rows)*0.6):int(len(rows)*0.6)+60]
t0=int(rows[0]["Start_Timestamp"])
for r in rows:
    n=r["Kernel_Name"].split("(")[0].replace("void frdev::","").replace("frdev::","")[:44]
    print("%9.1f %9.1f us  q=%-3s grid=%-8s %s"%((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-t0)/1e3,r.get("Queue_Id","?"),r.get("Grid_Size","?"),n))
PY
python -c "
import json; d=json.loads(open('gpurun_out/trace/bench.json').read().strip().splitlines()[-1]); print('value',d['value'],'ms_per_step',d['ms_per_step'],'iso',d['roofline']['avg_launch_ms'])"
