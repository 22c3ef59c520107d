#!/bin/bash
# random-forest profiles of the round: rates (30K defaults, 10K k = 32) and the kernels' counters
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/rf
python tools/rfbench.py --shape 30k --trees 100 --cpu-trees 1 --check 2>/dev/null | tail -1 > gpurun_out/rf/rfbench_30k.json
python tools/rfbench.py --shape 10k --trees 30 --split-candidates 32 --cpu-trees 1 --check 2>/dev/null | tail -1 > gpurun_out/rf/rfbench_10k_k32.json
python - <<'PY'
import json
for f in ("gpurun_out/rf/rfbench_30k.json", "gpurun_out/rf/rfbench_10k_k32.json"):
    d = json.loads(open(f).read()); print(f, "trees/s %.2f wall %.2f first %.2f" % (d["value"], d["wall_s"], d["first_call_s"]), d["kernels_ms"], d.get("first_trees_identical_to_oracle"))
PY
bash tools/pmc_kernels.sh gpurun_out/pmc_rf_after rf_ -- python tools/rfbench.py --shape 30k --trees 24 --cpu-trees 0 > /dev/null 2>&1
cp gpurun_out/pmc_rf_after/summary.txt gpurun_out/rf/pmc_rf_after_summary.txt; grep -E "launches=|VALU issue|HBM bytes" gpurun_out/rf/pmc_rf_after_summary.txt
