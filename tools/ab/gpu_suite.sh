#!/bin/bash
# the whole GPU suite + smoke + the headline bench line (what the driver runs at round end)
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/suite
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 | tee gpurun_out/suite/gputest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py --steps 20 --warmup 3 > gpurun_out/suite/bench.json 2> gpurun_out/suite/bench.err; tail -c 1500 gpurun_out/suite/bench.err; python - <<'PY'
import json
d=json.loads(open("gpurun_out/suite/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "isolated", d["roofline"]["avg_launch_ms"], "e2e", d["e2e_evals_per_s"], "cpu", d.get("cpu_baseline",{}).get("value"))
print({k: round(v,1) for k,v in d["kernels_ms"].items()})
PY
