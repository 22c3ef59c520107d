#!/bin/bash
# (round 6: the tuning / ablation switches this script sets exist only in a pricing build -- csrc/device.hpp pricing_env)
export FR_BUILD_FLAGS="${FR_BUILD_FLAGS:--DFR_PRICING}"; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/views
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "view or tree or sampl or ensemble or single_feature" 2>&1 | tail -4
python tools/viewbench.py 2>&1 | tail -1 > gpurun_out/views/views_30k.json
FR_VIEW_ALL_TILES=1 python tools/viewbench.py 2>&1 | tail -1 > gpurun_out/views/views_30k_all_tiles.json
python - <<'PY'
import json
for f in ("gpurun_out/views/views_30k.json","gpurun_out/views/views_30k_all_tiles.json"):
    d=json.load(open(f)); print(f, "all_tiles", d["all_tiles"])
    for k,r in d["rows"].items():
        print("  %-28s inst %.2f init %.3fs tick %.2fms eval %.2fms forest %.2fms  rel:"%(k, r["share_of_instances"], r["trainer_init_s"], r["ms_per_tick"], r["evaluate_ndcg10_ms"], r["forest_100_trees_pass_ms"]), {m: round(v,2) for m,v in r["relative_to_whole"].items()})
PY
