#!/usr/bin/env python3
"""Sampled views at the 30K shape: what a query sample of 10 / 25 / 50 / 80 / 100 % costs -- HBM it owns, time to its
device form, trainer start-up (the restarts' exact first scores), NDCG@10 tick, one evaluate call and one forest-scoring
pass -- each next to the same call on the whole dataset.  A view shares its parent's tiles and visits only the tiles that
hold its documents (DeviceDataset::create_view, PosMap), so every column should scale with the sample.
FR_VIEW_ALL_TILES=1 gives round 2's behaviour (position-parallel kernels visit every parent position) for comparison."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import fastrank_amd as fr  # noqa: E402
from fastrank_amd import native  # noqa: E402


def timed(fn, reps=3):
    fn()
    native.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    native.synchronize()
    return (time.perf_counter() - t0) / reps


def measure(ds, n_total, forest, linear):
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "ndcg@10"
    req.params.num_restarts, req.params.quiet, req.params.seed = 32, True, 42
    t0 = time.perf_counter()
    info = native.device_info(ds)
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    run = native.CoordinateAscentRun(ds, req)
    native.synchronize()
    t_init = time.perf_counter() - t0
    run.step(5)
    native.synchronize()
    t0 = time.perf_counter()
    run.step(40)
    native.synchronize()
    tick = (time.perf_counter() - t0) / 40
    run.close()
    return {"hbm_bytes_owned": info["hbm_bytes_owned"], "queries": info["queries"], "instances": info["instances"],
            "device_form_s": t_dev, "trainer_init_s": t_init, "ms_per_tick": tick * 1e3,
            "evaluate_ndcg10_ms": timed(lambda: native.evaluate_dense(linear, ds, "ndcg@10")) * 1e3,
            "forest_100_trees_pass_ms": timed(lambda: native.predict_scores_dense(forest, ds, 0)) * 1e3}


def main():
    n, d, q, seed = bench.SHAPES["30k"]
    X, y, qid = bench.gen_mslr_shaped(seed, n, d, q)
    parent = fr.CDataset.from_numpy(X, y, qid)
    rng = np.random.default_rng(7)
    trees = bench.random_trees(rng, X, 100, 8)
    forest = fr.CModel.from_dict({"Ensemble": {"weights": [1.0] * len(trees), "models": [{"DecisionTree": t} for t in trees]}})
    w = rng.uniform(-1, 1, d)
    linear = fr.CModel.from_dict({"Linear": {"weights": (w / np.abs(w).sum()).tolist()}})
    names = [str(v) for v in range(1, q + 1)]
    out = {"shape": "30k", "all_tiles": bool(os.environ.get("FR_VIEW_ALL_TILES")), "rows": {}}
    out["rows"]["100% (the dataset itself)"] = measure(parent, n, forest, linear)
    for pct in (80, 50, 25, 10):
        keep = rng.permutation(q)[: q * pct // 100]
        view = parent.subsample_queries([names[i] for i in sorted(keep.tolist())])
        out["rows"]["%d%% of the queries" % pct] = measure(view, n, forest, linear)
        del view
    base = out["rows"]["100% (the dataset itself)"]
    for k, r in out["rows"].items():
        r["relative_to_whole"] = {m: r[m] / base[m] for m in ("trainer_init_s", "ms_per_tick", "evaluate_ndcg10_ms", "forest_100_trees_pass_ms")}
        r["share_of_instances"] = r["instances"] / base["instances"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
