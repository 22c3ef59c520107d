#!/usr/bin/env python3
"""BASELINE.json configs[4]: 500-tree ensemble scoring on the MSLR-WEB30K shape with the batched
tree-traversal HIP kernel.  Prints one JSON line (secondary benchmark; bench.py carries the
headline metric).  Forest = 500 random trees of depth <= 8: fid uniform, split = a
uniform quantile of that column, leaves U[0,4), weights 1.0."""
import argparse
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


random_forest = bench.random_trees


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="30k")
    ap.add_argument("--trees", type=int, default=500)
    ap.add_argument("--depth", type=int, default=8)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--check", type=int, default=20000, help="documents verified against the CPU oracle")
    ap.add_argument("--d", type=int, default=0, help="override the number of features (experiments on LDS occupancy)")
    args = ap.parse_args()
    n, d, q, seed = bench.SHAPES[args.shape]
    d = args.d or d
    X, y, qid = bench.gen_mslr_shaped(seed, n, d, q)
    ds = fr.CDataset.from_numpy(X, y, qid)
    rng = np.random.default_rng(7)
    trees = random_forest(rng, X, args.trees, args.depth)
    model = fr.CModel.from_dict({"Ensemble": {"weights": [1.0] * len(trees), "models": [{"DecisionTree": t} for t in trees]}})
    out = native.predict_scores_dense(model, ds, n)  # upload + warm-up
    native.profile_reset()
    native.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        out = native.predict_scores_dense(model, ds, n)
    wall = (time.perf_counter() - t0) / args.reps
    native.profile_enable(False)
    stats = native.profile_stats()
    kname = "tree_rank_kernel" if "tree_rank_kernel" in stats else "tree_ensemble_kernel"
    k = stats[kname]
    ok = None
    if args.check:
        from oracle import pyoracle as o
        m = min(args.check, n)
        sub = o.Dataset(X[:m], y[:m], qid[:m])
        ok = bool(np.array_equal(sub.score_ensemble(trees, [1.0] * len(trees)), out[:m]))
    b_rf = n * (4 * d + 8)
    sec = k["avg_ms"] * 1e-3
    nodes = sum(json.dumps(t).count("FeatureSplit") for t in trees)
    print(json.dumps({
        "metric": "tree-ensemble scoring passes/sec on MSLR-WEB30K shape", "value": 1.0 / sec, "unit": "passes/s",
        "doc_trees_per_s": n * len(trees) / sec, "kernel": kname, "kernel_avg_ms": k["avg_ms"], "wall_ms_incl_download": wall * 1e3,
        "config": {"workload": "%d trees depth<=%d (%d split nodes) x %d docs x %d features" % (len(trees), args.depth, nodes, n, d)},
        "roofline": {"bound": "hbm", "achieved": b_rf / sec / 1e9, "peak": 8000.0, "unit": "GB/s",
                     "frac": b_rf / sec / 1e9 / 8000.0, "traffic": None},
        "parity_first_docs_bit_exact": ok,
    }))


if __name__ == "__main__":
    main()
