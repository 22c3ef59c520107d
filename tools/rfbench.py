#!/usr/bin/env python3
"""Random-forest TRAINING on a synthetic MSLR shape: wall time of train_model with the reference's
default parameters, HIP-event time per kernel, and the CPU oracle timed on a bounded sample (a few trees)."""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="30k")
    ap.add_argument("--trees", type=int, default=100)
    ap.add_argument("--split-candidates", type=int, default=3)
    ap.add_argument("--method", default="SquaredError")
    ap.add_argument("--cpu-trees", type=int, default=1, help="trees the CPU oracle grows for the baseline (0 = skip)")
    ap.add_argument("--check", action="store_true", help="compare the first --cpu-trees trees with the oracle's")
    args = ap.parse_args()
    n, d, q, seed = bench.SHAPES[args.shape]
    X, y, qid = bench.gen_mslr_shaped(seed, n, d, q)
    ds = fr.CDataset.from_numpy(X, y, qid)
    req = fr.TrainRequest.random_forest()
    req.measure = "ndcg@10"
    p = req.params
    p.quiet, p.seed, p.num_trees, p.split_candidates = True, 42, args.trees, args.split_candidates
    p.split_method = {args.method: []}
    native.predict_scores_dense(fr.CModel.from_dict({"Linear": {"weights": [0.0] * d}}), ds, n)  # upload
    # warm-up: a forest of two trees (the first call pays for the runtime's first large allocations -- seconds on a cold box --
    # which a process that trains more than once does not pay again); reported as first_call_s
    p.num_trees = 2
    t0 = time.perf_counter()
    ds.train_model(req)
    first_call = time.perf_counter() - t0
    p.num_trees = args.trees
    native.profile_reset()
    native.profile_enable(True)
    t0 = time.perf_counter()
    model = ds.train_model(req)
    wall = time.perf_counter() - t0
    native.profile_enable(False)
    st = native.last_train_stats()
    kern = {k: round(v["total_ms"], 1) for k, v in native.profile_stats().items()}
    md = model.to_dict()["Ensemble"]["models"]
    nodes = sum(json.dumps(m).count("FeatureSplit") for m in md)
    out = {"metric": "random-forest training, trees/s on MSLR-WEB%s shape" % args.shape.upper(), "value": args.trees / wall, "unit": "trees/s",
           "wall_s": wall, "trees": args.trees, "split_nodes": nodes, "levels": st["ticks"], "batches": st["groups"],
           "candidates": st["raw_evals"], "kernels_ms": kern, "first_call_s": first_call,
           "config": {"workload": "%d docs x %d features x %d queries, defaults: 50%% of the queries, 25%% of the features per tree, "
                                  "depth <= 8, min leaf 10, %d split candidates, %s" % (n, d, q, args.split_candidates, args.method)}}
    if args.cpu_trees:
        from oracle import pyoracle as o
        c = o.Dataset(X, y, qid)
        pp = dict(p.to_dict(), num_trees=args.cpu_trees)
        t0 = time.perf_counter()
        trees, w, _ = c.rf_learn("ndcg@10", pp)
        cpu = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": args.cpu_trees / cpu, "unit": "trees/s", "cores": 1, "kind": "port",
                               "sample": "%d tree(s) of the same request, oracle/fastrank_oracle.c, one thread (the reference runs rayon over trees)" % args.cpu_trees}
        if args.check:
            out["first_trees_identical_to_oracle"] = all({"DecisionTree": t} == md[i] for i, t in enumerate(trees))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
