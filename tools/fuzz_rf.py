#!/usr/bin/env python3
"""Randomised parity soak of random-forest TRAINING: random small datasets (ties, duplicates, integer columns, negative
gains), random RandomForestParams (all four split methods, weighted trees, sampling rates, depths, leaf supports, 2..40
split candidates, batches cut by FR_RF_BATCH_BYTES), full datasets and query / feature samples; the device trainer must
return the oracle's forest -- structure, thresholds, leaf values, weights -- bit for bit.
Usage: python tools/fuzz_rf.py --iters 300 [--seed 0]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import fastrank_amd as fr  # noqa: E402
from fuzz_parity import make_case  # noqa: E402
from oracle import pyoracle as o  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    o.set_mean_segment(o.DEVICE_MEAN_SEGMENT)
    t0 = time.time()
    bad = errs = nodes = sampled = fanned = 0
    methods = {}
    for it in range(args.iters):
        X, y, qid, measure, _ = make_case(rng)
        if rng.random() < 0.5:
            os.environ["FR_RF_BATCH_BYTES"] = str(int(rng.integers(20000, 400000)))
        else:
            os.environ.pop("FR_RF_BATCH_BYTES", None)
        if rng.random() < 0.2:  # the trees spread over two or three contexts on this device (capi.cpp, train_rf)
            os.environ["FR_DEVICES"] = str(rng.choice(["0,0", "0,0,0"]))
            fanned += 1
        else:
            os.environ.pop("FR_DEVICES", None)
        g = fr.CDataset.from_numpy(X, y, qid)
        fids = np.arange(X.shape[1], dtype=np.uint32)
        if rng.random() < 0.3 and len(np.unique(qid)) > 2:
            keep = rng.choice(np.unique(qid), size=max(1, len(np.unique(qid)) // 2), replace=False)
            g = g.subsample_queries([str(int(q)) for q in keep])
            mask = np.isin(qid, keep)
            X, y, qid = np.ascontiguousarray(X[mask]), np.ascontiguousarray(y[mask]), np.ascontiguousarray(qid[mask])
            sampled += 1
        if rng.random() < 0.25 and X.shape[1] > 2:
            fids = np.sort(rng.choice(X.shape[1], size=max(1, X.shape[1] // 2), replace=False)).astype(np.uint32)
            g = g.subsample_feature_names([str(int(f)) for f in fids])
            sampled += 1
        c = o.Dataset(X, y, qid)
        method = str(rng.choice(sorted(o.SPLIT_METHODS)))
        req = fr.TrainRequest.random_forest()
        req.measure = measure
        p = req.params
        p.quiet, p.seed, p.num_trees = True, int(rng.integers(0, 2 ** 31)), int(rng.integers(1, 7))
        p.weight_trees = bool(rng.random() < 0.4)
        p.split_method = {method: []}
        p.instance_sampling_rate = float(rng.choice([0.3, 0.5, 0.9, 1.0]))
        p.feature_sampling_rate = float(rng.choice([0.25, 0.5, 1.0]))
        p.min_leaf_support = int(rng.choice([1, 2, 3, 10])) if method != "TrueVarianceReduction" else int(rng.choice([2, 3, 10]))
        p.split_candidates = int(rng.choice([2, 3, 3, 8, 32, 40]))
        p.max_depth = int(rng.choice([1, 2, 4, 8, 12]))
        methods[method] = methods.get(method, 0) + 1
        try:
            exp_trees, exp_w, _ = c.rf_learn(measure, p.to_dict(), fids=fids)
        except RuntimeError as exc:  # the reference would have panicked (e.g. actual > ideal DCG with negative gains)
            try:
                g.train_model(req)
                bad += 1
                print("MISMATCH iter", it, "oracle:", exc, "device: no error")
            except Exception:
                errs += 1
            continue
        try:
            got = g.train_model(req).to_dict()
        except Exception as exc:
            bad += 1
            print("MISMATCH iter", it, "device error", str(exc)[:120])
            continue
        exp = {"Ensemble": {"weights": exp_w.tolist(), "models": [{"DecisionTree": t} for t in exp_trees]}}
        nodes += json.dumps(exp).count("FeatureSplit")
        if got != exp:
            bad += 1
            print("MISMATCH iter", it, json.dumps({"n": len(y), "d": X.shape[1], "measure": measure, "params": p.to_dict()}))
    print(json.dumps({"iters": args.iters, "mismatches": bad, "both_error": errs, "split_nodes": nodes, "methods": methods,
                      "sampled_views": sampled, "over_several_contexts": fanned, "seconds": round(time.time() - t0, 1)}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
