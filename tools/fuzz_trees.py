#!/usr/bin/env python3
"""Randomised parity soak of tree-ensemble SCORING (config 5's kernels; src/model.rs:64-84,104-112): random small
datasets (1..40 features, integer / duplicated / heavy-tailed columns, NaN, +-inf, -0.0, denormals) and random
forests -- depths 0..11, 1..400 trees, feature subsets of every size (odd and even slot counts of the threshold-rank
kernel), thresholds drawn from the data (x == thr ties), from between data values, repeated across trees, non-f32
doubles, beyond the f32 range; features the dataset does not have; negative / zero / huge weights; bare single trees.
The device scores must equal the oracle's bit for bit (NaN features compare false, as in the oracle; a sum that overflows to
inf - inf = NaN must be NaN on both sides).
Usage: python tools/fuzz_trees.py --iters 300 [--seed 0]"""
import argparse
import collections
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fastrank_amd as fr  # noqa: E402
from fastrank_amd import native  # noqa: E402
from oracle import pyoracle as o  # noqa: E402


def make_matrix(rng):
    n = int(rng.choice([1, 7, 64, 65, 191, 192, 193, 500, 1500, 4000]))
    d = int(rng.choice([1, 2, 3, 4, 5, 8, 13, 24, 40]))
    X = np.empty((n, d), dtype=np.float32)
    for j in range(d):
        kind = rng.integers(0, 6)
        if kind == 0:
            col = rng.integers(-3, 4, n).astype(np.float32)                      # few distinct integers
        elif kind == 1:
            col = rng.lognormal(0, 3, n).astype(np.float32)                      # heavy tail
        elif kind == 2:
            col = rng.normal(0, 1, n).astype(np.float32)
        elif kind == 3:
            col = np.float32(rng.choice([0.0, -0.0, 1e-45, -1e-45, 1.0, 3.4028235e38, -3.4028235e38], n))
        elif kind == 4:
            col = np.round(rng.normal(0, 2, n), 1).astype(np.float32)            # decimals that are not f32-exact
        else:
            col = np.zeros(n, dtype=np.float32)                                  # constant column
        if rng.random() < 0.25:
            bad = rng.random(n) < 0.05
            col[bad] = rng.choice(np.float32([np.nan, np.inf, -np.inf]), int(bad.sum()))
        X[:, j] = col
    y = rng.integers(0, 5, n).astype(np.float64)
    qid = np.sort(rng.integers(0, max(1, n // 20) + 1, n)).astype(np.int64)
    return X, y, qid


def make_forest(rng, X):
    n, d = X.shape
    nfeat_pool = int(rng.integers(1, d + 1))
    pool = rng.choice(d, size=nfeat_pool, replace=False)
    if rng.random() < 0.3:
        pool = np.concatenate([pool, [d + int(rng.integers(0, 5))]])             # a feature the dataset lacks: reads 0.0
    depth = int(rng.choice([0, 1, 2, 3, 5, 7, 8, 9, 10, 11], p=[.03, .07, .1, .1, .15, .2, .15, .08, .07, .05]))
    ntrees = int(rng.choice([1, 2, 3, 8, 31, 32, 33, 77, 200, 400], p=[.1, .05, .05, .1, .1, .1, .1, .2, .15, .05]))
    if depth >= 9:
        ntrees = min(ntrees, 33)
    p_leaf = float(rng.choice([0.0, 0.05, 0.2]))
    shared = [float(v) for v in rng.normal(0, 1, 4)]                              # thresholds repeated across trees

    def threshold(f):
        col = X[:, f] if f < d else np.zeros(1, dtype=np.float32)
        fin = col[np.isfinite(col)]
        k = rng.integers(0, 6)
        if k == 0 and len(fin):
            return float(fin[rng.integers(0, len(fin))])                          # exactly a data value
        if k == 1 and len(fin):
            return float(np.nextafter(np.float64(fin[rng.integers(0, len(fin))]), rng.choice([-np.inf, np.inf])))
        if k == 2:
            return shared[int(rng.integers(0, len(shared)))]
        if k == 3:
            return float(rng.choice([0.0, -0.0, 1e300, -1e300, 1e-46, 0.1, 3.4028235e38, -3.4028236e38]))
        if len(fin):
            return float(np.quantile(fin.astype(np.float64), rng.random()))
        return float(rng.normal())

    def grow(dd):
        if dd == 0 or rng.random() < p_leaf:
            return {"LeafNode": float(rng.choice([rng.uniform(-2, 4), 0.0, -0.0, 1e300, rng.integers(-3, 4)]))}
        f = int(pool[rng.integers(0, len(pool))])
        return {"FeatureSplit": {"fid": f, "split": threshold(f), "lhs": grow(dd - 1), "rhs": grow(dd - 1)}}

    trees = [grow(depth) for _ in range(ntrees)]
    wk = rng.integers(0, 4)
    if wk == 0:
        weights = [1.0] * ntrees
    elif wk == 1:
        weights = rng.uniform(-1, 1, ntrees).tolist()
    elif wk == 2:
        weights = rng.choice([0.0, -0.0, 1.0, 1e-300, 3.0], ntrees).tolist()
    else:
        weights = rng.lognormal(0, 2, ntrees).tolist()
    return trees, weights


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    t0 = time.time()
    bad = 0
    kernels = collections.Counter()
    for it in range(args.iters):
        X, y, qid = make_matrix(rng)
        g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
        for _ in range(3):
            trees, weights = make_forest(rng, X)
            if len(trees) == 1 and rng.random() < 0.5:  # a bare tree (its output is the leaf itself)
                model = fr.CModel.from_dict({"DecisionTree": trees[0]})
                exp = c.score_ensemble(trees, [1.0])
            else:
                model = fr.CModel.from_dict({"Ensemble": {"weights": weights, "models": [{"DecisionTree": t} for t in trees]}})
                exp = c.score_ensemble(trees, weights)
            native.profile_reset()
            native.profile_enable(True)
            got = native.predict_scores_dense(model, g)
            native.profile_enable(False)
            for k in native.profile_stats():
                if k.startswith("tree_"):
                    kernels[k] += 1
            same = np.array_equal(got, exp, equal_nan=True)
            if not same:
                bad += 1
                print("MISMATCH iter", it, "n,d", X.shape, "trees", len(trees), "max |diff|", np.nanmax(np.abs(got - exp)), flush=True)
    print("fuzz_trees: %d datasets x 3 forests, %d mismatches, kernels %s, %.0f s" % (args.iters, bad, dict(kernels), time.time() - t0))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
