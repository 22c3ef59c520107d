#!/usr/bin/env python3
"""Attempt to pin the Rand64 restatement (oorandom =11.1.0) against the reference's only RNG-dependent
known answer: src/random_forest.rs:427-463 (10 trees, seed 42 -> mean NDCG@5 = 0.4367914517387043).
Restates random-forest TRAINING (src/random_forest.rs:211-408, src/sampling.rs:36-65,
src/normalizers.rs:13-37) in plain Python for that purpose only (test tooling, not product)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as o  # noqa: E402

M128 = (1 << 128) - 1
MULT = 47026247687942121848144207491837523525
DEFAULT_INC = 0x2FE0E169FFBD06E35BC307BD4D2F814F


class Rand64:
    def __init__(self, seed, variant="oorandom"):
        self.variant = variant
        self.state = 0
        self.inc = ((DEFAULT_INC << 1) | 1) & M128
        self.rand_u64()
        self.state = (self.state + seed) & M128
        self.rand_u64()

    def rand_u64(self):
        old = self.state
        self.state = (old * MULT + self.inc) & M128
        if self.variant == "oorandom":
            xs = (((old >> 29) ^ old) >> 58) & 0xFFFFFFFFFFFFFFFF
        else:  # canonical PCG XSL-RR 128/64
            xs = ((old >> 64) ^ old) & 0xFFFFFFFFFFFFFFFF
        rot = old >> 122
        return ((xs >> rot) | (xs << ((64 - rot) & 63))) & 0xFFFFFFFFFFFFFFFF

    def rand_range(self, lo, hi):
        s = hi - lo
        m = self.rand_u64() * s
        left = m & 0xFFFFFFFFFFFFFFFF
        if left < s:
            thr = ((1 << 64) - s) % s
            while left < thr:
                m = self.rand_u64() * s
                left = m & 0xFFFFFFFFFFFFFFFF
        return (m >> 64) + lo


def shuffle(v, rand):
    n = len(v)
    for i in range(n):
        j = rand.rand_range(i, n)
        v[i], v[j] = v[j], v[i]


def sample_without_replacement(data, rand, count):
    v = list(data)
    shuffle(v, rand)
    return v[:count]


def mean_gain(ids, y):
    if len(ids) == 0:
        return 0.0
    s = 0.0
    for i in ids:
        s += float(y[i])
    return s / len(ids)


def sq_error(ids, y):
    out = mean_gain(ids, y)
    s = 0.0
    for i in ids:
        d = out - float(y[i])
        s += d * d
    return s


def learn_tree(X, y, ids, fids, params, depth=1):
    """learn_recursive: returns a tree dict or None (= Err)."""
    if not fids or not ids:
        return None
    if depth >= params["max_depth"]:
        return None
    if len(ids) < params["min_leaf_support"]:
        return None
    cands = []
    for f in fids:
        vals = [float(X[i, f]) for i in ids]
        if len(vals) <= 1:
            continue  # StreamingStats::finish needs > 1 element
        fmin, fmax = min(vals), max(vals)
        labels = [float(y[i]) for i in ids]
        if max(labels) == min(labels):
            continue
        k = params["split_candidates"]
        rng_ = fmax - fmin
        order = sorted(range(len(ids)), key=lambda t: vals[t])
        scores = [vals[t] for t in order]
        sids = [ids[t] for t in order]
        splits = [(i / k) * rng_ + fmin for i in range(1, k)]
        positions = []
        pi = 0
        for pos in splits:
            while pi < len(sids) and scores[pi] < pos:
                pi += 1
            if positions and positions[-1][1] == pi:
                continue
            positions.append((pos, pi))
        best = None
        for pos, right in positions:
            lhs, rhs = sids[:right], sids[right:]
            if len(lhs) < params["min_leaf_support"] or len(rhs) < params["min_leaf_support"]:
                continue
            imp = -(sq_error(lhs, y) + sq_error(rhs, y))
            if best is None or imp >= best[0]:  # sort_unstable_by_key(...).last(): a maximum
                best = (imp, pos, right)
        if best is not None:
            cands.append((best[0], f, best[1], sids[:best[2]], sids[best[2]:]))
    if not cands:
        return None
    imp, f, split, lhs, rhs = max(cands, key=lambda c: c[0])
    left = learn_tree(X, y, lhs, fids, params, depth + 1)
    if left is None:
        left = {"LeafNode": mean_gain(lhs, y)}
    right = learn_tree(X, y, rhs, fids, params, depth + 1)
    if right is None:
        right = {"LeafNode": mean_gain(rhs, y)}
    return {"FeatureSplit": {"fid": f, "split": split, "lhs": left, "rhs": right}}


def learn_ensemble(X, y, qid_str, params, variant):
    rand = Rand64(params["seed"], variant)
    seeds = [rand.rand_u64() for _ in range(params["num_trees"])]
    features = sorted(range(X.shape[1]))
    queries = sorted(set(qid_str))
    by_q = {}
    for i, q in enumerate(qid_str):
        by_q.setdefault(q, []).append(i)
    trees = []
    for s in seeds:
        lr = Rand64(s, variant)
        nf = max(1, int(len(features) * params["feature_sampling_rate"]))
        nq = max(1, int(len(queries) * params["instance_sampling_rate"]))
        fs = sample_without_replacement(features, lr, nf)
        qs = set(sample_without_replacement(queries, lr, nq))
        ids = [i for q in by_q for i in by_q[q] if q in qs]
        t = learn_tree(X, y, ids, fs, params)
        if t is None:
            t = {"LeafNode": mean_gain(ids, y)}
        trees.append(t)
    return trees


def main():
    d = np.load(os.path.join(ROOT, "tests", "golden", "trec_news_2018.npz"))
    X, y, qid = d["train_X"], d["train_y"], d["train_qid"]
    known = json.load(open(os.path.join(ROOT, "tests", "golden", "known_answers.json")))
    params = dict(num_trees=10, seed=42, min_leaf_support=1, max_depth=10, split_candidates=32,
                  instance_sampling_rate=0.5, feature_sampling_rate=0.25)
    ds = o.Dataset(X, y, qid)
    qs = [str(int(q)) for q in qid]
    for variant in ("oorandom", "xslrr"):
        trees = learn_ensemble(X, y, qs, params, variant)
        scores = ds.score_ensemble(trees, [1.0] * len(trees))
        vals, _ = ds.metric_from_scores("ndcg@5", scores)
        feats = [t.get("FeatureSplit", {}).get("fid") for t in trees]
        print(variant, "mean NDCG@5 =", repr(float(np.mean(vals))), "target", known["rf_seed42_ndcg5"], "root features", feats)


if __name__ == "__main__":
    main()
