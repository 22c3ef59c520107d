#!/usr/bin/env python3
"""VERDICT r04, missing 2: ONE targeted check of the random-forest golden (src/random_forest.rs:427-463, 0.4367914517387043).
The test's configuration samples 1 of 6 features per tree (feature_sampling_rate 0.25, src/sampling.rs:46).  Question: do the
trees of the restated forest that draw a feature other than 5 collapse to single leaves (FeatureStats::compute,
src/normalizers.rs:13-37, + generate_split_candidate, src/random_forest.rs:211-283), so that the forest scores like "ten trees
on one feature"?  Prints, per tree of the restatement's forest, the feature drawn and whether it split; then the forest every
tree of which is FORCED onto feature f, for every f (query samples as drawn).  Test tooling (CPU, uses tools/rf_pin.py)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import rf_pin as R  # noqa: E402
from oracle import pyoracle as o  # noqa: E402


def depth(t):
    return 1 if "LeafNode" in t else 1 + max(depth(t["FeatureSplit"]["lhs"]), depth(t["FeatureSplit"]["rhs"]))


def main():
    d = np.load(os.path.join(ROOT, "tests", "golden", "trec_news_2018.npz"))
    X, y, qid = d["train_X"], d["train_y"], d["train_qid"]
    params = dict(num_trees=10, seed=42, min_leaf_support=1, max_depth=10, split_candidates=32, instance_sampling_rate=0.5,
                  feature_sampling_rate=0.25)
    ds = o.Dataset(X, y, qid)
    qs = [str(int(q)) for q in qid]

    def score(trees):
        v, _ = ds.metric_from_scores("ndcg@5", ds.score_ensemble(trees, [1.0] * len(trees)))
        return float(np.mean(v))

    rand = R.Rand64(42)
    seeds = [rand.rand_u64() for _ in range(10)]
    features, queries = list(range(X.shape[1])), sorted(set(qs))
    by_q = {}
    for i, q in enumerate(qs):
        by_q.setdefault(q, []).append(i)

    def sample(seed, force=None):
        lr = R.Rand64(seed)
        fs = R.sample_without_replacement(features, lr, max(1, int(len(features) * 0.25)))
        qsel = set(R.sample_without_replacement(queries, lr, max(1, int(len(queries) * 0.5))))
        ids = [i for q in by_q for i in by_q[q] if q in qsel]
        return (fs if force is None else [force]), ids

    trees = []
    for k, s in enumerate(seeds):
        fs, ids = sample(s)
        t = R.learn_tree(X, y, ids, fs, params) or {"LeafNode": R.mean_gain(ids, y)}
        trees.append(t)
        print("tree %d: feature %s, %s, depth %d, NDCG@5 alone %.4f" % (k, fs, "LEAF" if "LeafNode" in t else "splits", depth(t), score([t])))
    print("restated forest: %.6f   (reference: 0.4367914517387043)" % score(trees))
    for f in features:
        tf = []
        for s in seeds:
            fs, ids = sample(s, force=f)
            tf.append(R.learn_tree(X, y, ids, fs, params) or {"LeafNode": R.mean_gain(ids, y)})
        print("all ten trees forced onto feature %d: %d leaves, depths %s, forest NDCG@5 %.6f" % (
            f, sum("LeafNode" in t for t in tf), [depth(t) for t in tf], score(tf)))


if __name__ == "__main__":
    main()
