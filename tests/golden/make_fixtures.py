#!/usr/bin/env python3
"""Generate tests/golden/* from the reference's own test DATA and literal known answers.

Run once in the build container (needs /root/reference); the outputs are committed so the
GPU box (which has no /root/reference) can run the same parity tests.

What is copied: data only -- the example feature file / qrel the reference's tests load
(tests/test_with_example_data.py:53-57), re-encoded as numeric arrays, plus the literal
expected numbers those tests assert.  No reference source text is stored.
"""
import json
import os
import shutil
from fractions import Fraction

import numpy as np

REF = os.environ.get("FASTRANK_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))


def f32_correctly_rounded(text: str) -> np.float32:
    """Decimal text -> nearest float32 (ties-to-even), like Rust's fast_float::parse::<f32>
    (src/libsvm.rs:90-91).  float(text)->float32 double-rounds in rare cases; fix them up."""
    exact = Fraction(text)
    c = np.float32(float(text))
    best = c
    for cand in (np.nextafter(c, np.float32(-np.inf)), np.nextafter(c, np.float32(np.inf))):
        if not np.isfinite(cand):
            continue
        d_best = abs(Fraction(float(best)) - exact)
        d_cand = abs(Fraction(float(cand)) - exact)
        if d_cand < d_best:
            best = cand
        elif d_cand == d_best:
            # tie -> even mantissa
            if (np.float32(cand).view(np.uint32) & 1) == 0:
                best = cand
    return np.float32(best)


def parse_ranksvm(path):
    """src/libsvm.rs:131-189 + src/instance.rs:104-130 + src/dataset.rs:211-256, for a file
    whose instances are all dense-representable: returns X [N, n_dim] float32 (missing = 0),
    y float64 (label parsed as f64 then narrowed to f32, libsvm.rs:147-155), qid strings."""
    rows, labels, qids = [], [], []
    max_f = 0
    with open(path) as fh:
        for line in fh:
            data = line.split("#", 1)[0]
            toks = data.split()
            if not toks:
                continue
            labels.append(float(np.float32(float(toks[0]))))
            assert toks[1].startswith("qid:")
            qids.append(toks[1][4:])
            feats = {}
            for t in toks[2:]:
                k, v = t.split(":", 1)
                feats[int(k)] = f32_correctly_rounded(v)
            max_f = max(max_f, max(feats) if feats else 0)
            rows.append(feats)
    n_dim = max_f + 1
    X = np.zeros((len(rows), n_dim), dtype=np.float32)
    for i, feats in enumerate(rows):
        for k, v in feats.items():
            X[i, k] = v
    return X, np.asarray(labels, dtype=np.float64), qids


def parse_qrel(path):
    """src/qrel.rs:65-102: 'qid unused docid gain' per line."""
    out = {}
    with open(path) as fh:
        for line in fh:
            row = line.split()
            if not row:
                continue
            out.setdefault(row[0], {})[row[2]] = float(np.float32(float(row[3])))
    return out


def main():
    ex = os.path.join(REF, "examples")
    X, y, qids = parse_ranksvm(os.path.join(ex, "trec_news_2018.train"))
    Xt, yt, qidst = parse_ranksvm(os.path.join(ex, "trec_news_2018.test"))
    if Xt.shape[1] < X.shape[1]:
        Xt = np.pad(Xt, ((0, 0), (0, X.shape[1] - Xt.shape[1])))
    with open(os.path.join(ex, "trec_news_2018.features.json")) as fh:
        names = json.load(fh)
    np.savez_compressed(
        os.path.join(HERE, "trec_news_2018.npz"),
        train_X=X, train_y=y, train_qid=np.asarray([int(q) for q in qids], dtype=np.int64),
        test_X=Xt, test_y=yt, test_qid=np.asarray([int(q) for q in qidst], dtype=np.int64),
    )
    qrel = parse_qrel(os.path.join(ex, "newsir18-entity.qrel"))
    with open(os.path.join(HERE, "newsir18_entity_qrel.json"), "w") as fh:
        json.dump(qrel, fh, sort_keys=True)
    # raw data files (data, not source) for the ranksvm-loader row of SURVEY.md section 8(f)
    os.makedirs(os.path.join(HERE, "data"), exist_ok=True)
    for fn in ("trec_news_2018.train", "trec_news_2018.test", "trec_news_2018.features.json",
               "newsir18-entity.qrel"):
        shutil.copyfile(os.path.join(ex, fn), os.path.join(HERE, "data", fn))
        os.chmod(os.path.join(HERE, "data", fn), 0o644)

    known = {
        "_doc": "Literal known answers asserted by the reference's own tests; file:line cites /root/reference.",
        "feature_names": names,  # examples/trec_news_2018.features.json
        "expected_n": 782,  # tests/test_with_example_data.py:41
        "expected_d": 6,  # tests/test_with_example_data.py:42
        "single_feature_ndcg5": {  # tests/test_with_example_data.py:16-23 (assertAlmostEqual, 7 places)
            "0": 0.10882970494872854,
            "para-fraction": 0.43942925167146063,
            "caption_position": 0.3838323029697044,
            "caption_count": 0.363671198812673,
            "pagerank": 0.28879573536768505,
            "caption_partial": 0.2119744912371782,
        },
        "expected_queries": sorted(set(qids)),  # tests/test_with_example_data.py:33-39 (same set)
        "rank_ties": {  # src/evaluators.rs:61-79
            "scores": [2.0, 2.0, 2.0, 2.0, 1.0],
            "gains": [0.0, 1.0, 2.0, 2.0, 2.0],
            "ids": [4, 3, 1, 2, 5],
            "expected_order": [4, 3, 1, 2, 5],
        },
        "compute_ndcg": {  # src/evaluators.rs:285-295, TREC_TOLERANCE = 5e-5
            "gains": [0.0, 1.0, 1.0, 1.0, 0.0, 0.0],
            "expected": 0.7328,
            "tolerance": 0.00005,
        },
        "regression_tree": {  # src/random_forest.rs:465-506: a tree that fits these exactly
            "xs": [1, 1, 2, 3, 4, 5, 6, 7, 8, 9],
            "ys": [7, 7, 7, 7, 2, 2, 2, 12, 12, 12],
            "tolerance": 1e-5,
        },
        "rf_seed42_ndcg5": 0.4367914517387043,  # src/random_forest.rs:462 (RNG-dependent; unpinned)
        "version": "0.7.0",  # tests/test_with_example_data.py:78
    }
    with open(os.path.join(HERE, "known_answers.json"), "w") as fh:
        json.dump(known, fh, indent=1, sort_keys=True)
    print("wrote fixtures:", X.shape, Xt.shape, len(qrel), "qrel queries")


if __name__ == "__main__":
    main()
