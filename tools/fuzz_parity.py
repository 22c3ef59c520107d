#!/usr/bin/env python3
"""Randomised parity soak: small random datasets (ties, duplicates, integer columns, negative gains, long and
one-document queries), random coordinate-ascent parameters and measures; the device trainer must reproduce the
CPU oracle's restarts bit for bit.  Usage: python tools/fuzz_parity.py --iters 200 [--seed 0]"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import fastrank_amd as fr  # noqa: E402
from fastrank_amd import native  # noqa: E402
from oracle import pyoracle as o  # noqa: E402


def make_case(rng, measures=None, long_bias=False):
    nq = int(rng.integers(1, 40))
    # (long_bias: query lengths spread over the full-ranking kernel's size classes -- 16 .. 96 keys per lane, 1 .. 32 lanes)
    lens = np.maximum(1, rng.lognormal(np.log(rng.choice([3, 20, 80, 130, 250, 600] if long_bias else [3, 20, 80])), 0.8, nq).astype(int))
    lens = np.minimum(lens, 2500)
    if rng.random() < 0.2:
        lens[rng.integers(0, nq)] = int(rng.integers(300, 2500))
    n = int(lens.sum())
    d = int(rng.integers(1, 30))
    qid = np.repeat(rng.permutation(nq).astype(np.int64) + 1, lens)
    cols = []
    for j in range(d):
        kind = rng.integers(0, 5)
        if kind == 0:
            c = rng.uniform(0, 1, n)
        elif kind == 1:
            c = np.floor(rng.exponential(2.0, n))
        elif kind == 2:
            c = rng.lognormal(0, 2, n) * rng.choice([-1, 1], n)
        elif kind == 3:
            c = np.where(rng.random(n) < 0.7, 0.0, rng.uniform(0, 1, n))
        else:
            c = rng.integers(-2, 3, n).astype(float)
        cols.append(c)
    X = np.stack(cols, axis=1).astype(np.float32)
    if rng.random() < 0.4:  # duplicated documents: exact score ties decided by gain / id
        k = int(rng.integers(1, max(2, n // 3)))
        src, dst = rng.integers(0, n, k), rng.integers(0, n, k)
        X[dst] = X[src]
    labels = [0.0, 0.0, 1.0, 2.0, 3.0, 4.0]
    if rng.random() < 0.15:
        labels += [-1.0, 0.5]
    y = rng.choice(labels, n)
    if rng.random() < 0.2:
        y[qid == qid[0]] = 0.0
    measure = str(rng.choice(measures or ["ndcg@1", "ndcg@3", "ndcg@5", "ndcg@10", "ndcg@20", "ndcg@10", "mrr", "map", "ndcg", "ndcg@50"]))
    params = dict(num_restarts=int(rng.integers(1, 4)), num_max_iterations=int(rng.integers(1, 9)),
                  step_base=float(rng.choice([0.05, 0.01, 0.5])), step_scale=float(rng.choice([2.0, 1.5, 3.0])),
                  tolerance=float(rng.choice([0.001, 0.0, 0.01])), seed=int(rng.integers(0, 2 ** 31)),
                  normalize=bool(rng.random() < 0.8), init_random=bool(rng.random() < 0.8), output_ensemble=False, quiet=True)
    return X, y, qid, measure, params


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=100)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--measures", default="", help="comma list to draw the measure from (default: the mixed list)")
    ap.add_argument("--long", action="store_true", help="bias query lengths towards 100 .. 2000 documents")
    args = ap.parse_args()
    measures = [m for m in args.measures.split(",") if m] or None
    rng = np.random.default_rng(args.seed)
    o.set_mean_segment(o.DEVICE_MEAN_SEGMENT)
    t0 = time.time()
    paths, redone, bad, sampled = {}, 0, 0, 0
    for it in range(args.iters):
        X, y, qid, measure, params = make_case(rng, measures, args.long)
        g = fr.CDataset.from_numpy(X, y, qid)
        if rng.random() < 0.3 and len(np.unique(qid)) > 2:  # a query-sampled view (train/test split style)
            keep = rng.choice(np.unique(qid), size=max(1, len(np.unique(qid)) // 2), replace=False)
            g = g.subsample_queries([str(int(q)) for q in keep])
            mask = np.isin(qid, keep)
            X, y, qid = np.ascontiguousarray(X[mask]), np.ascontiguousarray(y[mask]), np.ascontiguousarray(qid[mask])
            sampled += 1
        c = o.Dataset(X, y, qid)
        req = fr.TrainRequest.coordinate_ascent()
        req.measure = measure
        req.params = fr.CoordinateAscentParams(**params)
        try:
            shard = native.train_model_shard(g, req, 0, params["num_restarts"])
        except Exception as exc:  # the reference panics in the same situations (e.g. actual > ideal with negative gains)
            exp = c.ca_learn(measure, params, threads=2)
            print("iter %d: device error %r, oracle err=%r" % (it, str(exc)[:80], exp[3]))
            if exp[3] == 0:
                bad += 1
            continue
        exp_s, exp_w, exp_e, err = c.ca_learn(measure, params, threads=2)
        st = shard["stats"]
        paths[st["path"]] = paths.get(st["path"], 0) + 1
        redone += st.get("verify_redone", 0)
        ok = err == 0 and st["useful_evals"] == int(exp_e.sum())
        for r in shard["restarts"]:
            ok = ok and r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist()
        if ok:  # the trained model through the reference API: scores and per-query metric
            best = exp_w[o.select_best(exp_s)]
            model = fr.CModel.from_dict({"Linear": {"weights": best.tolist()}})
            scores = c.score_linear(best)
            exp_pq, e2 = c.metric_from_scores(measure, scores)
            got = g.evaluate(model, measure)
            ok = e2 == 0 and got == dict(zip((str(int(q)) for q in c.query_ids()), exp_pq.tolist()))
        if not ok:
            bad += 1
            print("MISMATCH iter", it, json.dumps({"n": len(y), "d": X.shape[1], "measure": measure, "params": params, "oracle_err": err}))
    print(json.dumps({"iters": args.iters, "mismatches": bad, "paths": paths, "verify_redone": int(redone), "sampled_views": sampled,
                      "seconds": round(time.time() - t0, 1)}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
