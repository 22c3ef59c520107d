#!/usr/bin/env python3
"""Micro-benchmark of the hot operator (fr_evaluate_candidates -> linesearch_ndcg_kernel) on a
synthetic MSLR-shaped matrix: G line groups x 51 candidates per launch.  Used for kernel tuning
and for the rocprofv3 --pmc passes whose summaries live under profiles/."""
import argparse
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
    ap.add_argument("--groups", type=int, default=32)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--iters", type=int, default=25)
    ap.add_argument("--measure", default="ndcg@10")
    ap.add_argument("--feature", type=int, default=-1, help="fixed feature for all groups (-1 = random)")
    ap.add_argument("--calib", action="store_true", help="also run score_linear_kernel once (reads the matrix exactly once: FETCH_SIZE calibration)")
    args = ap.parse_args()
    n, d, q, seed = bench.SHAPES[args.shape]
    X, y, qid = bench.gen_mslr_shaped(seed, n, d, q)
    ds = fr.CDataset.from_numpy(X, y, qid)
    rng = np.random.default_rng(1)
    feats, bases, cands = [], [], []
    for g in range(args.groups):
        w = rng.uniform(-1, 1, d)
        w /= np.abs(w).sum()
        f = int(rng.integers(0, d)) if args.feature < 0 else args.feature
        orig = w[f]
        c = [0.0]
        for sign in (-1.0, 1.0):
            step = 0.05 * sign
            if orig != 0 and abs(step) > 0.5 * abs(orig):
                step = 0.05 * abs(orig) * sign
            tot = step
            for _ in range(args.iters):
                c.append(orig + tot)
                step *= 2.0
                tot += step
        feats.append(f), bases.append(w), cands.append(np.asarray(c))
    bases = np.asarray(bases)
    native.evaluate_candidates(ds, args.measure, feats, bases, cands)  # upload + warm-up
    native.profile_reset()
    native.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.reps):
        means = native.evaluate_candidates(ds, args.measure, feats, bases, cands)
    if args.calib:
        m = fr.CModel.from_dict({"Linear": {"weights": bases[0].tolist()}})
        native.predict_scores_dense(m, ds, n)
    native.synchronize()
    wall = (time.perf_counter() - t0) / args.reps
    native.profile_enable(False)
    st = native.profile_stats()
    evals = sum(len(c) for c in cands)
    k = st.get("linesearch_verify_kernel") or st.get("linesearch_ndcg_kernel") or st.get("rank_metric_kernel") or st.get("metric_sort_kernel")
    print("shape=%s groups=%d evals/launch=%d  kernel avg %.3f ms  wall/call %.3f ms  -> %.0f evals/s (kernel)  mean[0][:3]=%s" % (
        args.shape, args.groups, evals, k["avg_ms"], wall * 1e3, evals / (k["avg_ms"] * 1e-3), means[0][:3]))
    for name, v in sorted(st.items()):
        print("   %-28s launches=%d avg=%.3f ms" % (name, v["launches"], v["avg_ms"]))


if __name__ == "__main__":
    main()
