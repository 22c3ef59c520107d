#!/usr/bin/env python3
"""CPU simulation (numpy only) for a 32-bit key variant of the full-ranking kernel: the score of a document is quantised to
QB bits inside the (query, candidate)'s score range and the document's index rides in the low bits; two adjacent sorted keys
whose quantised scores differ by <= 1 and whose gain classes differ need the exact comparison (cheap, local), two such pairs
in a row are not resolved locally (the pair goes to the 64-bit kernel).  Prints, per feature kind and candidate step, the
share of (query, candidate) pairs with a local fix and with an unresolved chain."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tools.sim_verify_lanes import cands  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--queries", type=int, default=400)
    ap.add_argument("--bits", type=int, nargs="+", default=[25, 24, 22])
    ap.add_argument("--data", default="mslr")
    a = ap.parse_args()
    n, d, q, seed = bench.SHAPES["10k"]
    X, y, qid = bench.gen_mslr_shaped(seed, n, d, q, a.data)
    starts = np.concatenate(([0], np.nonzero(np.diff(qid))[0] + 1, [n]))
    r = np.random.default_rng(3)
    w = r.uniform(-1, 1, d) * 0.2
    w[::8] += 1.0
    w /= np.abs(w).sum()
    base = X.astype(np.float64) @ w
    for f in (0, 1, 2, 3, 9, 11):
        xs = cands(w[f])
        A = base - X[:, f].astype(np.float64) * w[f]
        fx = X[:, f].astype(np.float64)
        for qb in a.bits:
            fix = np.zeros(51)
            chain = np.zeros(51)
            tot = 0
            for qi in range(a.queries):
                s, e = starts[qi], starts[qi + 1]
                if e - s < 2 or e - s > 128:
                    continue
                tot += 1
                rr, ff, yy = A[s:e], fx[s:e], y[s:e]
                S = rr[:, None] + ff[:, None] * xs[None, :]
                lo = rr.min() + np.minimum(xs * ff.min(), xs * ff.max())
                hi = rr.max() + np.maximum(xs * ff.min(), xs * ff.max())
                scale = (2.0**qb - 1) / np.maximum(hi - lo, 1e-300)
                Q = np.floor((S - lo[None, :]) * scale[None, :]).astype(np.int64)
                for c in range(51):
                    o = np.argsort(-Q[:, c], kind="stable")
                    dq = -np.diff(Q[o, c])
                    amb = (dq <= 1) & (yy[o][1:] != yy[o][:-1])
                    if amb.any():
                        fix[c] += 1
                        if (amb[1:] & amb[:-1]).any():
                            chain[c] += 1
            step = np.abs(xs - w[f])
            order = np.argsort(step)
            print("feature %d (kind %d) bits %d: queries %d; local fix share %.4f, chain share %.5f" % (f, f % 4, qb, tot, fix.sum() / (51 * tot), chain.sum() / (51 * tot)))
            print("   chain share by candidate (smallest step first):", " ".join("%.2f" % (chain[c] / tot) for c in order))


if __name__ == "__main__":
    main()
