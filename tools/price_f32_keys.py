#!/usr/bin/env python3
"""What would f32 keys cost the NDCG@k bound-and-verify kernel?  (the socket sits at its power cap and the f64
min / max / fma chains are the energy; f32 keys would halve that datapath and the list registers.)  The verification rule stays
what it is -- two keys next to each other in a lane's top K + 1 must differ by more than the error bound unless they are of
one gain class -- but the bound widens from ~D * 2^-53 to the f32 roundings: the resident sum rounded to f32 and one f32 fma,
2^-23 (|key|) per key, plus the gain-class bits in the low mantissa (3 bits: 2^-21).  CPU only, numpy: the 10K shape, a model
that has found the signal columns, the reference's 51 candidates of a feature (tools/sim_verify_lanes.cands); prints per
data kind and feature kind the share of (query, candidate) values and of (query, group) pairs that would go to the exact
kernel.  Today's f64 bound is printed next to it as the check of the simulation (measured: 4.5e-6 of the pairs on mslr)."""
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
    ap.add_argument("--queries", type=int, default=1500)
    ap.add_argument("--depth", type=int, default=10)
    ap.add_argument("--data", nargs="+", default=["mslr", "hard"])
    a = ap.parse_args()
    n, d, q, seed = bench.SHAPES["10k"]
    bounds = (("f64 keys (today: ~ (D + 2) 2^-53)", 138.0 * 2.0 ** -53), ("f32 keys, no class bits (2^-23)", 2.0 ** -23),
              ("f32 keys, 3 class bits (2^-21)", 2.0 ** -21))
    for kind in a.data:
        X, y, qid = bench.gen_mslr_shaped(seed, n, d, q, kind)
        starts = np.concatenate(([0], np.nonzero(np.diff(qid))[0] + 1, [n]))
        r = np.random.default_rng(3)
        w = r.uniform(-1, 1, d) * 0.2
        w[::8] += 1.0
        w /= np.abs(w).sum()
        base = X.astype(np.float64) @ w
        print("data kind %s, %d queries, depth %d" % (kind, min(a.queries, q), a.depth))
        for f in (0, 1, 2, 3, 8, 9):
            xs = cands(w[f])
            bad_val = np.zeros(len(bounds))
            bad_pair = np.zeros(len(bounds))
            bad_slice = np.zeros(len(bounds))
            tot = 0
            for qi in range(min(a.queries, q)):
                s, e = starts[qi], starts[qi + 1]
                if e - s < 2:
                    continue
                A = base[s:e] - X[s:e, f].astype(np.float64) * w[f]
                keys = A[:, None] + X[s:e, f].astype(np.float64)[:, None] * xs[None, :]   # [docs][51]
                order = np.argsort(-keys, axis=0, kind="stable")
                top = min(a.depth + 1, e - s)
                ks = np.take_along_axis(keys, order[:top], axis=0)
                gs = y[s:e][order[:top]]
                gap = ks[:-1] - ks[1:]
                mag = np.abs(ks[:-1]) + np.abs(ks[1:])
                differ = gs[:-1] != gs[1:]
                tot += 1
                for bi, (_, rho) in enumerate(bounds):
                    fail = ((gap <= rho * mag) & differ).any(axis=0)       # per candidate
                    bad_val[bi] += fail.mean()
                    bad_pair[bi] += fail.any()
                    bad_slice[bi] += sum(fail[i:i + 16].any() for i in range(0, 51, 16)) / 4.0
            col = ["uniform + signal", "small integers", "heavy tail", "sparse"][f % 4] if f != 8 else "uniform + signal (2nd)"
            print("  feature %d (%s)" % (f, col))
            for bi, (name, _) in enumerate(bounds):
                print("    %-36s values %.2e  (query, group) pairs %.2e  16-candidate slices %.2e" % (name, bad_val[bi] / tot, bad_pair[bi] / tot, bad_slice[bi] / tot))


if __name__ == "__main__":
    main()
