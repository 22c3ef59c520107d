#!/usr/bin/env python3
"""VERDICT r04 next 5, second half: the size classes of fullrank_verify_kernel (depth-less NDCG, MAP, NDCG@>20) priced on
the bench's query-length distribution.  Per class: how many queries it holds, how full its keys are (the rest are padding
sentinels that still ride through every compare-exchange of the data-independent network), its share of the tick by the
device's own cost model (fv_class_cost, device_dataset.inc) -- and what a finer grid of classes could save at most: the cost
if every query paid for exactly its own length (the model interpolated at nl = len / pl), and the cost with a few more classes
(56, 72, 88, 112 keys per lane).  CPU only."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

CLASSES = [(16, 1), (32, 1), (48, 1), (64, 1), (80, 1), (96, 1), (64, 2), (80, 2), (96, 2), (64, 4), (80, 4), (96, 4), (64, 8), (80, 8),
           (96, 8), (64, 16), (80, 16), (96, 16), (64, 32)]
SORT_CE = {16: 63, 32: 191, 48: 384, 64: 543, 80: 849, 96: 1056}   # compare-exchanges of the in-lane networks (kernels_sortnet.inc)
MERGE_CE = {16: 32, 32: 80, 48: 144, 64: 192, 80: 304, 96: 336}
PENALTY = [1.0, 1.1, 1.25, 1.6, 2.0, 2.4]


def ce_interp(table, nl):
    """compare-exchanges of a (hypothetical) network over nl keys: n log^2 n through the known points"""
    ks = sorted(table)
    if nl in table:
        return table[nl]
    f = lambda n: n * np.log2(max(n, 2)) ** 2
    lo = max([k for k in ks if k <= nl], default=ks[0])
    hi = min([k for k in ks if k >= nl], default=ks[-1])
    if lo == hi:
        return table[lo] * f(nl) / f(lo)
    t = (f(nl) - f(lo)) / (f(hi) - f(lo))
    return table[lo] + t * (table[hi] - table[lo])


def cost(nl, pl):
    rounds = int(np.ceil(np.log2(pl))) if pl > 1 else 0
    c = 2.0 * nl + 2.0 * ce_interp(SORT_CE, nl) + 5.0 * nl + nl * pl
    for r in range(1, rounds + 1):
        c += 5.0 * r * nl + 2.0 * ce_interp(MERGE_CE, nl)
    return c * pl * PENALTY[rounds]


def best_class(length, classes):
    best = None
    for nl, pl in classes:
        if nl * pl >= length:
            c = cost(nl, pl)
            if best is None or c < best[0]:
                best = (c, nl, pl)
    return best


def main():
    n, d, q, seed = bench.SHAPES["30k"]
    # (only the query lengths of gen_mslr_shaped are needed)
    rng = np.random.default_rng(seed)
    lens = np.clip(rng.lognormal(np.log(100.0), 0.6, q), 1, 1300)
    lens = np.maximum(1, np.floor(lens * (n / lens.sum()))).astype(np.int64)
    lens = np.minimum(lens, 1300)
    per = {}
    tot = 0.0
    for L in lens:
        c, nl, pl = best_class(int(L), CLASSES)
        e = per.setdefault((nl, pl), [0, 0, 0.0])
        e[0] += 1
        e[1] += int(L)
        e[2] += c
        tot += c
    print("class (keys per lane x lanes)  queries  mean fill of its keys  share of the tick (cost model)")
    for (nl, pl), (cnt, docs, c) in sorted(per.items(), key=lambda kv: -kv[1][2]):
        print("  %3d x %-2d                    %6d   %5.1f %% (%.0f %% sentinels)   %5.1f %%" % (nl, pl, cnt, 100.0 * docs / (cnt * nl * pl), 100.0 - 100.0 * docs / (cnt * nl * pl), 100.0 * c / tot))
    # lower bound of any finer grid with the same lane counts: every query pays for nl = ceil(len / pl) keys per lane
    ideal = 0.0
    for L in lens:
        _, nl, pl = best_class(int(L), CLASSES)
        ideal += cost(max(1, int(np.ceil(L / pl))), pl)
    finer = CLASSES + [(56, 1), (72, 1), (88, 1), (56, 2), (72, 2), (88, 2), (112, 1), (72, 4), (88, 4)]
    ft = sum(best_class(int(L), finer)[0] for L in lens)
    print("whole tick by the model: now 100 %%; with classes of 56 / 72 / 88 / 112 keys added %.1f %%; if every query paid for exactly its own "
          "length (no sentinels at all) %.1f %%" % (100.0 * ft / tot, 100.0 * ideal / tot))


if __name__ == "__main__":
    main()
