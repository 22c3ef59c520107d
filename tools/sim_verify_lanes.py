#!/usr/bin/env python3
"""CPU simulation of linesearch_verify_kernel's phase K on bench.py's synthetic data: how often does the insertion chain run?
(numpy only; no device, no oracle.)  Phase K visits a query's documents in storage order; a document whose key reaches the
(K+1)-th best of ANY candidate lane makes the whole wave run the min/max insertion chain -- about 3/4 of the kernel's cycles.
Two ideas from VERDICT r02 (use the 13 idle lanes) were priced with this before building anything:

  --mode pack   lanes 51..63 of a wave take 13 candidates of ANOTHER restart's line search (different feature, different
                base weights).  Their admissions are independent of the first group's, so the chain runs more often:
                measured 0.40 -> 0.49 chains per document for a 20 % saving in waves, i.e. ~5 % net -- below the bar.
  --mode lazy   every lane parks an admitted key in a one-deep pending slot and the chain runs only when some admitting lane's
                slot is already full.  Lanes admit the same documents too rarely for that to help: 0.49 -> 0.42 chains
                per document (the admitted key must still be parked: net ~0).

Neither was built; DESIGN.md section 4 records the numbers."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

K = 11  # K + 1 keys kept for NDCG@10


def cands(orig):
    c = [0.0]
    for sign in (-1.0, 1.0):
        step = 0.05 * sign
        if orig != 0 and abs(step) > 0.5 * abs(orig):
            step = 0.05 * abs(orig) * sign
        tot = step
        for _ in range(25):
            c.append(orig + tot)
            step *= 2
            tot += step
    return np.array(c)


def keys_of(X, seed, f):
    r = np.random.default_rng(seed)
    d = X.shape[1]
    w = r.uniform(-1, 1, d) * 0.2
    w[::8] += 1.0  # a model that has found the signal columns
    w /= np.abs(w).sum()
    A = X.astype(np.float64) @ w - X[:, f].astype(np.float64) * w[f]
    return A[:, None] + X[:, f].astype(np.float64)[:, None] * cands(w[f])[None, :]


def storage_order(y, s, e):
    idx = np.arange(s, e)
    return idx[np.lexsort((-idx, -y[s:e]))]  # gain desc, id desc (device.hpp)


def chains_eager(keys, y, starts):
    docs = runs = adm = 0
    for qi in range(len(starts) - 1):
        idx = storage_order(y, starts[qi], starts[qi + 1])
        kk = keys[idx]
        L = np.full((keys.shape[1], K), -np.inf)
        for t in range(len(idx)):
            a = kk[t] >= L[:, -1]
            if a.any():
                runs += 1
                adm += int(a.sum())
                for c in np.nonzero(a)[0]:
                    L[c, -1] = kk[t, c]
                    L[c][::-1].sort()
        docs += len(idx)
    return runs / docs, adm / docs


def chains_lazy(keys, y, starts):
    docs = runs = events = 0
    for qi in range(len(starts) - 1):
        idx = storage_order(y, starts[qi], starts[qi + 1])
        kk = keys[idx]
        nl = keys.shape[1]
        L = np.full((nl, K), -np.inf)
        pend = np.full(nl, -np.inf)
        full = np.zeros(nl, bool)
        for t in range(len(idx)):
            a = kk[t] >= L[:, -1]
            if a.any():
                events += 1
                if (a & full).any():
                    runs += 1
                    for c in np.nonzero(full)[0]:
                        if pend[c] >= L[c, -1]:
                            L[c, -1] = pend[c]
                            L[c][::-1].sort()
                    full[:] = False
                pend[a] = kk[t][a]
                full |= a
        runs += 1  # the flush at the end of the query
        docs += len(idx)
    return runs / docs, events / docs


def chains_primed(keys, y, starts, base_key, xf, with_xf, kprime=K):
    """every lane's list is first filled with the keys of a PRIMING SET of the query's documents -- the
    kprime best under the restart's current weights (= what the previous tick's accepted candidate ranked first), optionally
    also the K documents with the largest and the K with the smallest x_f (what the extreme candidates rank first) -- by
    running the chain for each of them; the main pass then skips those documents and admits against thresholds that are
    already high.  Any priming set is safe (a wrong one only admits more).  Returns (chains per document incl. the priming
    inserts, priming documents per document, admitted-after-priming chains per document)."""
    docs = runs = primed = 0
    for qi in range(len(starts) - 1):
        idx = storage_order(y, starts[qi], starts[qi + 1])
        kk = keys[idx]
        n = len(idx)
        S = set(np.argsort(-base_key[idx], kind="stable")[:kprime].tolist())
        if with_xf:
            o = np.argsort(xf[idx], kind="stable")
            S |= set(o[:K].tolist()) | set(o[-K:].tolist())
        S = sorted(S)
        L = np.full((keys.shape[1], K), -np.inf)
        for t in S:  # priming: one chain per document, no test
            for c in range(keys.shape[1]):
                if kk[t, c] >= L[c, -1]:
                    L[c, -1] = kk[t, c]
                    L[c][::-1].sort()
        primed += len(S)
        inS = np.zeros(n, bool)
        inS[S] = True
        for t in range(n):
            if inS[t]:
                continue
            a = kk[t] >= L[:, -1]
            if a.any():
                runs += 1
                for c in np.nonzero(a)[0]:
                    L[c, -1] = kk[t, c]
                    L[c][::-1].sort()
        docs += n
    return (runs + primed) / docs, primed / docs, runs / docs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["pack", "lazy", "prime"], default="pack")
    ap.add_argument("--docs", type=int, default=40000)
    ap.add_argument("--queries", type=int, default=330)
    args = ap.parse_args()
    X, y, qid = bench.gen_mslr_shaped(20250929, args.docs, 136, args.queries)
    starts = np.r_[0, np.nonzero(np.diff(qid))[0] + 1, len(qid)]
    k1, k2 = keys_of(X, 1, 3), keys_of(X, 2, 42)
    if args.mode == "prime":
        # per-document wave-instructions of the kernel: 3 (phase S) + 4.5 (word read, FMA, class insert, admission test) + 13 per
        # chain (the measured 10.4 at 0.45 chains per document); priming adds a gather of the priming words (~2 per primed document)
        def insts(chains, primed=0.0):
            return 7.5 + 13.0 * chains + 2.0 * primed
        for name, seed, f in (("feature 3 (sparse column)", 1, 3), ("feature 42 (heavy tail)", 2, 42), ("feature 8 (signal, uniform)", 3, 8), ("feature 5 (integer, ties)", 4, 5)):
            r = np.random.default_rng(seed)
            w = r.uniform(-1, 1, 136) * 0.2
            w[::8] += 1.0
            w /= np.abs(w).sum()
            base = X.astype(np.float64) @ w
            xf = X[:, f].astype(np.float64)
            keys = (base - xf * w[f])[:, None] + xf[:, None] * cands(w[f])[None, :]
            e, _ = chains_eager(keys, y, starts)
            p1 = chains_primed(keys, y, starts, base, xf, False)
            p2 = chains_primed(keys, y, starts, base, xf, True)
            print("%-28s chains per document: now %.3f | primed with the current top-%d %.3f (priming %.3f + admitted %.3f) | + top / bottom %d by x_f %.3f (priming %.3f + admitted %.3f)"
                  % (name, e, K, p1[0], p1[1], p1[2], K, p2[0], p2[1], p2[2]))
            print("%-28s modelled wave-instructions per document: now %.2f | primed %.2f (%+.0f %%) | + x_f extremes %.2f (%+.0f %%)" % (
                "", insts(e), insts(p1[0], p1[1]), 100 * (insts(p1[0], p1[1]) / insts(e) - 1), insts(p2[0], p2[1]), 100 * (insts(p2[0], p2[1]) / insts(e) - 1)))
    elif args.mode == "pack":
        print("chains per document: 51 lanes of one group %.3f | of another %.3f | 13 lanes of the other %.3f | 51 + 13 packed %.3f" % (
            chains_eager(k1, y, starts)[0], chains_eager(k2, y, starts)[0], chains_eager(k2[:, :13], y, starts)[0],
            chains_eager(np.concatenate([k1, k2[:, :13]], axis=1), y, starts)[0]))
    else:
        e, adm = chains_eager(k1, y, starts)
        l, ev = chains_lazy(k1, y, starts)
        print("chains per document: eager %.3f (%.1f lanes admit per document) | one-deep pending slot %.3f (documents that still park a key: %.3f)" % (e, adm, l, ev))


if __name__ == "__main__":
    main()
