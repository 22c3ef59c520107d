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

Neither was built; DESIGN.md section 4 records the numbers.

Round 6 modes model the kernel as it is now -- every (query, tile) segment staged twice, by R descending and by x_f
descending, lanes walking the copy and direction of their class (near / above / below the current weight), the first K + 1
documents of a query's first segment sorted by a network instead of inserted -- and price three more ideas:

  --mode defer  VERDICT r05 "Next" 2: an admitting lane parks its key in a pending queue of Q keys and the chain runs only
                when some lane's queue is full (and at a query's end).  The premise was "one to three lanes admit per chain
                run"; the model says 4-16 of 51 do (lanes of one class see the same document at the same step and admit it
                together) and the busiest lane of a query admits 12-23 times -- as often as the chain runs for the whole wave
                (17-34 times per 120-document query).  Q = 1 / 2 / 4 take 5-20 % off the chain runs while EVERY admitted
                document still pays for parking its key (key bits, two v_cndmask, the queue test): 3.5 -> 3.9 and 5.0 -> 5.6
                modelled instructions per document on two of the features, worse on all four.  Not built
                (profiles/r06_sim_defer.txt).
  --mode tiles  what a query cut into several segments costs: every further segment arrives best-first and re-admits into
                every lane's list.  64-position tiles cut a 120-document query into 2.85 segments; 128-position walk tiles
                cut at query boundaries (built: kernels_order.inc) into 1.45.
  --mode depth  where in the list the highest insertion of a chain run lands (nearly uniform): the chain in three parts
                instead of two (built).

Calibration: the kernel now counts its chain runs (bench.py verify.chain_runs_per_visit).  Hardware: 0.149 per (document,
group) visit on the headline data with walk tiles, 0.209 in storage order; this model says 0.14-0.28 (mean 0.21) for its
four features with walk tiles -- it over-counts by ~1.4x (its weights are not a trained model's), so its DIFFERENCES are
upper bounds: walk tiles were modelled at -29 % chain runs and measured at -4 % instructions (profiles/r06_walk_tiles_ab.txt)."""
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


# ---------------------------------------------------------------------------------------------- round 6: the kernel as it is

def model_weights(seed, d=136, trained=True):
    r = np.random.default_rng(seed)
    if not trained:
        w = r.uniform(-1, 1, d)
    else:
        w = r.uniform(-1, 1, d) * 0.2
        w[::8] += 1.0  # a model that has found the signal columns
    return w / np.abs(w).sum()


class Walk:
    """The resident walk of one line group over a dataset: per (query, tile) segment the lanes' document sequences and keys."""

    def __init__(self, X, y, qid, seed, f, ks, trained=True):
        self.ks = ks
        self.y, self.starts = y, np.r_[0, np.nonzero(np.diff(qid))[0] + 1, len(qid)]
        w = model_weights(seed, X.shape[1], trained)
        colstd = X.std(axis=0).astype(np.float64)
        self.base = X.astype(np.float64) @ w
        self.xf = X[:, f].astype(np.float64)
        self.cw = cands(w[f])
        delta = self.cw - w[f]
        gthr = np.sqrt(((w * colstd) ** 2).sum()) / colstd[f] if colstd[f] > 0 else np.inf
        nd = len(y)
        mode = (1 if (self.xf == self.xf.max()).sum() > 0.1 * nd else 0) | (2 if (self.xf == self.xf.min()).sum() > 0.1 * nd else 0)
        self.near = ~(np.abs(delta) >= gthr) | np.where(delta > 0, bool(mode & 1), bool(mode & 2))  # device_dataset.inc: gthr / gmode
        self.below = ~self.near & (delta < 0)
        self.above = ~self.near & ~self.below
        self.A = self.base - self.xf * w[f]

    def segments(self, tiling, T):
        """yields (query index, first segment of the query?, keys [lanes][documents of the segment in each lane's visiting order])"""
        pos, room = 0, 0
        nl = len(self.cw)
        for qi in range(len(self.starts) - 1):
            idx = storage_order(self.y, self.starts[qi], self.starts[qi + 1])
            n, done, first = len(idx), 0, True
            if tiling == "packed" and n > room:
                room = T if n <= T else 0  # a new walk tile (long queries are cut every T documents from a fresh one)
            while done < n:
                if tiling == "fixed":
                    seg_n = min(n - done, T - ((pos + done) % T))
                elif tiling == "global":
                    seg_n = n
                elif n <= T:
                    seg_n, room = n, room - n
                else:
                    seg_n, room = min(n - done, T), 0
                sidx = idx[done:done + seg_n]
                ro = sidx[np.argsort(-self.base[sidx], kind="stable")]
                xo = sidx[np.argsort(-self.xf[sidx], kind="stable")]
                seq = np.empty((nl, seg_n), int)
                seq[self.near], seq[self.above], seq[self.below] = ro, xo, xo[::-1]
                yield qi, first, self.A[seq] + self.xf[seq] * self.cw[:, None]
                first = False
                done += seg_n
            pos += n


def run_walk(wk, tiling, T, Q=0):
    """Chain statistics of one walk.  Q = 0: the kernel's eager chain; Q > 0: a pending queue of Q keys per lane."""
    ks = wk.ks
    st = dict(docs=0, segs=0, ev=0, adm=0, chains=0, up=0, flush=0, maxlane=0, nq=0, depth=np.zeros(ks + 1, int))
    nl = len(wk.cw)
    L, pend, lane_adm, cur = None, None, None, -1

    def flush():
        m = max(len(p) for p in pend)
        if m == 0:
            return
        st["flush"] += 1
        for i in range(m):
            st["chains"] += 1
            for c, p in enumerate(pend):
                if i < len(p) and p[i] >= L[c, -1]:
                    L[c, -1] = p[i]
                    L[c] = -np.sort(-L[c])
        for p in pend:
            p.clear()

    for qi, first, keys in wk.segments(tiling, T):
        if qi != cur:
            if cur >= 0:
                if Q:
                    flush()
                st["maxlane"] += lane_adm.max()
            cur, L, pend, lane_adm = qi, np.full((nl, ks), -np.inf), [[] for _ in range(nl)], np.zeros(nl, int)
            st["nq"] += 1
        seg_n, t0 = keys.shape[1], 0
        st["docs"] += seg_n
        st["segs"] += 1
        if first and seg_n >= ks:  # the fill network
            L = -np.sort(-keys[:, :ks], axis=1)
            t0 = ks
        for t in range(t0, seg_n):
            k = keys[:, t]
            a = k >= L[:, -1]
            if not a.any():
                continue
            st["ev"] += 1
            st["adm"] += int(a.sum())
            lane_adm += a
            if Q == 0:
                st["chains"] += 1
                top = int((L >= k[:, None]).sum(axis=1)[a].min())  # highest insertion position over the admitting lanes
                st["depth"][min(top, ks)] += 1
                if top < ks // 2:
                    st["up"] += 1
                for c in np.nonzero(a)[0]:
                    L[c, -1] = k[c]
                    L[c] = -np.sort(-L[c])
            else:
                if any(len(pend[c]) >= Q for c in np.nonzero(a)[0]):
                    flush()
                for c in np.nonzero(a)[0]:
                    pend[c].append(k[c])
    if cur >= 0:
        if Q:
            flush()
        st["maxlane"] += lane_adm.max()
    return st


R6_FEATURES = ((3, 1, "sparse"), (42, 2, "heavy tail"), (8, 3, "signal, uniform"), (5, 4, "integer"))


def main_r6(args):
    X, y, qid = bench.gen_mslr_shaped(20250929, args.docs, 136, args.queries, args.data)
    ks = args.keys
    for f, seed, what in R6_FEATURES:
        wk = Walk(X, y, qid, seed, f, ks, trained=not args.random_weights)
        cls = "near %d / above %d / below %d lanes" % (wk.near.sum(), wk.above.sum(), wk.below.sum())
        if args.mode == "defer":
            e = run_walk(wk, "packed", 128, 0)
            d = e["docs"]
            print("%-8s feature %-3d (%s; %s): eager %.3f chain runs per document, %.1f lanes admit per run, busiest lane %.1f admissions per query" % (
                args.data, f, what, cls, e["chains"] / d, e["adm"] / max(1, e["ev"]), e["maxlane"] / e["nq"]))
            for Q in (1, 2, 4):
                s = run_walk(wk, "packed", 128, Q)
                print("         Q = %d: keys parked on %.3f documents, %.3f chain runs per document in %.3f flushes" % (
                    Q, s["ev"] / d, s["chains"] / d, s["flush"] / d))
        elif args.mode == "tiles":
            row = []
            for name, til, T in (("64 fixed", "fixed", 64), ("128 fixed", "fixed", 128), ("128 at query boundaries", "packed", 128),
                                 ("192 at query boundaries", "packed", 192), ("whole queries", "global", 0)):
                s = run_walk(wk, til, T, 0)
                row.append("%s: %.3f runs (%.3f reach the upper half), %.2f segments per query" % (name, s["chains"] / s["docs"], s["up"] / s["docs"], s["segs"] / s["nq"]))
            print("%-8s feature %-3d (%s)\n         " % (args.data, f, what) + "\n         ".join(row))
        else:
            s = run_walk(wk, "packed", 128, 0)
            h = s["depth"][:ks] / max(1, s["depth"][:ks].sum())
            t1, t2 = ks // 3, (2 * ks) // 3
            halves = 2 + (ks - ks // 2) * 2 - 1 + h[:ks // 2].sum() * (ks // 2) * 2
            thirds = 2 + (ks - t2) * 2 - 1 + h[:t2].sum() * (1 + (t2 - t1) * 2) + h[:t1].sum() * t1 * 2
            print("%-8s feature %-3d (%s): %.3f runs per document; highest insertion position, share per slot 0..%d: %s" % (
                args.data, f, what, s["chains"] / s["docs"], ks - 1, " ".join("%.2f" % v for v in h)))
            print("         instructions per run: halves %.1f, thirds %.1f" % (halves, thirds))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["pack", "lazy", "prime", "defer", "tiles", "depth"], default="pack")
    ap.add_argument("--docs", type=int, default=40000)
    ap.add_argument("--queries", type=int, default=330)
    ap.add_argument("--data", default="mslr", choices=bench.DATA_KINDS, help="round-6 modes: the data kind")
    ap.add_argument("--keys", type=int, default=K, help="round-6 modes: keys per list (K + 1 = 11 at depth 10; 14 on the tie kinds)")
    ap.add_argument("--random-weights", action="store_true", help="round-6 modes: an untrained model (the start of a restart)")
    args = ap.parse_args()
    if args.mode in ("defer", "tiles", "depth"):
        main_r6(args)
        return
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
