import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, fastrank_amd as fr
from fastrank_amd import native
def run(lo, hi, copies_per_src=1, nq=24, seed=41, same_label=False):
    rng = np.random.default_rng(seed)
    lens = rng.integers(lo, hi, nq)
    qid = np.repeat(np.arange(1, len(lens) + 1, dtype=np.int64), lens)
    n = len(qid)
    X = rng.normal(0, 1, (n, 12)).astype(np.float32)
    y = rng.choice([0.0, 0.0, 1.0, 2.0, 3.0], n)
    start = 0
    for L in lens:
        k = int(L) // 5
        if copies_per_src == 0:
            src = dst = np.zeros(0, dtype=np.int64)
        elif copies_per_src == 1:   # disjoint pairs only: sources from the first half, destinations from the second
            src = start + rng.choice(L // 2, min(k, L // 2), replace=False)
            dst = start + L // 2 + rng.choice(L - L // 2, len(src), replace=False)
        else:
            src, dst = start + rng.integers(0, L, k), start + rng.integers(0, L, k)
        X[dst] = X[src]
        if same_label: y[dst] = y[src]
        start += int(L)
    g = fr.CDataset.from_numpy(X, y, qid)
    req = fr.TrainRequest.coordinate_ascent(); req.measure = "ndcg"
    p = req.params; p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 23, True, 3, 4
    st = native.train_model_shard(g, req, 0, 3)["stats"]
    print("lens %d..%d copies %s: redone %d of %d pairs" % (lo, hi, ("none" if copies_per_src == 0 else "pairs" if copies_per_src == 1 else "random") + (" same label" if same_label else ""), st["verify_redone"], st["verify_pairs"]))
run(30, 90, 0); run(30, 90, 1, same_label=True); run(30, 90, 1)
os.environ["FR_NO_DUP_GROUPS"] = "1"; run(30, 90, 1)
