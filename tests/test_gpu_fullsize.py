"""Parity at BASELINE.json's full sizes through size-independent properties (the CPU oracle is
only run on a sample of queries here; the small-size tests hold the bit-exact comparisons)."""
import os

import numpy as np
import pytest

import bench
import fastrank_amd as fr
from fastrank_amd import native
from oracle import pyoracle as o

pytestmark = pytest.mark.gpu


_CACHE = {}


def _shape(name):
    if name not in _CACHE:
        n, d, q, seed = bench.SHAPES[name]
        X, y, qid = bench.gen_mslr_shaped(seed, n, d, q)
        _CACHE[name] = (name, X, y, qid, fr.CDataset.from_numpy(X, y, qid))
    return _CACHE[name]


@pytest.fixture(scope="module", params=["10k", "30k"])
def big(request):
    return _shape(request.param)


@pytest.fixture(scope="module")
def big30():
    return _shape("30k")


def _groups(rng, d, G):
    feats, bases, cands = [], [], []
    for _ in range(G):
        w = rng.uniform(-1, 1, d)
        w /= np.abs(w).sum()
        f = int(rng.integers(0, d))
        feats.append(f), bases.append(w), cands.append(o.ca_candidates(w[f], 0.05, 2.0, 25))
    return feats, np.asarray(bases), cands


def test_fused_kernel_properties_at_full_size(big):
    name, X, y, qid, g = big
    rng = np.random.default_rng(101)
    feats, bases, cands = _groups(rng, X.shape[1], 2)
    means, pq = native.evaluate_candidates(g, "ndcg@10", feats, bases, cands, per_query=True)
    nq = native.num_queries(g)
    assert pq.shape == (nq, 2 * 64)
    for gi in range(2):
        cols = pq[:, gi * 64: gi * 64 + 51]
        assert np.isfinite(cols).all() and cols.min() >= 0.0 and cols.max() <= 1.0
        # fixed summation shape: 256-query segments, then the partials (device.hip MEAN_SEG)
        for ci in (0, 17, 50):
            v = cols[:, ci]
            seg = [np.add.reduce(np.concatenate([[0.0], v[s:s + 256]])) for s in range(0, nq, 256)]
            # np.add.reduce is pairwise; rebuild the exact sequential sums in Python floats
            tot = 0.0
            for s in range(0, nq, 256):
                part = 0.0
                for t in v[s:s + 256].tolist():
                    part += t
                tot += part
            assert means[gi][ci] == tot / nq
    # cross-check against the independent general path (ordered scoring kernel + LDS bitonic sort)
    sel = [(0, 0), (0, 26), (1, 50)]
    for gi, ci in sel:
        w = bases[gi].copy()
        w[feats[gi]] = cands[gi][ci]
        m = fr.CModel.from_dict({"Linear": {"weights": w.tolist()}})
        qids, vals = native.evaluate_dense(m, g, "ndcg@10")
        assert np.array_equal(vals, pq[:, gi * 64 + ci]), (name, gi, ci)
    # dir-0 candidate == the base weights with that coordinate zeroed (coordinate_ascent.rs:152-155)
    assert cands[0][0] == 0.0


def test_rank_order_is_sorted_and_a_permutation_at_full_size(big):
    name, X, y, qid, g = big
    rng = np.random.default_rng(103)
    w = rng.uniform(-1, 1, X.shape[1])
    m = fr.CModel.from_dict({"Linear": {"weights": w.tolist()}})
    scores = native.predict_scores_dense(m, g, len(y))
    ids, offs = native.rank_order(m, g)
    assert np.array_equal(np.sort(ids), np.arange(len(y), dtype=np.uint32))
    s = scores[ids]
    gains = y[ids].astype(np.float32)
    same_q = np.ones(len(ids), dtype=bool)
    same_q[offs[1:-1].astype(np.int64)] = False  # first element of every query but the first
    same_q[0] = False
    prev_s, prev_g, prev_id = np.roll(s, 1), np.roll(gains, 1), np.roll(ids, 1)
    ok = (prev_s > s) | ((prev_s == s) & ((prev_g < gains) | ((prev_g == gains) & (prev_id < ids))))
    assert ok[same_q].all(), "score desc, gain asc, id asc within every query"
    assert np.array_equal(qid[ids[offs[:-1].astype(np.int64)]], qid[ids[(offs[1:] - 1).astype(np.int64)]])


def test_sampled_queries_match_the_oracle_at_full_size(big):
    name, X, y, qid, g = big
    rng = np.random.default_rng(107)
    feats, bases, cands = _groups(rng, X.shape[1], 1)
    _, pq = native.evaluate_candidates(g, "ndcg@10", feats, bases, cands, per_query=True)
    uq = np.unique(qid)
    pick = np.sort(rng.choice(len(uq), 150, replace=False))
    mask = np.isin(qid, uq[pick])
    c = o.Dataset(X[mask], y[mask], qid[mask])
    # device query order is first appearance = ascending qid here, same as the oracle's
    for ci in (0, 9, 33, 50):
        w = bases[0].copy()
        w[feats[0]] = cands[0][ci]
        exp, err = c.metric_from_scores("ndcg@10", c.score_linear(w))
        assert err == 0
        assert np.array_equal(pq[pick, ci], exp), (name, ci)
    m = fr.CModel.from_dict({"Linear": {"weights": bases[0].tolist()}})
    for measure in ("map", "mrr", "ndcg"):
        _, vals = native.evaluate_dense(m, g, measure)
        exp, _ = c.metric_from_scores(measure, c.score_linear(bases[0]))
        assert np.array_equal(vals[pick], exp), (name, measure)


def test_verify_path_equals_exact_kernel_at_full_size(big, monkeypatch):
    """Bound-and-verify (default) against the exact kernel alone (FR_LS_EXACT=1) on the whole matrix:
    every per-query NDCG@10 of every candidate, and a stretch of real training (resident sums, incremental
    updates), must agree bit for bit."""
    name, X, y, qid, g = big
    rng = np.random.default_rng(103)
    feats, bases, cands = _groups(rng, X.shape[1], 4)
    means_v, pq_v = native.evaluate_candidates(g, "ndcg@10", feats, bases, cands, per_query=True)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "ndcg@10"
    req.params.seed, req.params.quiet, req.params.num_restarts = 5, True, 4
    run = native.CoordinateAscentRun(g, req)
    run.step(25)
    st_v = run.state()
    run.close()
    assert st_v["stats"]["verify_pairs"] > 0 or os.environ.get("FR_LS_EXACT")
    monkeypatch.setenv("FR_LS_EXACT", "1")
    means_e, pq_e = native.evaluate_candidates(g, "ndcg@10", feats, bases, cands, per_query=True)
    run = native.CoordinateAscentRun(g, req)
    run.step(25)
    st_e = run.state()
    run.close()
    assert st_e["stats"]["verify_pairs"] == 0
    for gi in range(4):  # (columns beyond a group's 51 candidates are not defined)
        assert np.array_equal(pq_v[:, gi * 64: gi * 64 + 51], pq_e[:, gi * 64: gi * 64 + 51])
    for a, b in zip(means_v, means_e):
        assert np.array_equal(a, b)
    assert st_v["restarts"] == st_e["restarts"]
    assert st_v["stats"]["useful_evals"] == st_e["stats"]["useful_evals"]


@pytest.mark.parametrize("kind", ["mslr", "tiesmix"])
def test_resident_training_equals_the_oracle_at_the_30k_shape(kind):
    """The kernel bench.py times (linesearch_verify_kernel on RESIDENT sums, default path) against the ORACLE, not
    against another HIP kernel, on the whole 3.8 M x 136 matrix: after 25 resident ticks every restart's best_score
    must be what the CPU restatement of evaluate_mean (src/evaluators.rs:173-224, src/dense_dataset.rs:67-76) gives
    for that restart's best weights, bit for bit (oracle in the HIP path's 256-query summation shape).  tiesmix =
    duplicate rows and integer columns, i.e. the variant with duplicate groups."""
    n, d, q, seed = bench.SHAPES["30k"]
    if kind == "mslr":
        _, X, y, qid, g = _shape("30k")
    else:
        X, y, qid = bench.gen_mslr_shaped(seed, n, d, q, kind)
        g = fr.CDataset.from_numpy(X, y, qid)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "ndcg@10"
    req.params.seed, req.params.quiet, req.params.num_restarts = 5, True, 4
    run = native.CoordinateAscentRun(g, req)
    run.step(25)
    st = run.state()
    run.close()
    if not os.environ.get("FR_LS_EXACT"):
        assert st["stats"]["verify_pairs"] > 0 and st["stats"]["exact_groups"] * 2 < st["stats"]["groups"], st["stats"]
    c = o.Dataset(X, y, qid)
    o.set_mean_segment(o.DEVICE_MEAN_SEGMENT)
    try:
        for r in st["restarts"]:
            assert c.evaluate_mean("ndcg@10", np.asarray(r["weights"])) == r["score"], (kind, r["restart_id"])
    finally:
        o.set_mean_segment(0)


@pytest.mark.parametrize("kind", ["mslr", "tiesmix"])
def test_every_resident_verify_variant_runs_at_the_30k_shape(kind, monkeypatch):
    """Round 5: one variant of linesearch_verify_kernel (K = 10, K + 2 keys) faulted at the 30K shape only -- an inline-asm
    output overlapped the address register of the scalar load behind it, which shows when that load waits to issue, i.e.
    under load -- while every small-size parity test of it passed.  So every resident variant (depth 5 / 10 / 20 x list
    length K+1..K+4, without and with duplicate groups = tiesmix) takes a few ticks of real training on the whole matrix
    here, and every value it publishes is recomputed by the exact kernel (FR_VERIFY_AUDIT: bit for bit, 0 mismatches)."""
    n, d, q, seed = bench.SHAPES["30k"]
    if kind == "mslr":
        g = _shape("30k")[4]
    else:
        X, y, qid = bench.gen_mslr_shaped(seed, n, d, q, kind)
        g = fr.CDataset.from_numpy(X, y, qid)
    monkeypatch.setenv("FR_VERIFY_AUDIT", "1")
    for measure in ("ndcg@5", "ndcg@10", "ndcg@20"):
        for xs in ("1", "2", "3", "4"):
            monkeypatch.setenv("FR_VERIFY_XS", xs)
            req = fr.TrainRequest.coordinate_ascent()
            req.measure = measure
            req.params.seed, req.params.quiet, req.params.num_restarts = 11, True, 6
            run = native.CoordinateAscentRun(g, req)
            run.step(3)
            st = run.state()["stats"]
            run.close()
            if not os.environ.get("FR_LS_EXACT"):
                assert st["verify_pairs"] > 0 and st["audit_values"] > 0, (kind, measure, xs, st)
            assert st["audit_mismatches"] == 0, (kind, measure, xs, st)


# ------------------------------------------------- BASELINE.json configs[4]: 500 trees x 30K shape

def _forest(rng, X, ntrees, max_depth, p_leaf=0.05):
    """fid uniform, split = a uniform quantile of that column, leaves U[0,4)."""
    sample = X[rng.integers(0, X.shape[0], 4096)]

    def grow(depth):
        if depth >= max_depth or rng.random() < p_leaf:
            return {"LeafNode": float(rng.uniform(0, 4))}
        f = int(rng.integers(0, X.shape[1]))
        return {"FeatureSplit": {"fid": f, "split": float(np.quantile(sample[:, f], rng.random())),
                                 "lhs": grow(depth + 1), "rhs": grow(depth + 1)}}

    return [grow(1) for _ in range(ntrees)]


def _ens(trees, weights):
    return fr.CModel.from_dict({"Ensemble": {"weights": list(weights), "models": [{"DecisionTree": t} for t in trees]}})


def _check_slices(X, y, qid, trees, weights, got, slices):
    for lo, hi in slices:
        sub = o.Dataset(X[lo:hi], y[lo:hi], qid[lo:hi])
        assert np.array_equal(sub.score_ensemble(trees, weights), got[lo:hi]), (lo, hi)


def test_config5_500_trees_on_the_30k_shape(big30):
    """configs[4]: the 500-tree, depth <= 8 forest over all 3.8 M documents through fr_predict_scores_dense,
    bit-exact against the oracle (src/model.rs:64-84,104-112) on four disjoint 20 000-document slices --
    the first block, two interior ones that straddle kernel blocks, and the ragged last block."""
    name, X, y, qid, g = big30
    n = len(y)
    rng = np.random.default_rng(7)
    trees = _forest(rng, X, 500, 8)
    weights = [1.0] * len(trees)
    got = native.predict_scores_dense(_ens(trees, weights), g, n)
    assert np.isfinite(got).all()
    slices = [(0, 20000), (1_234_567, 1_254_567), (2_500_033, 2_520_033), (n - 20000, n)]
    _check_slices(X, y, qid, trees, weights, got, slices)
    # signed, non-unit tree weights take the same kernel (the products w_t*leaf are formed on the host)
    weights = rng.uniform(-1.0, 1.0, len(trees)).tolist()
    got = native.predict_scores_dense(_ens(trees, weights), g, n)
    _check_slices(X, y, qid, trees, weights, got, [(777_777, 787_777), (n - 10000, n)])


def test_config5_deep_forest_fallback_on_the_30k_shape(big30):
    """Trees outside the compact LDS encoding (depth 11, no early leaves: a level wider than 255 nodes)
    take the L2 fallback kernel; same bits at the full size."""
    name, X, y, qid, g = big30
    n = len(y)
    rng = np.random.default_rng(8)
    trees = _forest(rng, X, 6, 11, p_leaf=0.0) + _forest(rng, X, 10, 8)
    weights = rng.uniform(0.0, 1.0, len(trees)).tolist()
    got = native.predict_scores_dense(_ens(trees, weights), g, n)
    _check_slices(X, y, qid, trees, weights, got, [(0, 10000), (1_900_001, 1_910_001), (n - 10000, n)])


# ------------------------------- BASELINE.json configs[3] on one GPU: 256 restarts as 8 restart shards

def test_config4_256_restarts_as_8_shards_equal_the_unsharded_run(big30):
    """configs[3] shards 256 restarts over 8 GPUs (32 each) and exchanges (score, weights) once.  One GPU
    stands in for the eight: the 8 blocks of native.shard_bounds are trained one after the other with
    fr_train_model_shard, put through gather_restarts / select_model, and must give the very restarts
    (score, weights, per restart) and the very model of the unsharded 256-restart run
    (src/coordinate_ascent.rs:211-251: child seeds in restart order, last maximum wins)."""
    name, X, y, qid, g = big30
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "ndcg@10"
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 42, True, 256, 25
    whole = native.train_model_shard(g, req, 0, 256)
    assert [r["restart_id"] for r in whole["restarts"]] == list(range(256))
    parts = []
    for rank in range(8):
        b, e = native.shard_bounds(256, rank, 8)
        assert e - b == 32
        shard = native.train_model_shard(g, req, b, e)
        assert [r["restart_id"] for r in shard["restarts"]] == list(range(b, e))
        parts.extend(shard["restarts"])
    gathered = native.gather_restarts(parts, 256)
    assert gathered == whole["restarts"]
    a = native.select_model(gathered, False).to_dict()
    b = native.select_model(whole["restarts"], False).to_dict()
    assert a == b and list(a) == ["Linear"]
    best = max(r["score"] for r in gathered)
    last = [r for r in gathered if r["score"] == best][-1]
    assert a["Linear"]["weights"] == last["weights"]
    # and the plain entry point (train_model) returns that model too
    assert g.train_model(req).to_dict() == a
