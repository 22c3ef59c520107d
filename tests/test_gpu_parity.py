"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle and the
reference's golden known answers.  Bit-exact for scores, rank order and per-query metrics;
means within 1e-12 (and in practice bit-exact, since both sum sequentially in query order).

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import json
import os

import numpy as np
import pytest

import fastrank_amd as fr
from fastrank_amd import native
from oracle import pyoracle as o
from tests.conftest import GOLDEN, synth_dataset

pytestmark = pytest.mark.gpu


def _names(known):
    names = {int(k): v for k, v in known["feature_names"].items()}
    names[0] = "0"
    return names


@pytest.fixture(scope="module")
def small():
    X, y, qid = synth_dataset(7, 6000, 24, 60, max_len=700)
    return X, y, qid, fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)


@pytest.fixture(scope="module")
def mslr_small():
    """MSLR-shaped (136 features) but small enough for the CPU oracle."""
    X, y, qid = synth_dataset(11, 30000, 136, 250)
    return X, y, qid, fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)


def _device_query_order(ds_gpu, model, measure="ndcg"):
    qids, vals = native.evaluate_dense(model, ds_gpu, measure)
    return qids, vals


def test_native_library_is_loaded_and_gpu_visible():
    assert native.device_count() >= 1
    assert os.path.exists(fr.clib._build.LIB_PATH)


def test_linear_scores_bit_exact(mslr_small):
    X, y, qid, g, c = mslr_small
    rng = np.random.default_rng(3)
    for trial in range(3):
        w = rng.uniform(-1, 1, X.shape[1])
        if trial == 1:
            w[rng.random(len(w)) < 0.5] = 0.0
        m = fr.CModel.from_dict({"Linear": {"weights": w.tolist()}})
        got = native.predict_scores_dense(m, g)
        exp = c.score_linear(w)
        assert np.array_equal(got, exp), "ordered unfused f64 dot product must match bit for bit"
    # shorter weight vector: zip() stops at the shorter side (dense_dataset.rs:72)
    m = fr.CModel.from_dict({"Linear": {"weights": [0.5, -0.25, 2.0]}})
    assert np.array_equal(native.predict_scores_dense(m, g), c.score_linear([0.5, -0.25, 2.0]))
    # JSON form
    js = m.predict_scores(g)
    assert len(js) == len(y) and js[0] == c.score_linear([0.5, -0.25, 2.0])[0]


def test_single_feature_and_tree_models(small):
    X, y, qid, g, c = small
    m = fr.CModel.from_dict({"SingleFeature": {"fid": 5, "dir": -2.5}})
    assert np.array_equal(native.predict_scores_dense(m, g), c.score_single_feature(5, -2.5))
    rng = np.random.default_rng(5)

    def rand_tree(depth):
        if depth == 0 or rng.random() < 0.15:
            return {"LeafNode": float(rng.uniform(0, 4))}
        f = int(rng.integers(0, X.shape[1]))
        return {"FeatureSplit": {"fid": f, "split": float(np.quantile(X[:, f], rng.random())),
                                 "lhs": rand_tree(depth - 1), "rhs": rand_tree(depth - 1)}}

    trees = [rand_tree(7) for _ in range(40)]
    weights = rng.uniform(0.1, 1.0, len(trees)).tolist()
    ens = {"Ensemble": {"weights": weights, "models": [{"DecisionTree": t} for t in trees]}}
    got = native.predict_scores_dense(fr.CModel.from_dict(ens), g)
    assert np.array_equal(got, c.score_ensemble(trees, weights))
    # bare decision tree: the leaf value itself
    got1 = native.predict_scores_dense(fr.CModel.from_dict({"DecisionTree": trees[0]}), g)
    assert np.array_equal(got1, c.score_ensemble([trees[0]], [1.0]))
    # mixed ensemble (coordinate ascent's output_ensemble shape): sum_t w_t * linear_t(x), unfused
    lin = [rng.uniform(-1, 1, X.shape[1]) for _ in range(3)]
    mixed = {"Ensemble": {"weights": [0.3, 0.5, 0.2], "models": [{"Linear": {"weights": w.tolist()}} for w in lin]}}
    exp = np.zeros(len(y))
    for wt, w in zip([0.3, 0.5, 0.2], lin):
        exp = exp + wt * c.score_linear(w)
    assert np.array_equal(native.predict_scores_dense(fr.CModel.from_dict(mixed), g), exp)


def test_regression_tree_known_answer(known):
    # src/random_forest.rs:465-506
    t = known["regression_tree"]
    X = np.asarray(t["xs"], dtype=np.float32).reshape(-1, 1)
    y = np.asarray(t["ys"], dtype=np.float64)
    g = fr.CDataset.from_numpy(X, y, np.zeros(len(y), dtype=np.int64))
    tree = {"FeatureSplit": {"fid": 0, "split": 3.0, "lhs": {"LeafNode": 7.0},
                             "rhs": {"FeatureSplit": {"fid": 0, "split": 6.0, "lhs": {"LeafNode": 2.0},
                                                      "rhs": {"LeafNode": 12.0}}}}}
    pred = native.predict_scores_dense(fr.CModel.from_dict({"DecisionTree": tree}), g)
    assert np.max(np.abs(pred - y)) <= t["tolerance"]


def test_rank_order_bit_exact_including_ties(small):
    X, y, qid, g, c = small
    rng = np.random.default_rng(9)
    cases = [rng.uniform(-1, 1, X.shape[1]), np.zeros(X.shape[1])]
    w_int = np.zeros(X.shape[1])
    w_int[1] = 1.0  # integer-valued column: massive score ties -> gain/id tie-breaks decide
    cases.append(w_int)
    for w in cases:
        m = fr.CModel.from_dict({"Linear": {"weights": w.tolist()}})
        ids, offs = native.rank_order(m, g)
        _, exp_rank, err = c.metric_from_scores("ndcg", c.score_linear(w), want_rank=True)
        assert err == 0
        # both sides list queries in first-appearance order
        assert np.array_equal(offs, c.query_offsets())
        assert np.array_equal(ids, exp_rank), "per-query rank order must be bit-exact"


def test_rank_ties_known_answer(known):
    # src/evaluators.rs:61-79 through the device sort
    r = known["rank_ties"]
    ids = np.asarray(r["ids"])
    order = np.argsort(ids)
    X = np.zeros((6, 1), dtype=np.float32)  # instance id = row index; ids 1..5 used
    y = np.zeros(6)
    for s, gn, i in zip(r["scores"], r["gains"], r["ids"]):
        X[i, 0] = s
        y[i] = gn
    g = fr.CDataset.from_numpy(X[1:], y[1:], np.zeros(5, dtype=np.int64))
    got, _ = native.rank_order(fr.CModel.from_dict({"Linear": {"weights": [1.0]}}), g)
    assert (got + 1).tolist() == r["expected_order"]


@pytest.mark.parametrize("measure", ["ndcg@10", "ndcg@5", "ndcg", "ndcg@1", "ndcg@20", "ndcg@1000", "map", "mrr", "AP", "RR"])
def test_per_query_metrics_bit_exact(small, measure):
    X, y, qid, g, c = small
    rng = np.random.default_rng(13)
    for w in (rng.uniform(-1, 1, X.shape[1]), np.zeros(X.shape[1])):
        m = fr.CModel.from_dict({"Linear": {"weights": w.tolist()}})
        qids, got = native.evaluate_dense(m, g, measure)
        exp, err = c.metric_from_scores(measure, c.score_linear(w))
        assert err == 0
        assert qids == [str(int(q)) for q in c.query_ids()]
        assert np.array_equal(got, exp), measure
        by_q = g.evaluate(m, measure)
        assert by_q == dict(zip(qids, exp.tolist()))


def _ca_groups(rng, d, n_groups, iters=25):
    feats, bases, cands = [], [], []
    for _ in range(n_groups):
        w = rng.uniform(-1, 1, d)
        w /= np.abs(w).sum()
        f = int(rng.integers(0, d))
        feats.append(f)
        bases.append(w)
        cands.append(o.ca_candidates(w[f], 0.05, 2.0, iters))
    return feats, np.asarray(bases), cands


@pytest.mark.parametrize("measure", ["ndcg@10", "ndcg@5", "ndcg@20", "ndcg@3"])
def test_fused_linesearch_matches_oracle_per_query(mslr_small, measure):
    X, y, qid, g, c = mslr_small
    rng = np.random.default_rng(17)
    feats, bases, cands = _ca_groups(rng, X.shape[1], 3)
    feats[0] = 0  # no shared prefix
    feats[1] = X.shape[1] - 1  # everything is prefix
    cands[0] = o.ca_candidates(bases[0][0], 0.05, 2.0, 25)
    cands[1] = o.ca_candidates(bases[1][X.shape[1] - 1], 0.05, 2.0, 25)
    means, pq = native.evaluate_candidates(g, measure, feats, bases, cands, per_query=True)
    norms = c.default_norms(measure)
    for gi in range(len(feats)):
        for ci in (0, 1, 7, 25, 26, 50):
            w = bases[gi].copy()
            w[feats[gi]] = cands[gi][ci]
            exp, err = c.metric_from_scores(measure, c.score_linear(w), norms)
            assert err == 0
            assert np.array_equal(pq[:, gi * 64 + ci], exp), (measure, gi, ci)
            assert means[gi][ci] == pytest.approx(c.evaluate_mean(measure, w, norms), abs=1e-12)
            assert means[gi][ci] == float(np.add.reduce(np.concatenate([[0.0], exp]))) / len(exp) or \
                abs(means[gi][ci] - exp.mean()) < 1e-12


def test_fused_and_generic_paths_agree(small):
    """Two independent device implementations (register top-k vs LDS bitonic sort)."""
    X, y, qid, g, c = small
    rng = np.random.default_rng(19)
    feats, bases, cands = _ca_groups(rng, X.shape[1], 4, iters=6)
    fused = native.evaluate_candidates(g, "ndcg@10", feats, bases, cands)
    generic = native.evaluate_candidates(g, "ndcg@1000", feats, bases, cands)  # depth > 20 -> sort kernel
    exact = native.evaluate_candidates(g, "ndcg", feats, bases, cands)
    for gi in range(len(feats)):
        assert np.array_equal(generic[gi], exact[gi])  # depth 1000 > every query length
        for ci in range(len(cands[gi])):
            w = bases[gi].copy()
            w[feats[gi]] = cands[gi][ci]
            assert fused[gi][ci] == c.evaluate_mean("ndcg@10", w)
            assert exact[gi][ci] == c.evaluate_mean("ndcg", w)
    ap = native.evaluate_candidates(g, "map", feats, bases, cands)
    w = bases[2].copy()
    w[feats[2]] = cands[2][3]
    assert ap[2][3] == c.evaluate_mean("map", w)


def test_edge_shapes_singletons_long_queries_negative_gains():
    rng = np.random.default_rng(23)
    lens = [1, 1, 2, 63, 64, 65, 127, 128, 129, 1300, 5, 10, 11]
    qid = np.repeat(np.arange(100, 100 + len(lens), dtype=np.int64), lens)
    n = len(qid)
    perm = rng.permutation(n)  # documents of a query need not be contiguous
    qid = qid[perm]
    X = np.floor(rng.exponential(2.0, (n, 5))).astype(np.float32)  # many ties
    y = rng.choice([-1.0, 0.0, 0.0, 1.0, 2.0, 4.0], size=n)
    y[qid == 101] = 0.0  # a query with no relevant document counts as 0 in the mean
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    for w in ([1.0, 0.0, -1.0, 0.5, 0.0], [0.0] * 5):
        m = fr.CModel.from_dict({"Linear": {"weights": w}})
        for measure in ("ndcg@10", "ndcg", "map", "mrr"):
            _, got = native.evaluate_dense(m, g, measure)
            exp, _ = c.metric_from_scores(measure, c.score_linear(w))
            assert np.array_equal(got, exp), measure
        base = np.asarray([w], dtype=np.float64)
        cand = [np.asarray([0.0, -0.3, 0.7, 1.5])]
        for measure in ("ndcg@10", "ndcg@5"):
            means, pq = native.evaluate_candidates(g, measure, [2], base, cand, per_query=True)
            for ci, cv in enumerate(cand[0]):
                ww = np.asarray(w, dtype=np.float64).copy()
                ww[2] = cv
                exp, _ = c.metric_from_scores(measure, c.score_linear(ww))
                assert np.array_equal(pq[:, ci], exp), (measure, ci)
                assert means[0][ci] == c.evaluate_mean(measure, ww)


def test_golden_single_feature_ndcg5_numpy_and_ranksvm(known, trec):
    """The reference's own end-to-end known answers (tests/test_with_example_data.py:16-23,
    139-167): single-feature coordinate ascent, then mean NDCG@5 on the full dataset."""
    names = _names(known)
    rd = fr.CDataset.open_ranksvm(os.path.join(GOLDEN, "data", "trec_news_2018.train"),
                                  os.path.join(GOLDEN, "data", "trec_news_2018.features.json"))
    dense = fr.CDataset.from_numpy(trec["train_X"], trec["train_y"], trec["train_qid"])
    req = fr.TrainRequest.coordinate_ascent()
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 42, True, 1, 1
    p.step_base, p.normalize, p.init_random = 1.0, False, False
    name_to_index = rd.feature_name_to_index()
    for f in range(known["expected_d"]):
        single = rd.subsample_feature_names([names[f]])
        model = single.train_model(req)
        got = np.mean(list(rd.evaluate(model, "ndcg@5").values()))
        assert got == pytest.approx(known["single_feature_ndcg5"][names[f]], abs=5e-8)
        weights = model.to_dict()["Linear"]["weights"]
        for i, w in enumerate(weights):
            if i != name_to_index[names[f]]:
                assert w == 0.0
        # same thing through the numpy/DenseDataset entry point
        model_d = dense.subsample_feature_names([str(f)]).train_model(req)
        got_d = np.mean(list(dense.evaluate(model_d, "ndcg@5").values()))
        assert got_d == pytest.approx(known["single_feature_ndcg5"][names[f]], abs=5e-8)


def test_coordinate_ascent_trajectory_matches_oracle(trec):
    """Same seed -> same restarts, same accepted candidates, same final weights (bit-exact).
    (Parity with the *Rust* RNG stream is unpinned; this pins device-vs-oracle.)"""
    X, y, qid = trec["train_X"], trec["train_y"], trec["train_qid"]
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    for measure, iters in (("ndcg@5", 25), ("ndcg@10", 5), ("map", 3), ("ndcg", 3)):
        req = fr.TrainRequest.coordinate_ascent()
        req.measure = measure
        p = req.params
        p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 42, True, 4, iters
        shard = native.train_model_shard(g, req, 0, 4)
        exp_s, exp_w, exp_e, err = c.ca_learn(measure, p.to_dict(), threads=4)
        assert err == 0
        for r in shard["restarts"]:
            k = r["restart_id"]
            assert r["score"] == exp_s[k], (measure, k)
            assert r["weights"] == exp_w[k].tolist(), (measure, k)
        assert shard["stats"]["useful_evals"] == int(exp_e.sum())
        model = g.train_model(req)
        best = o.select_best(exp_s)
        assert model.to_dict() == {"Linear": {"weights": exp_w[best].tolist()}}


def test_coordinate_ascent_on_mslr_shape_matches_oracle(mslr_small):
    X, y, qid, g, c = mslr_small
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "ndcg@10"
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 42, True, 2, 25
    run = native.CoordinateAscentRun(g, req)
    assert run.step(6) == 6  # six feature ticks of both restarts
    st = run.state()
    run.close()
    assert st["stats"]["ticks"] == 6 and st["stats"]["path"] == "fused_linesearch"
    assert st["stats"]["raw_evals"] == 2 + 6 * 2 * 51
    # The oracle cannot be stopped after exactly 6 ticks, so compare through the model: the
    # device's best-so-far weights must evaluate (on the oracle) to the device's best score.
    for r in st["restarts"]:
        w = np.asarray(r["weights"])
        assert c.evaluate_mean("ndcg@10", w) == r["score"]


def test_coordinate_ascent_full_run_wide_matrix_matches_oracle():
    """Default hyper-parameters (25 steps, normalise, random init) to convergence on a 40-feature
    matrix with long queries: identical restarts, scores, weights and useful-eval counts."""
    X, y, qid = synth_dataset(31, 4000, 40, 25, max_len=400)
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "ndcg@10"
    p = req.params
    p.seed, p.quiet, p.num_restarts = 20250929, True, 3
    shard = native.train_model_shard(g, req, 0, 3)
    exp_s, exp_w, exp_e, err = c.ca_learn("ndcg@10", p.to_dict(), threads=3)
    assert err == 0
    for r in shard["restarts"]:
        k = r["restart_id"]
        assert r["score"] == exp_s[k] and r["weights"] == exp_w[k].tolist()
    assert shard["stats"]["useful_evals"] == int(exp_e.sum())
    # restart sharding returns the same restarts (child seeds drawn in order from the master)
    part = native.train_model_shard(g, req, 1, 3)
    assert [r["restart_id"] for r in part["restarts"]] == [1, 2]
    assert [r["score"] for r in part["restarts"]] == exp_s[1:].tolist()


def test_ensemble_output_and_last_max_selection(trec):
    X, y, qid = trec["train_X"], trec["train_y"], trec["train_qid"]
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "ndcg@5"
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations, p.output_ensemble = 7, True, 3, 3, True
    ens = g.train_model(req).to_dict()["Ensemble"]
    exp_s, exp_w, _, _ = c.ca_learn("ndcg@5", p.to_dict(), threads=3)
    assert ens["weights"] == exp_s.tolist()
    for k, member in enumerate(ens["models"]):
        w = exp_w[k] / np.abs(exp_w[k]).sum() if np.abs(exp_w[k]).sum() > 0 else exp_w[k]
        assert member["Linear"]["weights"] == w.tolist()
    tie = native.select_model([{"restart_id": 0, "score": 0.5, "weights": [1.0]},
                               {"restart_id": 2, "score": 0.5, "weights": [3.0]},
                               {"restart_id": 1, "score": 0.5, "weights": [2.0]}])
    assert tie.to_dict() == {"Linear": {"weights": [3.0]}}  # Iterator::max -> last maximum


def test_qrel_norms_and_sampled_evaluation(trec, qrel_dict):
    # tests/test_with_example_data.py:243-269
    X, y, qid = trec["train_X"], trec["train_y"], trec["train_qid"]
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    qrel = fr.CQRel.from_dict(qrel_dict)
    m = fr.CModel.from_dict({"Linear": {"weights": [0.0, 0.3, -0.2, 0.5, 0.1, 0.9]}})
    with_q = g.evaluate(m, "ndcg@5", qrel)
    without = g.evaluate(m, "ndcg@5")
    assert abs(np.mean(list(with_q.values())) - np.mean(list(without.values()))) < 1e-7
    w = np.array([0.0, 0.3, -0.2, 0.5, 0.1, 0.9])
    for measure in ("ndcg@5", "ndcg", "map"):
        exp, _ = c.metric_from_scores(measure, c.score_linear(w), c.qrel_norms(measure, qrel_dict))
        got = g.evaluate(m, measure, qrel)
        assert got == dict(zip((str(int(q)) for q in c.query_ids()), exp.tolist())), measure
    first_ten = sorted(without.keys())[:10]
    part = g.subsample_queries(first_ten)
    part_scores = part.evaluate(m, "ndcg@5")
    assert len(part_scores) == 10
    for q in first_ten:
        assert part_scores[q] == without[q]
    sparse = m.predict_scores(part)
    assert len(sparse) == part.num_instances()
    assert len(m.predict_dense_scores(part)) >= len(sparse)


def test_trecrun_errors_without_docids_and_nan_is_an_error(trec, tmp_path):
    X, y, qid = trec["train_X"], trec["train_y"], trec["train_qid"]
    g = fr.CDataset.from_numpy(X, y, qid)
    m = fr.CModel.from_dict({"Linear": {"weights": [1.0] * 6}})
    with pytest.raises(Exception, match="Dataset does not contain document ids"):
        g.predict_trecrun(m, str(tmp_path / "run.txt"))  # tests/test_with_example_data.py:271-279
    Xn = X.copy()
    Xn[3, 2] = np.nan
    gn = fr.CDataset.from_numpy(Xn, y, qid)
    with pytest.raises(Exception, match="NaN"):
        gn.evaluate(m, "ndcg@5")
    with pytest.raises(Exception, match="NaN"):
        native.evaluate_candidates(gn, "ndcg@5", [2], np.ones((1, 6)), [np.asarray([0.5, 1.0])])


def test_model_serialization_keeps_map(trec):
    # tests/test_with_example_data.py:203-214
    X, y, qid = trec["train_X"], trec["train_y"], trec["train_qid"]
    g = fr.CDataset.from_numpy(X, y, qid)
    req = fr.TrainRequest.coordinate_ascent()
    req.params.seed, req.params.quiet = 42, True
    model = g.train_model(req)
    a = g.evaluate(model, "map")
    b = g.evaluate(fr.CModel.from_dict(model.to_dict()), "map")
    assert a == b
    scores = model.predict_scores(g)
    assert 0 in scores and len(scores) - 1 in scores and len(scores) == len(y)
    stats = native.last_train_stats()
    assert stats["path"] in ("fused_linesearch", "fused_fullrank", "generic_sort")
    assert stats["useful_evals"] <= stats["raw_evals"] and stats["ticks"] > 0


def test_hip_event_profile_reports_hot_kernels(small):
    X, y, qid, g, c = small
    native.profile_reset()
    native.profile_enable(True)
    rng = np.random.default_rng(29)
    feats, bases, cands = _ca_groups(rng, X.shape[1], 2, iters=25)
    native.evaluate_candidates(g, "ndcg@10", feats, bases, cands)
    native.profile_enable(False)
    stats = native.profile_stats()
    if os.environ.get("FR_LS_EXACT"):  # (A/B runs of the suite with the exact kernel only)
        assert stats["linesearch_ndcg_kernel"]["launches"] == 1 and stats["linesearch_ndcg_kernel"]["total_ms"] > 0.0
        return
    # bound-and-verify launch first; the exact kernel only runs for the pairs it could not verify
    assert stats["linesearch_verify_kernel"]["launches"] == 1
    assert stats["linesearch_verify_kernel"]["total_ms"] > 0.0
    assert stats.get("linesearch_ndcg_kernel", {"launches": 0})["launches"] <= 1


def test_mean_summation_shape_many_queries():
    """More than 256 queries: the device sums 256-query segments, then the segment partials
    (device.hip MEAN_SEG).  Bit-exact against the oracle run with the same shape; within 1e-12 of
    the plain sequential sum (the reference's own order is unspecified)."""
    X, y, qid = synth_dataset(37, 9000, 12, 900, max_len=60)
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    rng = np.random.default_rng(41)
    feats, bases, cands = _ca_groups(rng, X.shape[1], 2, iters=4)
    fused = native.evaluate_candidates(g, "ndcg@10", feats, bases, cands)
    generic = native.evaluate_candidates(g, "map", feats, bases, cands)
    try:
        for gi in range(2):
            for ci in range(len(cands[gi])):
                w = bases[gi].copy()
                w[feats[gi]] = cands[gi][ci]
                o.set_mean_segment(o.DEVICE_MEAN_SEGMENT)
                assert fused[gi][ci] == c.evaluate_mean("ndcg@10", w)
                assert generic[gi][ci] == c.evaluate_mean("map", w)
                o.set_mean_segment(0)
                assert abs(fused[gi][ci] - c.evaluate_mean("ndcg@10", w)) < 1e-12
        # trajectories stay identical when both sides use the same summation shape
        o.set_mean_segment(o.DEVICE_MEAN_SEGMENT)
        req = fr.TrainRequest.coordinate_ascent()
        req.measure = "ndcg@10"
        p = req.params
        p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 5, True, 2, 6
        shard = native.train_model_shard(g, req, 0, 2)
        exp_s, exp_w, exp_e, err = c.ca_learn("ndcg@10", p.to_dict(), threads=2)
        for r in shard["restarts"]:
            assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist()
        assert shard["stats"]["useful_evals"] == int(exp_e.sum())
    finally:
        o.set_mean_segment(0)


def test_tiny_queries_many_boundaries_per_tile_and_odd_feature_count():
    """Queries of 1-3 documents: dozens of query boundaries inside one 64-document tile, 13
    features (not a multiple of the 4-feature tile group)."""
    rng = np.random.default_rng(43)
    lens = rng.integers(1, 4, size=400)
    qid = np.repeat(np.arange(7, 7 + len(lens), dtype=np.int64), lens)
    n = len(qid)
    X = np.round(rng.normal(0, 2, (n, 13))).astype(np.float32)
    y = rng.choice([0.0, 0.0, 1.0, 2.0, 3.5], size=n)
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    feats, bases, cands = _ca_groups(rng, 13, 3, iters=25)
    feats = [0, 12, 5]
    cands = [o.ca_candidates(bases[i][feats[i]], 0.05, 2.0, 25) for i in range(3)]
    try:
        o.set_mean_segment(o.DEVICE_MEAN_SEGMENT)
        for measure in ("ndcg@10", "ndcg@2"):
            means, pq = native.evaluate_candidates(g, measure, feats, bases, cands, per_query=True)
            for gi in range(3):
                for ci in (0, 1, 13, 50):
                    w = bases[gi].copy()
                    w[feats[gi]] = cands[gi][ci]
                    exp, _ = c.metric_from_scores(measure, c.score_linear(w))
                    assert np.array_equal(pq[:, gi * 64 + ci], exp), (measure, gi, ci)
                    assert means[gi][ci] == c.evaluate_mean(measure, w)
    finally:
        o.set_mean_segment(0)


def test_more_than_64_candidates_per_line_search(trec):
    """num_max_iterations = 40 -> 81 candidates per feature -> two 64-wide line groups."""
    X, y, qid = trec["train_X"], trec["train_y"], trec["train_qid"]
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "ndcg@10"
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations, p.step_scale = 3, True, 2, 40, 1.3
    shard = native.train_model_shard(g, req, 0, 2)
    exp_s, exp_w, exp_e, err = c.ca_learn("ndcg@10", p.to_dict(), threads=2)
    assert err == 0
    for r in shard["restarts"]:
        assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist()
    assert shard["stats"]["useful_evals"] == int(exp_e.sum())
    assert shard["stats"]["path"] == "fused_linesearch"


def test_infinite_features_take_the_general_path(trec):
    """inf * 0 = NaN would break zero-weight masking, so such datasets must not use the fused
    kernel; results still match the oracle (inf scores are legal, NaN scores are an error)."""
    X = trec["train_X"].copy()
    y, qid = trec["train_y"], trec["train_qid"]
    X[5, 1] = np.inf
    X[40, 4] = -np.inf
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    w = np.array([0.0, 0.4, 0.1, 0.2, 0.3, 0.5])
    base = w[None, :].copy()
    cands = [np.asarray([0.0, 0.1, 0.7])]
    means = native.evaluate_candidates(g, "ndcg@10", [2], base, cands)
    for ci, cv in enumerate(cands[0]):
        ww = w.copy()
        ww[2] = cv
        assert means[0][ci] == c.evaluate_mean("ndcg@10", ww)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "ndcg@10"
    req.params.seed, req.params.quiet, req.params.num_restarts, req.params.num_max_iterations = 1, True, 1, 2
    req.params.init_random = False
    req.params.normalize = False
    with pytest.raises(Exception, match="NaN"):
        g.train_model(req)  # dir 0 makes a weight 0.0 -> inf * 0 = NaN -> the reference panics too
    assert native.last_train_stats is not None


def test_ranksvm_file_and_numpy_entry_points_train_identically(trec, known):
    rd = fr.CDataset.open_ranksvm(os.path.join(GOLDEN, "data", "trec_news_2018.train"))
    dense = fr.CDataset.from_numpy(trec["train_X"], trec["train_y"], trec["train_qid"])
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "ndcg@5"
    req.params.seed, req.params.quiet, req.params.num_restarts, req.params.num_max_iterations = 9, True, 2, 4
    a, b = rd.train_model(req).to_dict(), dense.train_model(req).to_dict()
    assert a == b
    # a query-sampled view trains on exactly its queries (tests/test_with_example_data.py:106-137)
    subset = sorted(known["expected_queries"])[:12]
    part = rd.subsample_queries(subset)
    mask = np.isin(trec["train_qid"], [int(q) for q in subset])
    c = o.Dataset(trec["train_X"][mask], trec["train_y"][mask], trec["train_qid"][mask])
    shard = native.train_model_shard(part, req, 0, 2)
    exp_s, exp_w, _, _ = c.ca_learn("ndcg@5", req.params.to_dict(), threads=2)
    for r in shard["restarts"]:
        assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist()
    sparse = fr.CModel.from_dict(a).predict_scores(part)
    assert len(sparse) == part.num_instances() == int(mask.sum())


@pytest.mark.parametrize("measure", ["map", "mrr", "ndcg", "ndcg@50"])
def test_fullrank_line_search_matches_oracle_and_general_path(measure):
    """AP / RR / depth-less NDCG line search: scores kernel + rank-counting kernel, against the
    oracle per query and against the independent sort-based path (FR_FORCE_GENERIC)."""
    X, y, qid = synth_dataset(53, 5000, 20, 40, max_len=500)
    y[qid == 3] = 0.0          # a query without relevant documents
    y[(qid == 5) & (y == 0)] = -1.0  # negative gains contribute negative NDCG terms
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    rng = np.random.default_rng(59)
    feats, bases, cands = _ca_groups(rng, X.shape[1], 3, iters=25)
    feats[0], feats[1] = 0, X.shape[1] - 1
    cands[0] = o.ca_candidates(bases[0][0], 0.05, 2.0, 25)
    cands[1] = o.ca_candidates(bases[1][feats[1]], 0.05, 2.0, 25)
    means, pq = native.evaluate_candidates(g, measure, feats, bases, cands, per_query=True)
    os.environ["FR_FORCE_GENERIC"] = "1"
    try:
        generic = native.evaluate_candidates(g, measure, feats, bases, cands)
    finally:
        del os.environ["FR_FORCE_GENERIC"]
    norms = c.default_norms(measure)
    for gi in range(3):
        assert np.array_equal(means[gi], generic[gi]), (measure, gi)
        for ci in (0, 1, 12, 25, 26, 50):
            w = bases[gi].copy()
            w[feats[gi]] = cands[gi][ci]
            exp, err = c.metric_from_scores(measure, c.score_linear(w), norms)
            assert err == 0
            assert np.array_equal(pq[:, gi * 64 + ci], exp), (measure, gi, ci)
            assert means[gi][ci] == c.evaluate_mean(measure, w, norms)
    # all-tie candidates (zero weights): ranking decided by the gain/id tie-break alone
    z = np.zeros((1, X.shape[1]))
    m0, pq0 = native.evaluate_candidates(g, measure, [2], z, [np.asarray([0.0])], per_query=True)
    exp, _ = c.metric_from_scores(measure, np.zeros(len(y)), norms)
    assert np.array_equal(pq0[:, 0], exp)


def test_default_measure_training_uses_fullrank_path(trec):
    X, y, qid = trec["train_X"], trec["train_y"], trec["train_qid"]
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    req = fr.TrainRequest.coordinate_ascent()  # measure "ndcg" (no depth) is the reference's default
    assert req.measure == "ndcg"
    req.params.seed, req.params.quiet, req.params.num_restarts, req.params.num_max_iterations = 11, True, 2, 5
    shard = native.train_model_shard(g, req, 0, 2)
    assert shard["stats"]["path"] == "fused_fullrank"
    exp_s, exp_w, exp_e, err = c.ca_learn("ndcg", req.params.to_dict(), threads=2)
    for r in shard["restarts"]:
        assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist()
    assert shard["stats"]["useful_evals"] == int(exp_e.sum())


# ---------------------------------------------------------------- tree-ensemble kernel (config 5)

def _rand_tree(rng, X, depth, p_leaf=0.1, nfeat=None):
    nfeat = X.shape[1] if nfeat is None else nfeat
    if depth == 0 or rng.random() < p_leaf:
        return {"LeafNode": float(rng.uniform(-2, 4))}
    f = int(rng.integers(0, nfeat))
    col = X[:, min(f, X.shape[1] - 1)]
    return {"FeatureSplit": {"fid": f, "split": float(np.quantile(col, rng.random())),
                             "lhs": _rand_tree(rng, X, depth - 1, p_leaf, nfeat),
                             "rhs": _rand_tree(rng, X, depth - 1, p_leaf, nfeat)}}


def _ensemble(trees, weights):
    return fr.CModel.from_dict({"Ensemble": {"weights": list(weights), "models": [{"DecisionTree": t} for t in trees]}})


@pytest.mark.parametrize("shape", ["", "256,1", "256,2", "256,4", "192,2", "192,4", "128,2", "128,4", "64,4"])
def test_tree_kernel_block_shapes(mslr_small, shape, monkeypatch):
    """Every block shape of the LDS tree walk gives the oracle's scores bit for bit: 77 trees (so the
    last batches are ragged and padded), missing features (fid >= D reads 0.0), negative weights."""
    X, y, qid, g, c = mslr_small
    if shape:
        monkeypatch.setenv("FR_TREE_SHAPE", shape)
        monkeypatch.setenv("FR_TREE_RANK", "0")  # the f32 walk; "" is the default path (threshold ranks)
    rng = np.random.default_rng(21)
    trees = [_rand_tree(rng, X, 7, nfeat=X.shape[1] + 4) for _ in range(77)]
    weights = rng.uniform(-1.0, 1.0, len(trees)).tolist()
    got = native.predict_scores_dense(_ensemble(trees, weights), g)
    assert np.array_equal(got, c.score_ensemble(trees, weights))


def _tree_kernels_used(model, g):
    native.profile_reset()
    native.profile_enable(True)
    try:
        out = native.predict_scores_dense(model, g)
    finally:
        native.profile_enable(False)
    return out, {k for k in native.profile_stats() if k.startswith("tree_")}


def test_tree_rank_kernel(mslr_small, small):
    """The threshold-rank walk (kernels_treerank.inc) against the oracle, bit for bit, and that it is the kernel that
    ran: ragged last batches, missing features, negative weights; trees of depth 9 and 10 (fewer walks per thread);
    more than 255 and more than 511 distinct thresholds on a feature (tables staged two / one feature per round);
    forests outside the encoding (a leaf deeper than 10, > 1023 thresholds on a feature) take the f32 walks."""
    X, y, qid, g, c = mslr_small
    rng = np.random.default_rng(23)
    trees = [_rand_tree(rng, X, 7, nfeat=X.shape[1] + 4) for _ in range(77)]
    weights = rng.uniform(-1.0, 1.0, len(trees)).tolist()
    got, used = _tree_kernels_used(_ensemble(trees, weights), g)
    assert used == {"tree_rank_kernel"} and np.array_equal(got, c.score_ensemble(trees, weights))
    for depth, count in ((9, 21), (10, 7), (2, 300)):
        trees = [_rand_tree(rng, X, depth, p_leaf=0.03) for _ in range(count)]
        weights = rng.uniform(0.0, 1.0, count).tolist()
        got, used = _tree_kernels_used(_ensemble(trees, weights), g)
        assert used == {"tree_rank_kernel"} and np.array_equal(got, c.score_ensemble(trees, weights)), depth
    X, y, qid, g, c = small
    nf = X.shape[1]
    for count, path in ((6 * nf, "tree_rank_kernel"), (12 * nf, "tree_rank_kernel"), (24 * nf, "tree_ensemble_kernel")):
        # ~63 split nodes per tree, thresholds drawn from a continuous quantile: count * 63 / nf distinct per feature
        trees = [_rand_tree(rng, X, 6, p_leaf=0.0) for _ in range(count)]
        weights = [1.0] * count
        got, used = _tree_kernels_used(_ensemble(trees, weights), g)
        assert used == {path} and np.array_equal(got, c.score_ensemble(trees, weights)), count
    deep = [_rand_tree(rng, X, 11, p_leaf=0.0) for _ in range(2)]
    got, used = _tree_kernels_used(_ensemble(deep, [1.0, 0.5]), g)
    assert "tree_rank_kernel" not in used and np.array_equal(got, c.score_ensemble(deep, [1.0, 0.5]))


def test_wrong_answer_switches_do_nothing_in_the_shipped_library(mslr_small, monkeypatch):
    """FR_TREE_NOWALK / FR_TREE_NOSTAGE / FR_LS_DEBUG=1 skip a kernel phase to price the rest and return garbage: they
    are compiled in only with -DFR_PRICING.  In the default build the environment cannot change a score."""
    if "FR_PRICING" in os.environ.get("FR_BUILD_FLAGS", ""):
        pytest.skip("pricing build")
    X, y, qid, g, c = mslr_small
    rng = np.random.default_rng(31)
    trees = [_rand_tree(rng, X, 7) for _ in range(40)]
    weights = rng.uniform(-1.0, 1.0, len(trees)).tolist()
    for name in ("FR_TREE_NOWALK", "FR_TREE_NOSTAGE", "FR_LS_DEBUG"):
        monkeypatch.setenv(name, "1")
    got, used = _tree_kernels_used(_ensemble(trees, weights), g)
    assert used == {"tree_rank_kernel"} and np.array_equal(got, c.score_ensemble(trees, weights))
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "ndcg@10"
    req.params.seed, req.params.quiet, req.params.num_restarts = 3, True, 2
    run = native.CoordinateAscentRun(g, req)
    run.step(3)
    st = run.state()
    run.close()
    for r in st["restarts"]:
        assert c.evaluate_mean("ndcg@10", np.asarray(r["weights"])) == r["score"]


def test_tree_rank_kernel_slot_layouts(mslr_small):
    """Codes sit two slots to a 32-bit word and the constant slots follow the last feature's: forests over 1, 2, 3, 5
    and 7 distinct features (odd and even slot counts, the constant slots sharing / not sharing a word with a feature;
    float4 groups with one to four ranked features, searched deepest table first) against the oracle, bit for bit."""
    X, y, qid, g, c = mslr_small
    rng = np.random.default_rng(29)
    for feats in ([5], [2, 3], [0, 1, 6], [1, 4, 5, 6, 7], [0, 2, 3, 8, 9, 10, 11], list(range(9, 20))):
        def grow(depth):
            if depth == 0 or rng.random() < 0.08:
                return {"LeafNode": float(rng.uniform(-2, 4))}
            f = int(feats[rng.integers(0, len(feats))])
            # very different table depths inside one float4 group: feature ids divisible by 3 take few thresholds
            q = float(rng.integers(1, 4)) / 4.0 if f % 3 == 0 else float(rng.random())
            return {"FeatureSplit": {"fid": f, "split": float(np.quantile(X[:, f], q)), "lhs": grow(depth - 1), "rhs": grow(depth - 1)}}
        trees = [grow(6) for _ in range(41)]
        weights = rng.uniform(-1.0, 1.0, len(trees)).tolist()
        got, used = _tree_kernels_used(_ensemble(trees, weights), g)
        assert used == {"tree_rank_kernel"} and np.array_equal(got, c.score_ensemble(trees, weights)), feats


def test_tree_kernel_edge_forests(small):
    X, y, qid, g, c = small
    rng = np.random.default_rng(22)
    # 200 stumps and bare leaves: many trees per LDS batch
    stumps = [_rand_tree(rng, X, 1, p_leaf=0.2) for _ in range(200)]
    w = rng.uniform(0.0, 1.0, len(stumps)).tolist()
    assert np.array_equal(native.predict_scores_dense(_ensemble(stumps, w), g), c.score_ensemble(stumps, w))
    # deep, wide trees (a level wider than the compact encoding's 8-bit child offset): L2 fallback kernel
    deep = [_rand_tree(rng, X, 11, p_leaf=0.0) for _ in range(3)]
    assert np.array_equal(native.predict_scores_dense(_ensemble(deep, [1.0, 0.5, 0.25]), g),
                          c.score_ensemble(deep, [1.0, 0.5, 0.25]))
    # a single tree inside an ensemble is 0.0 + w * leaf; a bare DecisionTree is the leaf itself (-0.0 survives)
    t = {"FeatureSplit": {"fid": 0, "split": float(np.median(X[:, 0])), "lhs": {"LeafNode": -0.0}, "rhs": {"LeafNode": 1.5}}}
    bare = native.predict_scores_dense(fr.CModel.from_dict({"DecisionTree": t}), g)
    exp = c.score_ensemble([t], [1.0])
    assert np.array_equal(bare, exp) and np.array_equal(np.signbit(bare), np.signbit(np.where(X[:, 0] <= np.float32(t["FeatureSplit"]["split"]), -0.0, 1.5)))
    ens1 = native.predict_scores_dense(_ensemble([t], [2.0]), g)
    assert np.array_equal(ens1, 0.0 + 2.0 * exp) and not np.signbit(ens1).any()


def test_tree_kernel_thresholds_at_f32_boundaries():
    """f64(x) <= split with splits that are not f32 values or lie beyond the f32 range; x = +-inf, NaN, denormals."""
    vals = np.array([-np.inf, -3.4028235e38, -1.0, -1e-45, -0.0, 0.0, 1e-45, 0.1, 0.5, 1.0, 3.4028235e38, np.inf, np.nan],
                    dtype=np.float32)
    X = np.repeat(vals[:, None], 3, axis=1)
    y = np.zeros(len(vals))
    qid = np.zeros(len(vals), dtype=np.int64)
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    # (infinite / NaN splits cannot be written in the JSON model format, as with serde_json)
    splits = [0.1, float(np.float32(0.1)), float(np.nextafter(float(np.float32(0.1)), -1.0)), 0.0, -0.0, 1e300, -1e300,
              3.4028235e38, -3.4028235e38, 1e-46, -1e-46, 0.5]
    trees = [{"FeatureSplit": {"fid": i % 3, "split": s, "lhs": {"LeafNode": 1.0}, "rhs": {"LeafNode": 2.0}}} for i, s in enumerate(splits)]
    for t in trees:
        got = native.predict_scores_dense(fr.CModel.from_dict({"DecisionTree": t}), g)
        assert np.array_equal(got, c.score_ensemble([t], [1.0])), t
    w = [1.0] * len(trees)
    assert np.array_equal(native.predict_scores_dense(_ensemble(trees, w), g), c.score_ensemble(trees, w))


def test_query_longer_than_the_lds_sort():
    """One query of 20 000 documents (beyond the 8192-document LDS sort and the 2048-document
    rank-counting kernel): the general evaluator sorts it in global memory; the fused NDCG@k line
    search takes it as it is.  The reference has no length limit (src/evaluators.rs:186-224)."""
    rng = np.random.default_rng(61)
    lens = np.array([20000, 3, 700, 1, 9000])
    qid = np.repeat(np.arange(1, 1 + len(lens), dtype=np.int64), lens)
    n = len(qid)
    X = np.round(rng.normal(0, 3, (n, 6))).astype(np.float32)  # heavy score ties
    X[:, 5] = rng.normal(0, 1, n).astype(np.float32)
    y = rng.choice([0.0, 0.0, 0.0, 1.0, 2.0, 4.0], size=n)
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    w = np.array([0.4, -0.2, 0.1, 0.0, 0.3, 0.05])
    model = fr.CModel.from_dict({"Linear": {"weights": w.tolist()}})
    ids, offs = native.rank_order(model, g)
    _, exp_rank, err = c.metric_from_scores("ndcg", c.score_linear(w), want_rank=True)
    assert err == 0 and np.array_equal(ids, exp_rank)
    for measure in ("ndcg", "ndcg@10", "ndcg@5000", "map", "mrr"):
        exp, _ = c.metric_from_scores(measure, c.score_linear(w))
        got = g.evaluate(model, measure)
        assert got == dict(zip((str(int(q)) for q in c.query_ids()), exp.tolist())), measure
    req = fr.TrainRequest.coordinate_ascent()
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 3, True, 1, 2
    for measure, path in (("ndcg@10", "fused_linesearch"), ("map", "generic_sort")):
        req.measure = measure
        shard = native.train_model_shard(g, req, 0, 1)
        assert shard["stats"]["path"] == path
        exp_s, exp_w, _, err = c.ca_learn(measure, p.to_dict(), threads=1, max_evals_per_restart=0)
        assert err == 0
        assert shard["restarts"][0]["score"] == exp_s[0] and shard["restarts"][0]["weights"] == exp_w[0].tolist()


# ---------------------------------------------------------------- bound-and-verify line search

def _verify_path_on(resident_needed=False):
    """False when the environment forces the exact kernels (the suite is also run under FR_LS_EXACT=1,
    FR_LS_RESIDENT=0, ... as an A/B check: results must not change, only these path assertions do)."""
    if os.environ.get("FR_LS_EXACT"):
        return False
    if resident_needed and os.environ.get("FR_LS_RESIDENT", "1")[:1] == "0":
        return False
    return True


def _train_stats(g, req):
    shard = native.train_model_shard(g, req, 0, int(req.params.num_restarts))
    return shard, shard["stats"]


def test_verify_kernel_falls_back_on_ties_and_duplicates():
    """The hot path evaluates from approximate scores and keeps a value only if the order of the
    best documents is provable; queries with exactly tied or nearly tied scores must come from the
    exact kernel.  Duplicated documents (exact ties whose order is decided by gain / id), integer
    features and a single-feature model force that path; the trajectory must still be the oracle's."""
    rng = np.random.default_rng(71)
    X, y, qid = synth_dataset(73, 6000, 8, 60, max_len=300)
    X = np.round(X * 2).astype(np.float32)          # few distinct values per column
    dup = rng.integers(0, len(y), 1500)              # duplicate rows inside the same query
    for i in dup:
        j = i + 1 if i + 1 < len(y) and qid[i + 1] == qid[i] else i
        X[j] = X[i]
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "ndcg@10"
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 9, True, 2, 4
    shard, st = _train_stats(g, req)
    exp_s, exp_w, exp_e, err = c.ca_learn("ndcg@10", p.to_dict(), threads=2)
    assert err == 0
    for r in shard["restarts"]:
        assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist()
    assert st["path"] == "fused_linesearch"
    if _verify_path_on():
        assert st["verify_redone"] > 0, "this dataset must exercise the exact fallback"
    # per-query values of single candidates, including the all-ties candidate (zero weights)
    feats = [0, 3]
    bases = np.zeros((2, X.shape[1]))
    bases[1, 5] = 1.0
    cands = [np.asarray([0.0, 1.0, -1.0]), np.asarray([0.0, 0.5])]
    means, pq = native.evaluate_candidates(g, "ndcg@10", feats, bases, cands, per_query=True)
    for gi in range(2):
        for ci in range(len(cands[gi])):
            w = bases[gi].copy()
            w[feats[gi]] = cands[gi][ci]
            exp, _ = c.metric_from_scores("ndcg@10", c.score_linear(w))
            assert np.array_equal(pq[:, gi * 64 + ci], exp), (gi, ci)


def test_redo_list_longer_than_the_first_exact_launch(monkeypatch):
    """The exact kernel's first launch on the redo list has a fixed grid (8192 pairs; 512 here, compared with the count on
    the device); the rest of a longer list is recomputed after the results came back.  Every query here has
    duplicated documents on top, so every (query, group) pair of 700 queries x 3 groups is on the list."""
    monkeypatch.setenv("FR_REDO_GRID", "512")
    monkeypatch.setenv("FR_NO_DUP_GROUPS", "1")  # (duplicate groups would let the verify kernel keep these pairs)
    rng = np.random.default_rng(5)
    nq, per = 700, 14
    X = rng.random((nq * per, 6)).astype(np.float32)
    y = rng.integers(0, 3, nq * per).astype(np.float64)
    qid = np.repeat(np.arange(1, nq + 1), per)
    X[1::2] = X[0::2]  # pairs of identical rows: exact ties for every weight vector
    y[0::2] = 2.0      # the duplicated rows are the relevant ones, so the ties sit among the best documents
    y[1::2] = 1.0      # ... with different gains: the reference's gain / id tie-break decides their order
    X[0::2, 0] += 5.0
    X[1::2, 0] += 5.0
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    feats = [0, 2, 5]
    bases = rng.random((3, 6)) + 0.1
    cands = [np.asarray([0.7, 0.1, 1.5]), np.asarray([0.0, 2.0]), np.asarray([-1.0, 0.25, 0.5, 3.0])]
    for rep in range(2):  # (the second call takes the exact kernel directly: more than a quarter was redone)
        means, pq = native.evaluate_candidates(g, "ndcg@5", feats, bases, cands, per_query=True)
        for gi in range(3):
            for ci in range(len(cands[gi])):
                w = bases[gi].copy()
                w[feats[gi]] = cands[gi][ci]
                exp, _ = c.metric_from_scores("ndcg@5", c.score_linear(w))
                assert np.array_equal(pq[:, gi * 64 + ci], exp), (rep, gi, ci)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "ndcg@5"
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 3, True, 4, 3
    shard, st = _train_stats(g, req)
    try:
        o.set_mean_segment(o.DEVICE_MEAN_SEGMENT)  # 700 queries: the mean's summation shape matters
        exp_s, exp_w, exp_e, err = c.ca_learn("ndcg@5", p.to_dict(), threads=2)
    finally:
        o.set_mean_segment(0)
    assert err == 0
    for r in shard["restarts"]:
        assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist()
    assert st["path"] == "fused_linesearch"  # (the evaluate_candidates calls above are the ones with > 512 redone pairs)


def test_all_scores_zero_with_a_zero_error_bound():
    """A model whose only weights sit on an all-zero column scores every document 0: the error bound computed
    from the column maxima is 0 as well, and the approximate keys then differ only by the gain class written
    into their lowest bits.  Those differences must not count as a proven order (the reference's tie-break puts
    the LOWEST gain first): small queries with distinct labels are the case that would slip through."""
    rng = np.random.default_rng(11)
    nq = 300
    lens = rng.integers(1, 6, nq)
    qid = np.repeat(np.arange(1, nq + 1), lens)
    n = len(qid)
    X = rng.random((n, 5)).astype(np.float32)
    X[:, 2] = 0.0
    y = np.concatenate([rng.permutation(5)[:k] for k in lens]).astype(np.float64)  # distinct labels inside a query
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    feats = [0, 2, 2]
    bases = np.zeros((3, 5))
    bases[:, 2] = 1.0                      # all of the weight on the zero column
    cands = [np.asarray([0.0]), np.asarray([1.0, -3.0, 0.0]), np.asarray([1e6, -1e6])]
    for measure in ("ndcg@5", "ndcg@3", "ndcg@10"):
        means, pq = native.evaluate_candidates(g, measure, feats, bases, cands, per_query=True)
        for gi in range(3):
            for ci in range(len(cands[gi])):
                w = bases[gi].copy()
                w[feats[gi]] = cands[gi][ci]
                exp, _ = c.metric_from_scores(measure, c.score_linear(w))
                assert np.array_equal(pq[:, gi * 64 + ci], exp), (measure, gi, ci)


def test_verify_kernel_near_ties_below_the_error_bound():
    """Scores that differ by less than the proven error bound (here ~1e-13 relative) cannot be ordered
    from the approximate sums: the pair is recomputed exactly and the result is the oracle's."""
    n, d = 400, 6
    rng = np.random.default_rng(77)
    X = rng.uniform(1.0, 2.0, (n, d)).astype(np.float32)
    X[1::2] = X[0::2]                                   # pairs of identical documents ...
    X[1::2, 0] = np.nextafter(X[0::2, 0], np.float32(4))  # ... one float32 ulp apart in feature 0
    y = rng.choice([0.0, 1.0, 2.0, 3.0], size=n)
    qid = np.repeat(np.arange(1, 5, dtype=np.int64), n // 4)
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    bases = np.full((1, d), 1.0 / d)
    bases[0, 0] = 1e-9                                   # the ulp in feature 0 moves a score by ~1e-16
    cands = [np.asarray([1e-9, 0.0, 1e-3, -1e-3])]
    native.profile_reset()
    native.profile_enable(True)
    means, pq = native.evaluate_candidates(g, "ndcg@5", [0], bases, cands, per_query=True)
    native.profile_enable(False)
    assert native.profile_stats()["linesearch_ndcg_kernel"]["launches"] == 1, "exact fallback expected"
    for ci in range(4):
        w = bases[0].copy()
        w[0] = cands[0][ci]
        exp, _ = c.metric_from_scores("ndcg@5", c.score_linear(w))
        assert np.array_equal(pq[:, ci], exp), ci


def test_duplicate_pairs_tied_with_a_third_document_go_to_the_exact_kernel():
    """The verify kernel keeps a tied pair of ONE duplicate group (bit-identical rows: the reference's gain tie-break orders
    them, and so do the class bits of their keys) only when nothing else is tied with it.  Here a third document equals the
    pair in every feature but one, so the `dir = 0` candidate of that feature (w_f = 0, src/coordinate_ascent.rs:152-155) ties
    all three exactly while their gains differ: such clusters must be recomputed, never decided from the keys.  Trajectory
    and evaluation counts are the oracle's; duplicate groups are in use and some pairs were redone."""
    rng = np.random.default_rng(151)
    X, y, qid = synth_dataset(151, 9000, 8, 90, max_len=200)
    X = np.abs(X)
    n = len(y)
    for i in range(2, n):
        if qid[i - 2] != qid[i]:
            continue
        u = rng.random()
        if u < 0.12:
            X[i - 1] = X[i - 2]                    # a duplicate pair with its own labels ...
            X[i] = X[i - 2]
            X[i, rng.integers(0, 8)] += 1.0        # ... and a third document that differs from it in one feature only
            y[i - 2], y[i - 1], y[i] = rng.permutation([0.0, 1.0, 2.0])
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "ndcg@10"
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations, p.init_random = 47, True, 4, 5, False
    for init_random in (False, True):
        p.init_random = init_random
        exp_s, exp_w, exp_e, err = c.ca_learn("ndcg@10", p.to_dict(), threads=2)
        assert err == 0
        shard, st = _train_stats(g, req)
        for r in shard["restarts"]:
            assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist(), init_random
        assert st["useful_evals"] == int(exp_e.sum())
        if _verify_path_on(resident_needed=True):
            assert st["verify_pairs"] > 0 and st["verify_redone"] > 0, st


def test_tie_heavy_restarts_are_routed_to_the_exact_kernel_one_by_one(monkeypatch):
    """Per-group routing (DeviceDataset::linesearch_ndcg_submit): documents duplicated with DIFFERENT labels tie exactly under
    every weight vector, so -- with the duplicate groups switched off -- the verify kernel cannot decide most pairs of any
    restart.  A restart whose verified line search left more than a quarter of its pairs undecided sends its next 4 / 8 /
    16 line searches straight to the exact kernel; the trainer's other sets keep using the verify kernel.  The trajectory
    is the oracle's either way, some group line searches were routed, and not every line search went to the exact kernel
    wholesale (round 4 sent the next 16 line searches of EVERY group there)."""
    monkeypatch.setenv("FR_NO_DUP_GROUPS", "1")
    monkeypatch.setenv("FR_VERIFY_XS", "1")  # (a pinned list length: routing is the first response, not longer lists)
    rng = np.random.default_rng(131)
    X, y, qid = synth_dataset(131, 8000, 10, 80, max_len=200)
    X = np.abs(X)
    for i in np.nonzero(rng.random(len(y)) < 0.5)[0]:
        if i > 0 and qid[i - 1] == qid[i]:
            X[i] = X[i - 1]            # same features, its own label
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "ndcg@10"
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 5, True, 6, 5
    exp_s, exp_w, exp_e, err = c.ca_learn("ndcg@10", p.to_dict(), threads=3)
    assert err == 0
    shard, st = _train_stats(g, req)
    for r in shard["restarts"]:
        assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist()
    assert st["useful_evals"] == int(exp_e.sum())
    print("routing:", {k: st[k] for k in ("groups", "exact_groups", "exact_ticks", "line_searches", "verify_pairs", "verify_redone")})
    if _verify_path_on(resident_needed=True):
        assert 0 < st["exact_groups"] < st["groups"], st
        assert st["verify_pairs"] > 0 and st["verify_redone"] * 4 > st["verify_pairs"], st  # (what triggers the routing)


def test_redo_unit_is_a_slice_of_sixteen_candidates():
    """Pairs of documents that differ ONLY in feature 0 and carry different labels tie exactly for the candidate w_0 = 0
    (the `dir = 0` candidate, src/coordinate_ascent.rs:152-155 -- candidate 0 of a line search on feature 0) and for no
    other: the verify kernel lists the (query, group) pair with a mask that names the first 16-candidate slice only, and
    the exact kernel recomputes one slice per listed pair instead of four.  Values: the oracle's, bit for bit."""
    rng = np.random.default_rng(137)
    nq, per, d = 200, 16, 6   # (at most 256 queries: the mean is one sequential sum)
    X = rng.uniform(1.0, 2.0, (nq * per, d)).astype(np.float32)
    y = rng.integers(0, 4, nq * per).astype(np.float64)
    qid = np.repeat(np.arange(1, nq + 1), per)
    X[1::2, 1:] = X[0::2, 1:]          # pairs equal in every feature but 0
    y[0::2], y[1::2] = 3.0, 1.0        # ... of different gains, and the best of the query
    X[0::2, 1] += 4.0
    X[1::2, 1] += 4.0
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "ndcg@10"
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations, p.init_random = 3, True, 3, 25, False
    exp_s, exp_w, exp_e, err = c.ca_learn("ndcg@10", p.to_dict(), threads=3)
    assert err == 0
    shard, st = _train_stats(g, req)
    for r in shard["restarts"]:
        assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist()
    print("slices:", {k: st[k] for k in ("verify_pairs", "verify_redone", "verify_redo_entries", "exact_groups")})
    if _verify_path_on(resident_needed=True):
        assert st["verify_redone"] > 0, st
        assert st["verify_redo_entries"] < 2 * st["verify_redone"], st  # (four per pair without the masks: 51 candidates = 4 slices)


def _with_same_label_duplicates(seed, n, d, nq, frac=0.1):
    """Duplicated documents that carry their source's label: exact score ties for every weight vector, none of
    which the reference's tie-break has to decide between gain classes (the continuous columns keep
    coincidental ties between different documents out)."""
    rng = np.random.default_rng(seed)
    X, y, qid = synth_dataset(seed, n, d, nq, max_len=200)
    for i in np.nonzero(rng.random(len(y)) < frac)[0]:
        if i > 0 and qid[i - 1] == qid[i]:
            X[i], y[i] = X[i - 1], y[i - 1]
    return X, y, qid


def test_verify_kernel_accepts_ties_inside_one_gain_class(monkeypatch):
    """Tied documents of ONE gain class may come in either order without changing DCG@k, so the verify kernel
    keeps such pairs (kernels_verify.inc); with more keys per list (XS = 2, 3; raised by the host while many
    pairs fail) a tied pair / triple may also straddle the cut.  The trajectory is the oracle's in every mode,
    and the share of pairs sent to the exact kernel falls with XS."""
    X, y, qid = _with_same_label_duplicates(91, 9000, 10, 90)
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "ndcg@10"
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 23, True, 4, 5
    exp_s, exp_w, exp_e, err = c.ca_learn("ndcg@10", p.to_dict(), threads=2)
    assert err == 0
    fracs = {}
    for xs in ("1", "2", "3", "4", ""):
        if xs:
            monkeypatch.setenv("FR_VERIFY_XS", xs)
        else:
            monkeypatch.delenv("FR_VERIFY_XS", raising=False)
        shard, st = _train_stats(g, req)
        for r in shard["restarts"]:
            assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist(), xs
        assert st["useful_evals"] == int(exp_e.sum())
        fracs[xs] = (st["verify_redone"] / max(1, st["verify_pairs"]), st["exact_ticks"], st["line_searches"])
    print("redo fraction / exact-only line searches / line searches by FR_VERIFY_XS:", fracs)
    if _verify_path_on(resident_needed=True):
        assert fracs["1"][0] > fracs["2"][0] > fracs["3"][0] >= fracs["4"][0], fracs
        assert fracs["3"][0] < 0.2, fracs
        assert fracs[""][0] < fracs["1"][0] and fracs[""][1] <= fracs["1"][1], fracs  # adaptive: raised after the first line searches


def test_verify_kernel_orders_exact_duplicates_by_the_tie_break(monkeypatch):
    """Documents of a query with bit-identical feature rows score exactly alike under every weight vector, so the
    reference orders them by its tie-break (gain ascending).  Their duplicate-group id rides in the keys next to the
    gain class, the class ids descend with the gain, and the verify kernel then KEEPS a tied cluster of two to four
    documents of one group even when their gains differ (kernels_verify.inc).  Same trajectory as the oracle, far fewer pairs redone
    than without the groups (FR_NO_DUP_GROUPS=1, where every such pair goes to the exact kernel)."""
    rng = np.random.default_rng(97)
    X, y, qid = synth_dataset(97, 9000, 12, 90, max_len=200)
    X = np.abs(X)
    for i in np.nonzero(rng.random(len(y)) < 0.15)[0]:
        if i > 0 and qid[i - 1] == qid[i]:
            X[i] = X[i - 1]            # same features, its own label: a tie between gain classes
    c = o.Dataset(X, y, qid)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "ndcg@10"
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations, p.init_random = 43, True, 4, 5, False  # (uniform start: positive weights)
    exp_s, exp_w, exp_e, err = c.ca_learn("ndcg@10", p.to_dict(), threads=2)
    assert err == 0
    fracs = {}
    preset_off = bool(os.environ.get("FR_NO_DUP_GROUPS"))  # (the suite is also run with the groups switched off: parity only)
    for off in ("", "1"):
        if off:
            monkeypatch.setenv("FR_NO_DUP_GROUPS", off)
        g = fr.CDataset.from_numpy(X, y, qid)
        shard, st = _train_stats(g, req)
        for r in shard["restarts"]:
            assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist(), off
        assert st["useful_evals"] == int(exp_e.sum())
        fracs[off] = (st["verify_redone"] / max(1, st["verify_pairs"]), st["exact_ticks"], st["line_searches"])
    print("redo fraction / exact-only line searches / line searches with and without duplicate groups:", fracs)
    if _verify_path_on(resident_needed=True) and not preset_off:
        assert fracs[""][0] < 0.5 * fracs["1"][0] or fracs[""][1] < fracs["1"][1], fracs
    # random-sign weights (negative keys: the groups are not used there) and per-query values of single candidates
    monkeypatch.delenv("FR_NO_DUP_GROUPS", raising=False)
    g = fr.CDataset.from_numpy(X, y, qid)
    p.init_random, p.seed = True, 44
    shard, st = _train_stats(g, req)
    exp_s, exp_w, exp_e, err = c.ca_learn("ndcg@10", p.to_dict(), threads=2)
    for r in shard["restarts"]:
        assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist()
    for sign in (1.0, -1.0):
        bases = np.full((1, 12), sign / 12.0)
        cands = [np.asarray([sign / 12.0, 0.0, sign * 0.5, -sign * 0.25])]
        means, pq = native.evaluate_candidates(g, "ndcg@10", [3], bases, cands, per_query=True)
        for ci in range(4):
            w = bases[0].copy()
            w[3] = cands[0][ci]
            exp, _ = c.metric_from_scores("ndcg@10", c.score_linear(w))
            assert np.array_equal(pq[:, ci], exp), (sign, ci)


def test_mrr_verify_ignores_ties_among_relevant_documents():
    """Reciprocal rank: near-ties and exact ties between RELEVANT documents need no resolving (whichever comes
    first has the same documents before it); only a non-relevant document close to the best relevant key sends
    the pair to rr_exact_kernel (kernels_rr.inc)."""
    X, y, qid = _with_same_label_duplicates(93, 9000, 10, 90)
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "mrr"
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 29, True, 3, 5
    shard, st = _train_stats(g, req)
    exp_s, exp_w, exp_e, err = c.ca_learn("mrr", p.to_dict(), threads=2)
    assert err == 0
    for r in shard["restarts"]:
        assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist()
    assert st["useful_evals"] == int(exp_e.sum())
    if _verify_path_on(resident_needed=True):
        assert st["verify_pairs"] > 0
        assert st["verify_redone"] / st["verify_pairs"] < 0.25, st


def _signed_heavy_tail_dataset(seed, n, d, nq):
    """Adversarial for the resident-sum error bound: signed heavy-tail columns (cancellation),
    columns of very different scale, small-integer columns (ties)."""
    rng = np.random.default_rng(seed)
    _, y, qid = synth_dataset(seed, n, d, nq, max_len=150)
    X = rng.lognormal(0.0, 2.5, (n, d)) * rng.choice([-1.0, 1.0], (n, d))
    X[:, ::5] = np.floor(rng.exponential(2.0, (n, len(range(0, d, 5)))))
    X[:, 3::7] *= 1e-4
    X[:, 1] += 0.5 * y * np.abs(X[:, 1]).mean()
    return X.astype(np.float32), y, qid


@pytest.mark.parametrize("measure", ["ndcg@10", "mrr", "ndcg"])
def test_resident_sums_never_refreshed_on_adversarial_columns(measure, monkeypatch):
    """FR_RESIDENT_REFRESH=100000: the resident sums are never re-derived exactly, so every accepted candidate of a
    whole run adds to their drift and the host's error recurrence alone keeps the verification honest.  With
    FR_VERIFY_AUDIT=1 every NDCG@k value the verify path publishes is also recomputed by the exact kernel and
    compared bit for bit (audit_mismatches == 0)."""
    monkeypatch.setenv("FR_RESIDENT_REFRESH", "100000")
    monkeypatch.setenv("FR_VERIFY_AUDIT", "1")
    X, y, qid = _signed_heavy_tail_dataset(201, 8000, 24, 80)
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = measure
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 31, True, 4, 8
    shard, st = _train_stats(g, req)   # to convergence
    exp_s, exp_w, exp_e, err = c.ca_learn(measure, p.to_dict(), threads=2)
    assert err == 0
    for r in shard["restarts"]:
        assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist()
    assert st["useful_evals"] == int(exp_e.sum())
    assert st["ticks"] > 3 * 24, "long accept chains wanted"
    if measure == "ndcg@10" and _verify_path_on(resident_needed=True):
        assert st["audit_values"] > 0 and st["audit_mismatches"] == 0, st


@pytest.mark.parametrize("env", [{}, {"FR_RESIDENT_REFRESH": "2"}, {"FR_LS_RESIDENT": "0"}, {"FR_LS_EXACT": "1"}])
def test_trainer_variants_share_one_trajectory(small, env, monkeypatch):
    """Resident base sums (updated incrementally on acceptance, refreshed exactly every N updates), sums
    from the tiles (two groups per wave), exact kernel only: all of them train to the oracle's result."""
    X, y, qid, g, c = small
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "ndcg@10"
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 13, True, 3, 6
    shard = native.train_model_shard(g, req, 0, 3)   # to convergence
    exp_s, exp_w, exp_e, err = c.ca_learn("ndcg@10", p.to_dict(), threads=3)
    assert err == 0
    for r in shard["restarts"]:
        assert r["score"] == exp_s[r["restart_id"]], env
        assert r["weights"] == exp_w[r["restart_id"]].tolist(), env
    assert shard["stats"]["useful_evals"] == int(exp_e.sum())
    assert shard["stats"]["ticks"] > 48, "several passes over the 24 features"


@pytest.mark.parametrize("measure", ["ndcg@10", "mrr"])
def test_dataset_without_the_column_major_copy_trains_from_the_tiles(measure, monkeypatch):
    """The resident line searches read their feature from the column-major copy of the tiles (as large as the matrix again).
    A dataset made without it (FR_XCOL=0; the same happens when HBM has no room) gets no resident sums: the trainer forms
    them from the tiles and lands on the oracle's result all the same."""
    X, y, qid = synth_dataset(19, 5000, 24, 50, max_len=600)
    monkeypatch.setenv("FR_XCOL", "0")
    g = fr.CDataset.from_numpy(X, y, qid)
    bare = native.device_info(g)["hbm_bytes_owned"]  # (the device dataset is made on first use: while the switch is set)
    monkeypatch.delenv("FR_XCOL")
    full = fr.CDataset.from_numpy(X, y, qid)
    assert bare < native.device_info(full)["hbm_bytes_owned"] - X.size * 4 // 2
    c = o.Dataset(X, y, qid)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = measure
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 23, True, 3, 4
    exp_s, exp_w, exp_e, err = c.ca_learn(measure, p.to_dict(), threads=3)
    assert err == 0
    for ds in (g, full):
        shard = native.train_model_shard(ds, req, 0, 3)
        for r in shard["restarts"]:
            assert r["score"] == exp_s[r["restart_id"]]
            assert r["weights"] == exp_w[r["restart_id"]].tolist()
        assert shard["stats"]["useful_evals"] == int(exp_e.sum())


@pytest.mark.parametrize("parts,measure", [("0", "ndcg@10"), ("2", "ndcg@10"), ("3", "ndcg@10"), ("4", "ndcg@10"),
                                           ("0", "mrr"), ("3", "mrr"), ("4", "mrr")])
def test_pipelined_stepping_keeps_every_restart_on_the_oracle_trajectory(small, parts, measure, monkeypatch):
    """fr_ca_step keeps one line search of each of a few sets of restarts in flight (FR_LS_PIPELINE sets, default
    3; 0 = plain lock step; NDCG@k and reciprocal rank).  Whatever the split and however the ticks are chunked into calls, every restart
    follows the oracle's trajectory and the counters agree with lock step."""
    X, y, qid, g, c = small
    monkeypatch.setenv("FR_LS_PIPELINE", parts)
    monkeypatch.setenv("FR_RESIDENT_REFRESH", "3")  # exact refreshes on the main stream between pipelined ticks
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = measure
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 17, True, 7, 5
    run = native.CoordinateAscentRun(g, req)
    chunks = [1, 2, 5, 1, 3, 11, 64]
    ticks = 0
    k = 0
    while not run.finished:
        ticks += run.step(chunks[k % len(chunks)])
        k += 1
        assert len(run.state()["restarts"]) == 7  # a consistent state between calls: nothing left in flight
    st = run.state()
    run.close()
    exp_s, exp_w, exp_e, err = c.ca_learn(measure, p.to_dict(), threads=4)
    assert err == 0
    for r in st["restarts"]:
        assert r["score"] == exp_s[r["restart_id"]], parts
        assert r["weights"] == exp_w[r["restart_id"]].tolist(), parts
    assert st["stats"]["useful_evals"] == int(exp_e.sum())
    assert st["stats"]["ticks"] == ticks
    # every restart runs whole passes over the 24 features, so the longest-lived one fixes the tick count
    monkeypatch.setenv("FR_LS_PIPELINE", "0")
    ref = native.train_model_shard(g, req, 0, 7)
    assert ref["stats"]["ticks"] == st["stats"]["ticks"]
    assert ref["stats"]["groups"] == st["stats"]["groups"]
    assert ref["restarts"] == st["restarts"]


@pytest.mark.parametrize("nlabels", [40, 200, 300])
def test_many_gain_classes_train_on_the_oracle_trajectory(nlabels):
    """The NDCG@k verify kernel reads its DCG terms from a copy of the gain-class table in LDS whose size follows the number
    of distinct gains (standard collections have 2-5; graded or continuous relevance has many): 40 and 200 classes go
    through it, more than 256 classes send the line search to the exact kernel -- the trajectory is the oracle's each time."""
    rng = np.random.default_rng(100 + nlabels)
    X, y, qid = synth_dataset(31, 5000, 24, 50, max_len=400)
    levels = np.round(np.linspace(0.0, 4.0, nlabels), 6)
    y = rng.choice(levels, size=len(y)).astype(np.float64)
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    o.set_mean_segment(o.DEVICE_MEAN_SEGMENT)
    try:
        for measure in ("ndcg@10", "ndcg@5", "ndcg@20"):
            req = fr.TrainRequest.coordinate_ascent()
            req.measure = measure
            p = req.params
            p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 5, True, 4, 4
            got = native.train_model_shard(g, req, 0, 4)
            exp_s, exp_w, exp_e, err = c.ca_learn(measure, p.to_dict(), threads=4)
            assert err == 0
            for r in got["restarts"]:
                assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist(), (nlabels, measure)
            assert got["stats"]["useful_evals"] == int(exp_e.sum())
            if nlabels <= 256:
                assert got["stats"]["verify_pairs"] > 0  # (the bound-and-verify kernel took the line searches)
    finally:
        o.set_mean_segment(0)


def test_many_gain_classes_on_a_wide_matrix_without_resident_sums(monkeypatch):
    """sums formed from the tiles (FR_LS_RESIDENT=0) keep two groups' weights in LDS next to the gain-class table;
    with ~240 classes and ~1800 columns that no longer fits a workgroup's 64 KB.  The launch must not fail: such a line
    search goes to the exact kernel (a narrower matrix with the same classes still takes the verify kernel), and the
    values are the oracle's."""
    monkeypatch.setenv("FR_LS_RESIDENT", "0")
    rng = np.random.default_rng(77)
    nlabels = 240
    for d, expect_verify in ((1800, False), (64, True)):
        X, y, qid = synth_dataset(33, 1500, d, 20, max_len=200)
        y = rng.choice(np.round(np.linspace(0.0, 4.0, nlabels), 6), size=len(y)).astype(np.float64)
        g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
        req = fr.TrainRequest.coordinate_ascent()
        req.measure = "ndcg@10"
        p = req.params
        p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 9, True, 2, 3
        run = native.CoordinateAscentRun(g, req)
        run.step(6)
        st = run.state()
        run.close()
        for r in st["restarts"]:
            assert c.evaluate_mean("ndcg@10", np.asarray(r["weights"])) == r["score"], d
        if _verify_path_on():
            assert (st["stats"]["verify_pairs"] > 0) == expect_verify, (d, st["stats"])


def test_ragged_call_chunking_keeps_the_trajectory(small):
    """However the caller chunks its fr_ca_step calls (3, 1, 7, 64 ticks ...), every restart stays on the oracle's
    trajectory and the counters are those of one uninterrupted run."""
    X, y, qid, g, c = small
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "ndcg@10"
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 23, True, 7, 5
    ref = native.train_model_shard(g, req, 0, 7)
    run = native.CoordinateAscentRun(g, req)
    chunks = [3, 1, 7, 64]
    k = 0
    while not run.finished:
        run.step(chunks[k % len(chunks)])
        k += 1
    st = run.state()
    run.close()
    exp_s, exp_w, exp_e, err = c.ca_learn("ndcg@10", p.to_dict(), threads=4)
    assert err == 0
    for r in st["restarts"]:
        assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist()
    assert st["restarts"] == ref["restarts"]
    for key in ("useful_evals", "raw_evals", "ticks", "groups", "verify_pairs"):
        assert st["stats"][key] == ref["stats"][key], key


def test_two_interleaved_trainers_on_one_dataset(small):
    """Only one trainer can own a dataset's resident sums; the one that loses them must keep producing
    the oracle's trajectory (it forms its sums from the tiles again)."""
    X, y, qid, g, c = small
    reqs = []
    for seed in (21, 22):
        req = fr.TrainRequest.coordinate_ascent()
        req.measure = "ndcg@10"
        req.params.seed, req.params.quiet, req.params.num_restarts, req.params.num_max_iterations = seed, True, 2, 5
        reqs.append(req)
    a = native.CoordinateAscentRun(g, reqs[0])
    a.step(7)
    b = native.CoordinateAscentRun(g, reqs[1])  # takes the resident buffers over
    while not (a.finished and b.finished):
        if not a.finished:
            a.step(3)
        if not b.finished:
            b.step(5)
    for run, req in ((a, reqs[0]), (b, reqs[1])):
        exp_s, exp_w, _, err = c.ca_learn("ndcg@10", req.params.to_dict(), threads=2)
        assert err == 0
        for r in run.state()["restarts"]:
            assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist()
        run.close()


@pytest.mark.parametrize("ties", [False, True])
def test_mrr_training_by_bound_and_verify(small, ties):
    """Reciprocal rank trains through rr_verify_kernel on resident sums; ties and duplicated documents
    force rr_exact_kernel.  Both must give the oracle's trajectory."""
    X, y, qid, g, c = small
    if ties:
        X = np.round(X * 2).astype(np.float32)
        X[1::2] = X[0::2][: len(X[1::2])]
        g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = "mrr"
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 17, True, 2, 5
    shard = native.train_model_shard(g, req, 0, 2)
    st = shard["stats"]
    assert st["path"] == "fused_fullrank"
    if _verify_path_on(resident_needed=True):
        assert st["verify_pairs"] > 0
        if ties:
            assert st["verify_redone"] > 0
    exp_s, exp_w, exp_e, err = c.ca_learn("mrr", p.to_dict(), threads=2)
    assert err == 0
    for r in shard["restarts"]:
        assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist()
    assert st["useful_evals"] == int(exp_e.sum())


@pytest.mark.parametrize("ties", [False, True])
@pytest.mark.parametrize("measure", ["ndcg", "map", "ndcg@50"])
def test_fullrank_training_by_sort_and_verify(small, measure, ties):
    """The reference's default measure (NDCG without a depth), MAP and NDCG beyond depth 20 train through
    fullrank_verify_kernel on resident sums: approximate keys sorted in registers, gaps between neighbours of different
    gain class verified; ties and duplicated documents fail the verification and go to the exact kernels' work-list
    mode.  Either way the trajectory is the oracle's (src/evaluators.rs:255-272, 350-380, 422-447)."""
    X, y, qid, g, c = small
    if ties:
        X = np.round(X * 2).astype(np.float32)
        X[1::2] = X[0::2][: len(X[1::2])]
        g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = measure
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 19, True, 3, 5
    shard = native.train_model_shard(g, req, 0, 3)
    st = shard["stats"]
    assert st["path"] == "fused_fullrank"
    if _verify_path_on() and not os.environ.get("FR_FV_OFF"):
        assert st["verify_pairs"] > 0
        if ties:
            assert st["verify_redone"] > 0
    exp_s, exp_w, exp_e, err = c.ca_learn(measure, p.to_dict(), threads=2)
    assert err == 0
    for r in shard["restarts"]:
        assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist()
    assert st["useful_evals"] == int(exp_e.sum())


@pytest.mark.parametrize("measure", ["ndcg", "map", "ndcg@25", "ndcg@150"])
def test_fullrank_verify_decides_mixed_label_duplicates(measure, monkeypatch):
    """Round 6: bit-identical rows with DIFFERENT labels inside a query (their exact scores tie under every weight vector, the
    reference orders them by gain ascending, src/evaluators.rs:34-49).  The sort-and-verify kernel's DUP instantiations carry
    the duplicate-group id in the keys and accept a cluster whose pairs all belong to one group; without the rule
    (FR_NO_DUP_GROUPS=1 when the dataset is made) every such pair goes to the exact kernels.  Same trajectory either way, the
    oracle's; far fewer pairs redone with the rule.  Queries of 30-400 documents: single- and multi-lane size classes; the
    depths 25 and 150 cut the lists in their first lane and in later ones (only clusters that reach into the cut matter)."""
    rng = np.random.default_rng(41)
    lens = rng.integers(30, 400, 24)
    qid = np.repeat(np.arange(1, len(lens) + 1, dtype=np.int64), lens)
    n = len(qid)
    X = rng.normal(0, 1, (n, 12)).astype(np.float32)
    y = rng.choice([0.0, 0.0, 1.0, 2.0, 3.0], n)
    start = 0
    for L in lens:  # a fifth of every query's documents are copies of others of the query, with labels of their own
        k = int(L) // 5
        src, dst = start + rng.integers(0, L, k), start + rng.integers(0, L, k)
        X[dst] = X[src]
        start += int(L)
    c = o.Dataset(X, y, qid)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = measure
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 23, True, 3, 4
    exp_s, exp_w, exp_e, err = c.ca_learn(measure, p.to_dict(), threads=2)
    assert err == 0
    redone = {}
    for rule in (True, False):
        if not rule:
            monkeypatch.setenv("FR_NO_DUP_GROUPS", "1")
        g = fr.CDataset.from_numpy(X, y, qid)
        shard = native.train_model_shard(g, req, 0, 3)
        st = shard["stats"]
        assert st["path"] == "fused_fullrank"
        for r in shard["restarts"]:
            assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist(), rule
        assert st["useful_evals"] == int(exp_e.sum())
        redone[rule] = (st["verify_redone"], st["verify_pairs"])
    if _verify_path_on() and not os.environ.get("FR_FV_OFF"):
        assert redone[True][1] > 0 and redone[True][0] * 4 < redone[False][0], redone


@pytest.mark.parametrize("measure", ["ndcg", "map", "ndcg@100"])
def test_fullrank_verify_every_size_class(measure):
    """Queries of 1 .. 2048 documents cover every instantiation of the sort kernel (16/32/64 keys in one lane; 2, 4,
    8, 16, 32 lanes per candidate with cross-lane merge rounds), with negative gains, a query without relevant
    documents, and near-duplicate columns: per-query values against the oracle, stateless (sums from the tiles)."""
    lens = np.array([1, 2, 7, 15, 16, 17, 31, 33, 63, 64, 65, 100, 128, 129, 200, 256, 300, 511, 513, 1000, 1025, 2048, 40])
    rng = np.random.default_rng(71)
    qid = np.repeat(np.arange(1, len(lens) + 1, dtype=np.int64), lens)
    n, d = len(qid), 12
    X = rng.normal(0, 1, (n, d)).astype(np.float32)
    X[:, 3] = np.floor(rng.exponential(2.0, n)).astype(np.float32)  # integer column: exact ties when it carries the weight
    X[:, 7] = np.where(rng.random(n) < 0.7, 0.0, rng.random(n)).astype(np.float32)
    y = rng.choice([0.0, 0.0, 1.0, 2.0, 3.0, 4.0], n)
    y[qid == 5] = 0.0
    y[(qid == 12) & (y == 0)] = -1.0
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    feats, bases, cands = _ca_groups(rng, d, 4, iters=25)
    feats[0], feats[1] = 3, 7
    bases[2][:] = 0.0
    bases[2][3] = 1.0  # group 2: all weight on the integer column -> ties inside and across classes
    feats[2] = 5
    for gi in range(4):
        cands[gi] = o.ca_candidates(bases[gi][feats[gi]], 0.05, 2.0, 25)
    means, pq = native.evaluate_candidates(g, measure, feats, bases, cands, per_query=True)
    norms = c.default_norms(measure)
    for gi in range(4):
        for ci in range(len(cands[gi])):
            w = bases[gi].copy()
            w[feats[gi]] = cands[gi][ci]
            exp, err = c.metric_from_scores(measure, c.score_linear(w), norms)
            assert err == 0
            assert np.array_equal(pq[:, gi * 64 + ci], exp), (measure, gi, ci, np.nonzero(pq[:, gi * 64 + ci] != exp)[0])


@pytest.mark.parametrize("measure", ["ndcg@3", "ndcg@20", "ndcg@1"])
def test_resident_training_other_depths(small, measure):
    X, y, qid, g, c = small
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = measure
    p = req.params
    p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 23, True, 2, 4
    shard = native.train_model_shard(g, req, 0, 2)
    assert shard["stats"]["path"] == "fused_linesearch"
    assert shard["stats"]["verify_pairs"] > 0 or not _verify_path_on()
    exp_s, exp_w, exp_e, err = c.ca_learn(measure, p.to_dict(), threads=2)
    assert err == 0
    for r in shard["restarts"]:
        assert r["score"] == exp_s[r["restart_id"]] and r["weights"] == exp_w[r["restart_id"]].tolist()
    assert shard["stats"]["useful_evals"] == int(exp_e.sum())


def test_randomised_parity_soak():
    """tools/fuzz_parity.py: random small datasets (ties, duplicates, integer columns, negative gains, long and
    one-document queries), random parameters and measures; every restart must equal the oracle's."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_parity.py"), "--iters", "80", "--seed", "7"],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    last = out.stdout.strip().splitlines()[-1]
    assert out.returncode == 0, out.stdout[-2000:]
    assert json.loads(last)["mismatches"] == 0


def test_sampled_views_share_the_parents_matrix_and_match_the_oracle(monkeypatch):
    """dataset_query_sampling / dataset_feature_sampling views do not tile a second copy of X:
    a query sample owns only query / run tables over the parent's tiles -- its runs start wherever its first query
    starts inside a tile -- and a feature sample IS its parent's device dataset.  Every path (fused NDCG@k training on
    resident sums, sort-and-verify, MRR, per-query evaluation, tree scoring, rank order, RF training) must give the
    oracle's numbers for the subset, and the same bits as a view that tiles its own copy (FR_VIEW_COPIES=1)."""
    X, y, qid = synth_dataset(401, 9000, 24, 120, max_len=200)
    parent = fr.CDataset.from_numpy(X, y, qid)
    names = [str(int(v)) for v in dict.fromkeys(qid.tolist())]  # first-appearance order
    rng = np.random.default_rng(5)
    chosen = sorted(rng.choice(len(names), size=70, replace=False).tolist())  # gaps of every length between kept queries
    sub_names = [names[i] for i in chosen]
    rows = np.isin(np.array([str(int(q)) for q in qid]), sub_names)
    Xs, ys, qs = np.ascontiguousarray(X[rows]), np.ascontiguousarray(y[rows]), np.ascontiguousarray(qid[rows])
    c = o.Dataset(Xs, ys, qs)
    sub = parent.subsample_queries(sub_names)
    info = native.device_info(sub)
    pinfo = native.device_info(parent)
    assert info["queries"] == 70 and info["instances"] == int(rows.sum())
    fsub = sub.subsample_feature_names([str(j) for j in range(0, 24, 2)])
    if not os.environ.get("FR_VIEW_COPIES"):  # (the suite is also run with every view tiling its own copy: parity only)
        assert info["shares_parent_matrix"] and not info["is_parent_device_dataset"]
        assert info["hbm_bytes_owned"] < pinfo["hbm_bytes_owned"] / 20, (info, pinfo)
        finfo = native.device_info(fsub)
        assert finfo["shares_parent_matrix"] and finfo["hbm_bytes_owned"] < pinfo["hbm_bytes_owned"] / 20
        fpar = parent.subsample_feature_names([str(j) for j in range(0, 24, 2)])
        assert native.device_info(fpar)["is_parent_device_dataset"]

    def run_all(view):
        out = {}
        for measure in ("ndcg@10", "ndcg", "mrr", "map"):
            req = fr.TrainRequest.coordinate_ascent()
            req.measure = measure
            p = req.params
            p.seed, p.quiet, p.num_restarts, p.num_max_iterations = 41, True, 3, 4
            shard = native.train_model_shard(view, req, 0, 3)
            out[measure] = ([(r["score"], r["weights"]) for r in shard["restarts"]], shard["stats"]["path"], p.to_dict())
        w = np.linspace(-1, 1, 24)
        m = fr.CModel.from_dict({"Linear": {"weights": w.tolist()}})
        out["eval"] = {k: native.evaluate_dense(m, view, k)[1].tolist() for k in ("ndcg@5", "map", "mrr", "ndcg")}
        out["rank"] = [a.tolist() for a in native.rank_order(m, view)]
        out["scores"] = native.predict_scores_dense(m, view, len(y)).tolist()
        rf = fr.TrainRequest.random_forest()
        rf.measure = "ndcg@5"
        rf.params.quiet, rf.params.num_trees, rf.params.seed = True, 5, 3
        out["rf"] = view.train_model(rf).to_dict()
        out["rf_params"] = rf.params.to_dict()
        return out

    got = run_all(sub)
    # oracle on the materialised subset
    for measure in ("ndcg@10", "ndcg", "mrr", "map"):
        restarts, path, params = got[measure]
        exp_s, exp_w, _, err = c.ca_learn(measure, params, threads=2)
        assert err == 0 and path in ("fused_linesearch", "fused_fullrank")
        for r, (s, w) in enumerate(restarts):
            assert s == exp_s[r] and w == exp_w[r].tolist(), measure
    w = np.linspace(-1, 1, 24)
    for k, vals in got["eval"].items():
        exp, _ = c.metric_from_scores(k, c.score_linear(w))
        assert vals == exp.tolist(), k
    exp_scores = np.full(len(y), np.nan)
    exp_scores[rows] = c.score_linear(w)
    assert np.array_equal(np.asarray(got["scores"]), exp_scores, equal_nan=True)
    trees, tw, _ = c.rf_learn("ndcg@5", got["rf_params"])
    # (instance ids differ between the view and the materialised subset, but their ORDER is the same: same forest)
    assert got["rf"] == {"Ensemble": {"weights": tw.tolist(), "models": [{"DecisionTree": t} for t in trees]}}
    # and bit for bit what a view with its own copy of the matrix gives
    monkeypatch.setenv("FR_VIEW_COPIES", "1")
    own = parent.subsample_queries(sub_names)
    assert not native.device_info(own)["shares_parent_matrix"]
    ref = run_all(own)
    for k in ("ndcg@10", "ndcg", "mrr", "map", "eval", "rank", "scores", "rf"):
        if k == "scores":
            assert np.array_equal(np.asarray(got[k]), np.asarray(ref[k]), equal_nan=True)
        else:
            assert got[k] == ref[k], k
