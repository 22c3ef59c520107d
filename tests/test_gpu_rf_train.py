"""Random-forest TRAINING on the device (csrc/rf_train.hpp, kernels_rf.inc) against the oracle's restatement of
src/random_forest.rs:211-408: identical trees -- structure, thresholds and leaf values bit for bit -- and identical
tree weights, for every split method, through the reference's own entry point (train_model with RandomForest params)."""
import json
import os

import numpy as np
import pytest

import fastrank_amd as fr
from fastrank_amd import native
from oracle import pyoracle as o
from tests.conftest import GOLDEN, ranksvm_presence, synth_dataset

pytestmark = pytest.mark.gpu


def _request(measure="ndcg@5", **kw):
    req = fr.TrainRequest.random_forest()
    req.measure = measure
    p = req.params
    p.quiet = True
    for k, v in kw.items():
        setattr(p, k, v)
    if isinstance(p.split_method, str):
        p.split_method = {p.split_method: []}
    return req


def _oracle(c, req):
    trees, w, sample = c.rf_learn(req.measure, req.params.to_dict())
    return {"Ensemble": {"weights": w.tolist(), "models": [{"DecisionTree": t} for t in trees]}}, sample


@pytest.fixture(scope="module")
def trec():
    d = np.load(os.path.join(GOLDEN, "trec_news_2018.npz"))
    X, y, qid = d["train_X"], d["train_y"], d["train_qid"]
    return X, y, qid, fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)


def test_regression_tree_known_answer_through_training():
    """src/random_forest.rs:465-506: one feature, ten instances; the learned tree must predict every label."""
    X = np.array([1, 1, 2, 3, 4, 5, 6, 7, 8, 9], dtype=np.float32)[:, None]
    y = np.array([7, 7, 7, 7, 2, 2, 2, 12, 12, 12], dtype=np.float64)
    qid = np.zeros(10, dtype=np.int64)
    g = fr.CDataset.from_numpy(X, y, qid)
    req = _request("ndcg", num_trees=1, min_leaf_support=1, max_depth=10, split_candidates=32, seed=5,
                   instance_sampling_rate=1.0, feature_sampling_rate=1.0)
    model = g.train_model(req)
    assert np.array_equal(native.predict_scores_dense(model, g), y)
    exp, _ = _oracle(o.Dataset(X, y, qid), req)
    assert model.to_dict() == exp


def test_fewer_than_two_split_candidates_gives_single_leaf_trees(trec):
    """split_candidates = 1: `(1..k)` is empty (src/random_forest.rs:236), no node splits and every tree is the mean of
    its sample.  (Round 2 launched an empty grid here, or read the previous training's node tables.)"""
    X, y, qid, g, c = trec
    warm = _request(num_trees=3, seed=1, split_candidates=8, max_depth=6, min_leaf_support=2)
    assert g.train_model(warm).to_dict() == _oracle(c, warm)[0]  # leaves node tables of another shape behind
    for k in (1, 0):
        req = _request(num_trees=4, seed=9, split_candidates=k, weight_trees=True)
        got = g.train_model(req).to_dict()
        exp, _ = _oracle(c, req)
        assert got == exp
        assert all("LeafNode" in m["DecisionTree"] for m in got["Ensemble"]["models"])
    assert g.train_model(warm).to_dict() == _oracle(c, warm)[0]


@pytest.mark.parametrize("rates", [(2.5, 1.5), (1e300, 7.0), (-1.0, 0.5), (0.0, 0.0)])
def test_sampling_rates_outside_the_unit_interval_saturate_like_rust(trec, rates):
    """`(len as f64 * rate) as usize` saturates in Rust (src/sampling.rs:49-50): a rate above one takes everything,
    a negative or NaN rate gives 0 and then max(1, .) = one item."""
    X, y, qid, g, c = trec
    req = _request(num_trees=2, seed=3, instance_sampling_rate=rates[0], feature_sampling_rate=rates[1], min_leaf_support=2)
    d = req.to_dict()
    got = g.train_model(req).to_dict()
    exp, sample = _oracle(c, req)
    assert got == exp
    nq, nf = len(set(qid.tolist())), X.shape[1]
    want_f = nf if rates[1] >= 1.0 else max(1, int(nf * rates[1])) if rates[1] > 0 else 1
    assert int(sample[0][0]) == want_f


@pytest.mark.parametrize("method", ["SquaredError", "BinaryGiniImpurity", "InformationGain", "TrueVarianceReduction"])
def test_reference_test_configuration_matches_oracle(trec, method):
    """The configuration of the reference's determinism test (src/random_forest.rs:427-463; 10 trees, seed 42,
    32 split candidates, depth 10) on its example data, for every split method, with weighted trees."""
    X, y, qid, g, c = trec
    req = _request("ndcg@5", num_trees=10, seed=42, min_leaf_support=2 if method == "TrueVarianceReduction" else 1,
                   max_depth=10, split_candidates=32, split_method=method, weight_trees=True)
    model = g.train_model(req)
    exp, sample = _oracle(c, req)
    got = model.to_dict()
    assert got["Ensemble"]["weights"] == exp["Ensemble"]["weights"]
    assert got == exp
    assert (sample[:, 0] == 1).all() and (sample[:, 1] > 100).all()  # floor(6 x 0.25) = 1 feature, ~half the queries
    st = native.last_train_stats()
    assert st["path"] == "random_forest" and st["restarts"] == 10
    # the forest scores like any loaded forest
    vals = native.evaluate_dense(model, g, "ndcg@5")[1]
    expv, _ = c.metric_from_scores("ndcg@5", c.score_ensemble([m["DecisionTree"] for m in exp["Ensemble"]["models"]], exp["Ensemble"]["weights"]))
    assert np.array_equal(vals, expv)


@pytest.mark.parametrize("batch_bytes", ["", "200000"])
def test_defaults_on_mslr_shape_with_ties_and_several_batches(batch_bytes, monkeypatch):
    """Default parameters (100 trees, 3 split candidates, depth 8, min leaf 10) on an MSLR-shaped matrix with
    integer columns (equal feature values inside nodes: the tie order matters) -- in one batch and cut into many."""
    if batch_bytes:
        monkeypatch.setenv("FR_RF_BATCH_BYTES", batch_bytes)
    X, y, qid = synth_dataset(301, 6000, 24, 60, max_len=300)
    X[:, 1::3] = np.floor(X[:, 1::3] * 3)
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    req = _request("ndcg@10", seed=7, num_trees=24)
    model = g.train_model(req)
    exp, _ = _oracle(c, req)
    assert model.to_dict() == exp
    st = native.last_train_stats()
    assert st["groups"] >= (2 if batch_bytes else 1), "batches"
    depths = [json.dumps(m).count("FeatureSplit") for m in model.to_dict()["Ensemble"]["models"]]
    assert max(depths) > 10, "real trees expected"


@pytest.mark.parametrize("method", ["SquaredError", "TrueVarianceReduction", "BinaryGiniImpurity", "InformationGain"])
def test_fractional_and_huge_labels_keep_the_sequential_sums(method):
    """Integer labels of moderate size let the device form a split's gain sums in integer arithmetic (any order is the
    reference's f64 sum then, kernels_rf.inc RFArgs::labels_int); fractional labels -- and integers too large for that
    argument -- must take the sequential chain, whose association is the reference's (random_forest.rs:32-51)."""
    X, y, qid = synth_dataset(77, 5000, 16, 50, max_len=300)
    rng = np.random.default_rng(5)
    # (labels beyond 2^21 with a measure that only asks whether they are positive: 2^label overflows NDCG's gains)
    for labels, measure in ((y + np.round(rng.random(len(y)), 3), "ndcg@10"), (y * 3.0e6, "map")):
        g, c = fr.CDataset.from_numpy(X, labels, qid), o.Dataset(X, labels, qid)
        req = _request(measure, seed=3, num_trees=6)
        req.params.split_method = {method: []}
        req.params.min_leaf_support = 5
        exp, _ = _oracle(c, req)
        assert g.train_model(req).to_dict() == exp


@pytest.mark.parametrize("env", [{"FR_RF_PARTITION": "0"}, {"FR_RF_INT_SUMS": "0"}, {"FR_RF_CHILDSUM": "1"},
                                 {"FR_RF_PARTITION": "0", "FR_RF_INT_SUMS": "0", "FR_RF_CHILDSUM": "1", "FR_RF_RESORT": "1"}])
def test_round2_formulations_grow_the_same_forest(env, monkeypatch):
    """Round 3 replaced three pieces of the level step -- rekey + radix sort by a stable partition, the gain sums' chain by
    integer arithmetic (integer labels), the children's sums by the chosen candidate's -- and keeps the round-2
    formulation of each behind a switch: every combination must grow the forest the oracle grows."""
    X, y, qid = synth_dataset(303, 7000, 20, 70, max_len=300)
    X[:, ::4] = np.floor(X[:, ::4] * 2)
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    req = _request("ndcg@10", seed=11, num_trees=8, min_leaf_support=4)
    exp, _ = _oracle(c, req)
    assert g.train_model(req).to_dict() == exp
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    assert g.train_model(req).to_dict() == exp


def test_sampled_view_trains_on_its_own_queries_and_features(trec):
    """A query / feature subsample (the Python API's train-test split) trains on exactly its instances and features."""
    X, y, qid, g, c = trec
    names = sorted(g.queries())
    sub_q = names[::2]
    sub = g.subsample_queries(sub_q)
    rows = np.isin(np.array([str(int(q)) for q in qid]), sub_q)
    c_sub = o.Dataset(np.ascontiguousarray(X[rows]), np.ascontiguousarray(y[rows]), np.ascontiguousarray(qid[rows]))
    req = _request("ndcg@5", num_trees=6, seed=11, min_leaf_support=3, split_candidates=8, instance_sampling_rate=0.7,
                   feature_sampling_rate=0.5)
    model = sub.train_model(req)
    exp, _ = _oracle(c_sub, req)
    assert model.to_dict() == exp


def test_degenerate_nodes_become_leaves():
    """All labels equal / one instance per query / max_depth 1: no split is looked for and the forest is made of leaves
    carrying the sample's mean gain (src/random_forest.rs:344-351, 366-377)."""
    X, y, qid = synth_dataset(303, 400, 6, 40, max_len=30)
    g, c = fr.CDataset.from_numpy(X, y, qid), o.Dataset(X, y, qid)
    for kw in (dict(max_depth=1), dict(min_leaf_support=100000)):
        req = _request("ndcg", num_trees=3, seed=3, **kw)
        model = g.train_model(req)
        exp, _ = _oracle(c, req)
        assert model.to_dict() == exp
        assert all("LeafNode" in m["DecisionTree"] for m in model.to_dict()["Ensemble"]["models"])
    y1 = np.full_like(y, 2.0)
    g1, c1 = fr.CDataset.from_numpy(X, y1, qid), o.Dataset(X, y1, qid)
    req = _request("ndcg", num_trees=2, seed=3, min_leaf_support=1)
    assert g1.train_model(req).to_dict() == _oracle(c1, req)[0]


@pytest.mark.parametrize("method", ["SquaredError", "TrueVarianceReduction"])
def test_file_loaded_dataset_feature_stats_skip_absent_values(method):
    """examples/trec_news_2018.train as the reference loads it: 13 rows do not HOLD feature 5 (Dense32 of length 5).  An
    absent value sorts as 0.0 (src/random_forest.rs:228) but FeatureStats -- the min / max the k-1 thresholds are spread
    over -- skips it (src/normalizers.rs:24-29).  Round 2 densified the file and took min / max over the zeros too."""
    path = os.path.join(GOLDEN, "data", "trec_news_2018.train")
    rd = fr.CDataset.open_ranksvm(path)
    d = np.load(os.path.join(GOLDEN, "trec_news_2018.npz"))
    X, y, qid = d["train_X"], d["train_y"], d["train_qid"]
    present = ranksvm_presence(path, X.shape[1])
    assert present.shape == X.shape and int((~present).sum()) > 0 and not X[~present].any()
    c = o.Dataset(X, y, qid)
    c.set_presence(present)
    req = _request(num_trees=10, seed=42, split_candidates=32, max_depth=10, min_leaf_support=2, split_method=method,
                   instance_sampling_rate=1.0, feature_sampling_rate=1.0)
    got = rd.train_model(req).to_dict()
    exp, _ = _oracle(c, req)
    assert got == exp
    # the same file as a numpy DenseDataset holds every value (src/dense_dataset.rs:143-147): different thresholds
    c.set_presence(None)
    dense_exp, _ = _oracle(c, req)
    assert fr.CDataset.from_numpy(X, y, qid).train_model(req).to_dict() == dense_exp
    assert dense_exp != exp
