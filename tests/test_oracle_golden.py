"""Pins oracle/ (the CPU restatement) to the reference's own known answers."""
import numpy as np
import pytest

from oracle import pyoracle as o


def _names(known):
    names = {int(k): v for k, v in known["feature_names"].items()}
    names[0] = "0"
    return names


def test_rank_ties_comparator(known):
    # src/evaluators.rs:61-79
    r = known["rank_ties"]
    assert o.rank_order(r["scores"], r["gains"], r["ids"]).tolist() == r["expected_order"]


def test_negative_zero_scores_tie():
    # NotNan<f64>::cmp treats -0.0 == +0.0 -> falls through to gain/id
    assert o.rank_order([0.0, -0.0, 0.0], [1.0, 0.0, 0.0], [0, 1, 2]).tolist() == [1, 2, 0]


def test_compute_ndcg(known):
    # src/evaluators.rs:285-295
    c = known["compute_ndcg"]
    v = o.compute_dcg(c["gains"], None, False) / o.compute_dcg(c["gains"], None, True)
    assert abs(v - c["expected"]) <= c["tolerance"]


def test_dcg_depth_pad_and_truncate():
    g = [3.0, 0.0, 2.0]
    full = o.compute_dcg(g, None, False)
    assert o.compute_dcg(g, 10, False) == full  # zero padding adds (2^0-1)/log = 0
    assert o.compute_dcg(g, 1, False) == (2.0 ** 3 - 1) / np.log2(2.0)
    assert o.compute_dcg(g, 2, True) == 7.0 / np.log2(2.0) + 3.0 / np.log2(3.0)


def test_single_feature_ndcg5(known, trec):
    # tests/test_with_example_data.py:16-23,139-167: weight 1.0 on one feature, mean NDCG@5
    ds = o.Dataset(trec["train_X"], trec["train_y"], trec["train_qid"])
    assert ds.n == known["expected_n"] and ds.d == known["expected_d"]
    assert sorted(str(int(q)) for q in ds.query_ids()) == known["expected_queries"]
    names = _names(known)
    for f in range(ds.d):
        w = np.zeros(ds.d)
        w[f] = 1.0
        got = ds.evaluate_mean("ndcg@5", w)
        assert got == pytest.approx(known["single_feature_ndcg5"][names[f]], abs=1e-12)


def test_ca_single_feature_reproduces_known_answers(known, trec):
    # the reference test's actual procedure: CA, 1 restart, 1 iteration, step_base 1, no
    # normalise, uniform init, restricted to one feature -> mean NDCG@5 (7 places)
    ds = o.Dataset(trec["train_X"], trec["train_y"], trec["train_qid"])
    names = _names(known)
    params = dict(num_restarts=1, num_max_iterations=1, step_base=1.0, step_scale=2.0,
                  tolerance=0.001, seed=42, normalize=False, init_random=False)
    for f in range(ds.d):
        scores, weights, evals, err = ds.ca_learn("ndcg", params, fids=[f])
        assert err == 0
        w = weights[0]
        assert all(w[j] == 0.0 for j in range(len(w)) if j != f)
        got = np.mean(list(ds.evaluate_by_query("ndcg@5", w).values()))
        assert got == pytest.approx(known["single_feature_ndcg5"][names[f]], abs=5e-8)


def test_with_and_without_qrel_equal(trec, qrel_dict):
    # tests/test_with_example_data.py:243-252
    ds = o.Dataset(trec["train_X"], trec["train_y"], trec["train_qid"])
    w = np.array([0.0, 0.3, -0.2, 0.5, 0.1, 0.9])
    a = ds.evaluate_mean("ndcg@5", w)
    b = ds.evaluate_mean("ndcg@5", w, norms=ds.qrel_norms("ndcg@5", qrel_dict))
    assert abs(a - b) < 1e-7


def test_regression_tree_rule(known):
    # src/random_forest.rs:465-506: `fval <= split -> lhs`; a tree splitting between the label
    # plateaus predicts every label exactly
    t = known["regression_tree"]
    X = np.asarray(t["xs"], dtype=np.float32).reshape(-1, 1)
    y = np.asarray(t["ys"], dtype=np.float64)
    ds = o.Dataset(X, y, np.zeros(len(y), dtype=np.int64))
    tree = {"FeatureSplit": {"fid": 0, "split": 3.0, "lhs": {"LeafNode": 7.0},
                             "rhs": {"FeatureSplit": {"fid": 0, "split": 6.0, "lhs": {"LeafNode": 2.0},
                                                      "rhs": {"LeafNode": 12.0}}}}}
    pred = ds.score_ensemble([tree], [1.0])
    assert np.max(np.abs(pred - y)) <= t["tolerance"]


def test_mean_counts_zero_relevant_queries():
    X = np.ones((4, 1), dtype=np.float32)
    y = np.array([1.0, 0.0, 0.0, 0.0])
    qid = np.array([1, 1, 2, 2], dtype=np.int64)
    ds = o.Dataset(X, y, qid)
    # q1: all tie -> gain asc -> [0,1]: dcg = 1/log2(3); ideal = 1 ; q2: no relevant -> 0
    assert ds.evaluate_mean("ndcg", [1.0]) == pytest.approx((1 / np.log2(3.0)) / 2)
    assert ds.evaluate_mean("rr", [1.0]) == pytest.approx(0.25)
    assert ds.evaluate_mean("map", [1.0]) == pytest.approx(0.25)


def test_ap_rr_small():
    X = np.array([[3.0], [2.0], [1.0], [0.5]], dtype=np.float32)
    y = np.array([0.0, 1.0, 0.0, 2.0])
    ds = o.Dataset(X, y, np.array([7, 7, 7, 7], dtype=np.int64))
    assert ds.evaluate_mean("rr", [1.0]) == 0.5
    assert ds.evaluate_mean("ap", [1.0]) == pytest.approx((1 / 2 + 2 / 4) / 2)
    assert ds.evaluate_mean("MAP", [-1.0]) == pytest.approx((1 / 1 + 2 / 3) / 2)


def test_qid_range_check():
    X = np.ones((2, 1), dtype=np.float32)
    with pytest.raises(ValueError):
        o.Dataset(X, np.zeros(2), np.array([1, -1], dtype=np.int64))
    with pytest.raises(ValueError):
        o.Dataset(X, np.zeros(2), np.array([1, 2 ** 32], dtype=np.int64))


def test_ca_candidates_recurrence():
    # coordinate_ascent.rs:145-171; dir 0 makes w exactly 0.0
    c = o.ca_candidates(0.25, 0.05, 2.0, 3)
    assert c[0] == 0.0
    step = -0.05
    tot = step
    exp = []
    for _ in range(3):
        exp.append(0.25 + tot)
        step *= 2.0
        tot += step
    assert c[1:4].tolist() == exp
    # |step| > 0.5|orig| -> proportional step
    c = o.ca_candidates(0.01, 0.05, 2.0, 1)
    assert c[1] == 0.01 + (-(0.05 * 0.01)) and c[2] == 0.01 + 0.05 * 0.01


def test_select_best_is_last_max():
    assert o.select_best([0.1, 0.5, 0.5, 0.2]) == 2


def test_rand64_is_deterministic_and_shuffle_is_permutation():
    # parity with oorandom is UNPINNED (see oracle header); only self-consistency here
    a = o.rand64_stream(42, 8)
    b = o.rand64_stream(42, 8)
    assert (a == b).all() and len(set(a.tolist())) == 8
    assert sorted(o.shuffle_with_seed(7, 136).tolist()) == list(range(136))


def test_ca_improves_and_is_thread_invariant(trec):
    ds = o.Dataset(trec["train_X"], trec["train_y"], trec["train_qid"])
    params = dict(num_restarts=4, num_max_iterations=5, step_base=0.05, step_scale=2.0,
                  tolerance=0.001, seed=42, normalize=True, init_random=True)
    s1, w1, e1, err1 = ds.ca_learn("ndcg@5", params, threads=1)
    s4, w4, e4, err4 = ds.ca_learn("ndcg@5", params, threads=4)
    assert err1 == 0 and err4 == 0
    assert (s1 == s4).all() and (w1 == w4).all() and (e1 == e4).all()
    best = o.select_best(s1)
    assert ds.evaluate_mean("ndcg@5", w1[best]) == s1[best]
    assert s1[best] > 0.3
    # restart shard returns the same restarts
    s_sh, w_sh, _, _ = ds.ca_learn("ndcg@5", params, threads=2, restart_range=(2, 4))
    assert (s_sh[2:] == s1[2:]).all() and (w_sh[2:] == w1[2:]).all()


def test_two_fma_quotient_equals_ieee_division_for_every_rank_pair():
    """fullrank_verify_kernel forms the AP term recall / rank as q0 = recall * y, q1 = fma(fma(-q0, rank, recall), y, q0)
    with y = 1 / rank: identical to the IEEE quotient the reference computes (src/evaluators.rs:443) for every pair
    the kernel can meet (ranks up to 2048; checked up to 4096, 8.4 million pairs)."""
    assert o.check_div_identity(4096) == 0
