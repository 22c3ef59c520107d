"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/fastrank.h
declares, speaks the reference's JSON protocol, and fails LOUDLY (no CPU fallback) when a
compute call is made without a GPU.  No device arithmetic happens in this file."""
import ctypes
import json
import os
import re

import numpy as np
import pytest

import fastrank_amd as fr
from fastrank_amd import clib, native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "fastrank.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"typedef[^;]*\(\s*\*[^;]*;", "", text)  # function-pointer typedefs are not exports
    names = re.findall(r"\b([a-z_][a-z0-9_]*)\s*\(", text)
    return sorted(set(n for n in names if n not in ("defined",)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(clib._build.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 35
    for name in syms:
        assert hasattr(lib, name), "missing export: " + name
    for name in clib.exported_symbols():  # the 20 reference symbols (src/lib.rs:78-326)
        assert name in syms


def test_pricing_switches_are_not_in_the_shipped_library():
    """Timing ablations that return wrong numbers on purpose (a kernel phase skipped: FR_TREE_NOWALK, FR_TREE_NOSTAGE,
    FR_LS_DEBUG) and the tuning sweeps exist only in a -DFR_PRICING build (csrc/device.hpp: pricing_env): the default
    library does not even hold their names, so its environment cannot talk it into wrong scores."""
    if os.environ.get("FR_BUILD_FLAGS", "").find("FR_PRICING") >= 0:
        pytest.skip("pricing build")
    blob = open(clib._build.LIB_PATH, "rb").read()
    for name in (b"FR_TREE_NOWALK", b"FR_TREE_NOSTAGE", b"FR_LS_DEBUG", b"FR_ORDER_KAPPA", b"FR_RANK_PERIOD", b"FR_TREE_NMAX"):
        assert name not in blob, name
    assert b"FR_LS_EXACT" in blob  # (a supported switch: the exact kernels alone, same results)
    n = 0
    for fn in os.listdir(os.path.join(ROOT, "fastrank_amd", "csrc")):
        n += sum(1 for line in open(os.path.join(ROOT, "fastrank_amd", "csrc", fn), errors="replace") if "getenv" in line)
    assert n <= 20, n


def test_walk_tiles_cut_runs_at_query_boundaries():
    """Round 6's layout for the resident NDCG@k verify kernel (csrc/device.hpp: build_walk_tiles; kernels_order.inc says why): a
    run's positions are cut into tiles of at most 128 positions; a query of up to 128 documents lies inside ONE tile; a longer
    one is cut every 128 documents from its start; tiles never span runs; every document's segment is the intersection of its
    query and its tile, relative to the tile's start.  Host arithmetic only."""
    rng = np.random.default_rng(17)
    for trial in range(30):
        nq = int(rng.integers(1, 60))
        qlen = np.clip(rng.lognormal(np.log(60), 1.0, nq), 1, 700).astype(np.uint32)
        if trial % 5 == 0:
            qlen[rng.integers(0, nq)] = 128
            qlen[rng.integers(0, nq)] = 129
            qlen[rng.integers(0, nq)] = 256
            qlen[rng.integers(0, nq)] = 1
        # runs like DeviceDataset::create's: consecutive queries up to ~768 documents, every run starting on a multiple of 64
        qstart, run_pos, run_q0, run_q1, pos, cur, q0 = [], [], [], [], 0, 0, 0
        for q in range(nq):
            if cur > 0 and cur + qlen[q] > 768:
                run_pos.append(pos - cur), run_q0.append(q0), run_q1.append(q)
                pos, cur, q0 = (pos + 63) // 64 * 64, 0, q
            qstart.append(pos)
            pos += int(qlen[q])
            cur += int(qlen[q])
        run_pos.append(pos - cur), run_q0.append(q0), run_q1.append(nq)
        npos = (pos + 63) // 64 * 64
        w = native.walk_tiles(run_pos, run_q0, run_q1, qstart, qlen, npos)
        T, wt, seg, wofs = w["walk_tile"], w["wt_start"], w["seg"], w["wofs"]
        assert T == 128 and wt[-1] == npos and wt[:-1] == sorted(set(wt[:-1])) and len(seg) == npos == len(wofs)
        run_end = [qstart[b - 1] + int(qlen[b - 1]) for b in run_q1]
        tile_len = []
        for i in range(len(wt) - 1):
            r = max(k for k in range(len(run_pos)) if run_pos[k] <= wt[i])          # the run the tile starts in
            tile_len.append(min(wt[i + 1], run_end[r]) - wt[i])
            assert 0 < tile_len[-1] <= T and wt[i] + tile_len[-1] <= run_end[r]     # inside its run, at most 128 positions
        assert [wt[i] for i in w["run_wt0"]] == run_pos                             # a run starts a tile
        covered = np.zeros(npos, dtype=bool)
        for i, n in enumerate(tile_len):
            assert not covered[wt[i]:wt[i] + n].any()
            covered[wt[i]:wt[i] + n] = True
        docs = np.zeros(npos, dtype=bool)
        for q in range(nq):
            b, e = qstart[q], qstart[q] + int(qlen[q])
            docs[b:e] = True
            tiles = [i for i in range(len(tile_len)) if wt[i] < e and wt[i] + tile_len[i] > b]
            if qlen[q] <= T:
                assert len(tiles) == 1, (trial, q, int(qlen[q]))                    # never cut
            else:
                assert [wt[i] for i in tiles] == [b + k * T for k in range(len(tiles))] and len(tiles) == -(-int(qlen[q]) // T)
            for i in tiles:
                lo, hi = max(b, wt[i]) - wt[i], min(e, wt[i] + tile_len[i]) - wt[i]
                for pp in range(wt[i] + lo, wt[i] + hi):
                    assert seg[pp] == (lo | (hi << 8)) and wofs[pp] == pp - wt[i]
        assert np.array_equal(covered, docs)                                        # the tiles hold exactly the documents
        assert all(seg[pp] == 0 for pp in np.nonzero(~docs)[0])
    with pytest.raises(Exception, match="beyond np"):
        native.walk_tiles([0], [0], [1], [0], [200], 128)


def test_exchange_records_round_trip_bit_for_bit():
    """The records of the job's one all-gather (include/fastrank.h: fr_pack_restart_records): fixed-size blocks
    of (valid, restart id, score, weights), padded with invalid records -- what train_model's multi-device path and
    bench.py's thread ranks put through RCCL.  Packing and unpacking keep every bit (denormals, -0.0, ids up to 2^32 - 1), drop
    the padding, and return restart order; the single-process RCCL call itself says why it does not apply here."""
    rng = np.random.default_rng(5)
    dim = 7
    parts = []
    ids = rng.permutation(11)
    for rank, chunk in enumerate((ids[:5], ids[5:5], ids[5:])):  # (a rank without restarts contributes padding only)
        parts.append([{"restart_id": int(i), "score": float(rng.random()), "weights": rng.uniform(-1, 1, dim).tolist()} for i in chunk])
    parts[0][0]["weights"][:3] = [-0.0, 5e-324, 1.7976931348623157e308]
    parts[2].append({"restart_id": 4294967295, "score": -0.0, "weights": [0.0] * dim})
    cap = max(len(p) for p in parts)
    blocks = np.stack([native.pack_restart_records(p, cap, dim) for p in parts])
    assert blocks.shape == (3, cap, 3 + dim) and blocks[1].tolist() == np.zeros((cap, 3 + dim)).tolist()
    assert (blocks[:, :, 0].sum(axis=1) == [len(p) for p in parts]).all()
    got = native.unpack_restart_records(blocks, dim)
    want = sorted((r for p in parts for r in p), key=lambda r: r["restart_id"])
    assert [r["restart_id"] for r in got] == [r["restart_id"] for r in want]
    for a, b in zip(got, want):
        assert np.array_equal(np.array([a["score"]] + a["weights"]).view(np.uint64), np.array([b["score"]] + b["weights"]).view(np.uint64))
    with pytest.raises(Exception, match="restarts for a block"):
        native.pack_restart_records(parts[2], 2, dim)
    # the RCCL call: not applicable on one rank / ranks that share a GPU (a reason, nothing run) ...
    rep, via = native.rccl_allgather_restarts([0], [parts[0]])
    assert rep["ran"] is False and via is None and "one rank" in rep["reason"]
    rep, via = native.rccl_allgather_restarts([0, 0, 0], parts)
    assert rep["ran"] is False and via is None and "share device 0" in rep["reason"] and rep["block_doubles"] == cap * (3 + dim)
    # ... and an error, not a fallback, when the ranks name distinct GPUs and the exchange cannot run (no GPU on this box)
    if native.device_count() < 3:
        with pytest.raises(Exception, match="not visible"):
            native.rccl_allgather_restarts([0, 1, 2], parts)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "fastrank_amd")):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "pyoracle" not in text and "liboracle" not in text and "oracle/" not in text, fn


def test_version_and_defaults():
    assert fr.__version__ == "0.7.0"  # tests/test_with_example_data.py:78
    d = fr.query_json("coordinate_ascent_defaults")
    assert d["measure"] == "ndcg" and d["judgments"] is None
    p = d["params"]["CoordinateAscent"]
    # src/coordinate_ascent.rs:25-41
    assert (p["num_restarts"], p["num_max_iterations"], p["step_base"], p["step_scale"], p["tolerance"]) == (
        5, 25, 0.05, 2.0, 0.001)
    assert p["normalize"] is True and p["quiet"] is False and p["init_random"] is True
    assert p["output_ensemble"] is False and 0 <= p["seed"] < 2 ** 64
    rf = fr.query_json("random_forest_defaults")["params"]["RandomForest"]
    assert rf["split_method"] == {"SquaredError": []} and rf["num_trees"] == 100 and rf["weight_trees"] is False
    with pytest.raises(Exception, match="unknown_query_str: nope"):
        fr.query_json("nope")


def test_train_request_round_trip():
    # tests/test_with_example_data.py:281-302 (train_req_object)
    rust = fr.TrainRequest.from_dict(fr.query_json("coordinate_ascent_defaults"))
    py = fr.TrainRequest()
    for _ in range(2):
        assert rust.measure == py.measure and rust.judgments == py.judgments
        for k in ("num_restarts", "num_max_iterations", "step_base", "step_scale", "tolerance", "init_random",
                  "output_ensemble", "quiet"):
            assert getattr(rust.params, k) == getattr(py.params, k)
        py = fr.TrainRequest.from_dict(py.to_dict())
    big = fr.TrainRequest.coordinate_ascent()
    big.params.seed = 2 ** 64 - 1
    assert big.clone().params.seed == 2 ** 64 - 1


def test_model_json_round_trip():
    # tests/test_with_example_data.py:203-214 (wire format)
    models = [
        {"Linear": {"weights": [0.1, -2.5e-07, 1e21, 3.0, 0.0]}},
        {"SingleFeature": {"fid": 3, "dir": -1.0}},
        {"DecisionTree": {"FeatureSplit": {"fid": 1, "split": 0.5, "lhs": {"LeafNode": 1.0},
                                           "rhs": {"FeatureSplit": {"fid": 0, "split": -2.0,
                                                                    "lhs": {"LeafNode": 0.25}, "rhs": {"LeafNode": 7.0}}}}}},
    ]
    models.append({"Ensemble": {"weights": [1.0, 0.5, 2.0], "models": list(models)}})
    for m in models:
        cm = fr.CModel.from_dict(m)
        assert cm.to_dict() == m
        assert fr.CModel.from_dict(cm.to_dict()).to_dict() == m
        assert str(cm) == str(m)
    with pytest.raises(Exception, match="unknown variant"):
        clib._unwrap(clib._load().model_from_json(b'{"Bogus": {}}'))
    with pytest.raises(Exception, match="error: Error"):
        clib._unwrap(clib._load().model_from_json(b'{"Linear": '))


def test_tiny_numbers_flush_to_zero_whatever_their_exponent_looks_like():
    """serde_json parses magnitudes below the subnormal range to 0.0; only magnitudes beyond f64 are "number out of
    range".  The decision must come from the value's decimal exponent, not from the sign written after `e`."""
    zeros = "0" * 400
    for text, want in (("0." + zeros + "1", 0.0), ("0." + zeros + "1e+5", 0.0), ("-0." + zeros + "1", -0.0), ("1e-400", 0.0),
                       ("0.0000001e-320", 0.0), ("1" + "0" * 20 + "e-400", 0.0)):
        raw = ('{"Linear":{"weights":[%s, 1.0]}}' % text).encode()
        w = fr.CModel(clib._unwrap(clib._load().model_from_json(raw)), None).to_dict()["Linear"]["weights"]
        assert w[0] == want and np.signbit(w[0]) == np.signbit(want) and w[1] == 1.0, text
    for text in ("1e400", "1" + "0" * 400, "0.1e310", "-123456789e301"):
        with pytest.raises(Exception, match="number out of range"):
            clib._unwrap(clib._load().model_from_json(('{"Linear":{"weights":[%s]}}' % text).encode()))


def test_serde_style_float_formatting():
    cm = fr.CModel.from_dict({"Linear": {"weights": [1.0, 0.05, 1e-7, 1.5e300, 123456.75, 1e16, -0.0, 0.001]}})
    raw = clib._take_str(clib._load().model_query_json(cm.pointer, b"to_json"))
    assert raw == '{"Linear":{"weights":[1.0,0.05,1e-7,1.5e300,123456.75,1e16,-0.0,0.001]}}'


def test_cqrel_round_trip_and_file(qrel_dict):
    q = fr.CQRel.from_dict(qrel_dict)
    assert q.to_dict() == qrel_dict  # tests/test_with_example_data.py:80-83
    assert q.queries() == set(qrel_dict.keys())
    one = sorted(qrel_dict.keys())[0]
    assert q.query_judgments(one) == qrel_dict[one]
    with pytest.raises(ValueError):
        q.query_judgments("no-such-query")
    f = fr.CQRel.load_file(os.path.join(GOLDEN, "data", "newsir18-entity.qrel"))
    assert f.to_dict() == qrel_dict
    with pytest.raises(Exception, match="NotFound"):
        fr.CQRel.load_file("/nonexistent/qrel")


def test_ranksvm_loader_introspection(known):
    # tests/test_with_example_data.py:90-104
    path = os.path.join(GOLDEN, "data", "trec_news_2018.train")
    rd = fr.CDataset.open_ranksvm(path)
    assert rd.queries() == set(known["expected_queries"])
    assert rd.feature_ids() == set(range(known["expected_d"]))
    assert rd.feature_names() == set(str(x) for x in range(known["expected_d"]))
    assert rd.num_features() == known["expected_d"] and rd.num_instances() == known["expected_n"]
    assert rd.is_sampled() is False
    named = fr.CDataset.open_ranksvm(path, os.path.join(GOLDEN, "data", "trec_news_2018.features.json"))
    assert named.feature_names() == set(known["feature_names"].values()) | {"0"}
    assert named.feature_name_to_index()["pagerank"] == 4
    with pytest.raises(Exception, match="NotFound"):
        fr.CDataset.open_ranksvm("/nonexistent/file.train")


def test_dense_dataset_and_sampling_views(trec, known):
    # tests/test_with_example_data.py:216-241 (introspection half) and :106-137
    X = trec["train_X"][:, 1:].copy()  # sklearn's zero_based=False loader drops column 0
    train = fr.CDataset.from_numpy(X, trec["train_y"], trec["train_qid"])
    assert train.is_sampled() is False
    assert train.num_features() == 5 and train.num_instances() == known["expected_n"]
    assert train.queries() == set(known["expected_queries"])
    assert train.feature_ids() == set(range(5)) and train.feature_names() == set("01234")
    ibq = train.instances_by_query()
    assert sorted(sum(ibq.values(), [])) == list(range(known["expected_n"]))
    assert all(v == sorted(v) for v in ibq.values())
    subset = sorted(known["expected_queries"])[:10]
    part = train.subsample_queries(subset)
    assert part.is_sampled() is True and part.queries() == set(subset)
    assert part.num_instances() == sum(len(ibq[q]) for q in subset)
    assert part.num_features() == 5
    with pytest.raises(ValueError):
        train.subsample_queries(["not-a-query"])
    one = train.subsample_feature_names(["3"])
    assert one.num_features() == 1 and one.feature_ids() == {3} and one.num_instances() == known["expected_n"]
    with pytest.raises(Exception, match=r"Missing Features: \{FeatureId\(77\)\}"):
        clib._unwrap(clib._load().dataset_feature_sampling(train.pointer, b"[77]"))
    with pytest.raises(Exception, match="No Features!"):
        clib._unwrap(clib._load().dataset_feature_sampling(train.pointer, b"[]"))
    with pytest.raises(Exception, match="unknown_dataset_query_str: wat"):
        train._query_json("wat")


def test_qid_out_of_u32_range_is_an_error():
    X = np.ones((2, 1), dtype=np.float32)
    for bad in (-1, 2 ** 32):
        with pytest.raises(Exception, match="TryFromIntError"):
            fr.CDataset.from_numpy(X, np.zeros(2), np.array([1, bad], dtype=np.int64))


def test_null_handles_report_reference_messages():
    L = clib._load()
    assert "Dataset pointer is null!" in clib._take_str(L.dataset_query_json(None, b"queries"))
    assert "Model pointer is null!" in clib._take_str(L.model_query_json(None, b"to_json"))
    assert "cqrel pointer is null!" in clib._take_str(L.cqrel_query_json(None, b"queries"))
    assert "NULL pointer: query_json_str" in clib._take_str(L.query_json(None))
    with pytest.raises(Exception, match="Dataset pointer is null!"):
        clib._unwrap(L.train_model(b"{}", None))


def test_bad_measure_and_unsupported_training_are_errors(trec):
    ds = fr.CDataset.from_numpy(trec["train_X"], trec["train_y"], trec["train_qid"])
    m = fr.CModel.from_dict({"Linear": {"weights": [0.0] * 6}})
    with pytest.raises(Exception, match='Invalid training measure: \\\\"p@5\\\\"'):
        ds.evaluate(m, "p@5")
    with pytest.raises(Exception, match="Couldn't parse after the @"):
        ds.evaluate(m, "ndcg@x")
    # the request's split_method must be serde's tuple-variant form (src/random_forest.rs:14-20): a bare string (the
    # Python dataclass default, as in the reference's fastrank/training.py:42) does not deserialize there either
    bad = fr.TrainRequest.random_forest()
    bad.params.split_method = "SquaredError"
    with pytest.raises(Exception, match="expected tuple variant"):
        ds.train_model(bad)
    bad.params.split_method = {"Entropy": []}
    with pytest.raises(Exception, match="unknown variant `Entropy`"):
        ds.train_model(bad)


@pytest.mark.skipif(native.device_count() > 0, reason="checks the no-GPU failure mode")
def test_compute_without_gpu_fails_loudly(trec):
    ds = fr.CDataset.from_numpy(trec["train_X"], trec["train_y"], trec["train_qid"])
    m = fr.CModel.from_dict({"Linear": {"weights": [1.0] * 6}})
    for call in (lambda: ds.evaluate(m, "ndcg@5"), lambda: m.predict_scores(ds),
                 lambda: ds.train_model(fr.TrainRequest.coordinate_ascent()),
                 lambda: ds.train_model(fr.TrainRequest.random_forest())):
        with pytest.raises(Exception, match="no MI355X/HIP device"):
            call()


def test_shard_bounds_cover_all_restarts():
    for R in (1, 5, 32, 256):
        for W in (1, 2, 3, 8):
            spans = [native.shard_bounds(R, r, W) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == R
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(e - b for b, e in spans) - min(e - b for b, e in spans) <= 1


def test_in_process_device_plan_partitions_like_the_multi_process_path():
    """train_model's fan-out over the GPUs of a node (FR_DEVICES): contiguous restart blocks, identical to
    native.shard_bounds; the device that already holds the dataset keeps it (slot 0), every other entry gets a
    device-to-device copy; never more devices than restarts; the same ordinal twice = two contexts."""
    for devices, count, R in (("0,1,2,3,4,5,6,7", 8, 256), ("0,1,2", 8, 32), ("0,0", 1, 32), ("3,1", 4, 5), ("0,1,2,3", 4, 2), ("2", 4, 7)):
        pl = native.device_plan(devices, count, R, primary_device=0)
        want = [int(x) for x in devices.split(",")][:max(1, min(R, len(devices.split(","))))]
        assert pl["devices"] == want
        k = len(want)
        assert [tuple(b) for b in pl["blocks"]] == [native.shard_bounds(R, i, k) for i in range(k)]
        assert pl["blocks"][0][0] == 0 and pl["blocks"][-1][1] == R
        assert sorted(pl["slots"]) == (list(range(k)) if 0 in want else list(range(1, k + 1)))
        if 0 in want:
            assert pl["slots"][want.index(0)] == 0  # the first listing of the primary's device reuses the primary
    assert native.device_plan("0,0", 1, 32)["slots"] == [0, 1]
    for bad, msg in (("0,9", "no device 9"), ("a,b", "not a list"), ("", "empty"), (",,", "empty")):
        with pytest.raises(Exception, match=msg):
            native.device_plan(bad, 8, 32)


def test_restart_queue_hands_every_restart_out_exactly_once():
    """train_model's devices pull restart ids from ONE queue (the reference's rayon pool, src/coordinate_ascent.rs:215-225):
    replayed here without a device.  Every id is started exactly once whatever the workers' speeds, a worker starts with its
    share and never holds more than its capacity, ids come out in ascending order, and a worker whose restarts converge
    early takes more of the rest than a slow one."""
    rng = np.random.default_rng(5)
    for R, workers, cap in ((256, 8, 32), (256, 8, 8), (5, 3, 2), (7, 2, 64), (1000, 8, 64), (0, 2, 4), (3, 8, 1)):
        lengths = rng.integers(1, 40, size=max(R, 1))
        out = native.restart_queue_replay(R, workers, cap, lengths)
        started = [i for w in out["order"] for i in w]
        assert sorted(started) == list(range(R)), (R, workers, cap)
        for w, ids in enumerate(out["order"]):
            assert ids[:cap] == sorted(ids[:cap]) and len(ids[:cap]) <= cap  # the initial share, pulled in one go
        if R >= workers * cap:
            assert all(len(ids) >= cap for ids in out["order"])
    # one fast worker (its restarts take 1 tick) next to one slow one (100 ticks): the fast one ends up with most restarts
    lengths = np.array([1 if (i // 4) % 2 == 0 else 100 for i in range(64)])
    out = native.restart_queue_replay(64, 2, 4, lengths)
    assert sorted(i for w in out["order"] for i in w) == list(range(64))
    assert max(out["ticks"]) < sum(lengths) / 4  # (a static split in two blocks would leave one worker with >= half the long ones)
    with pytest.raises(Exception, match="bad arguments"):
        native.restart_queue_replay(4, 0, 1, [1])


def test_fullrank_size_classes_hold_every_query_length():
    """kernels_fullverify.inc sorts a query as pl lanes of nl register-resident keys: every length up to 2048 must land in
    a class that holds it, the padding must stay small where the documents are, and a longer query never gets a
    smaller class."""
    L = clib._load()
    prev = 0
    waste = []
    for n in range(1, 2049):
        v = L.fr_debug_fullrank_class(n)
        nl, pl = v >> 16, v & 0xFFFF
        assert nl in (16, 32, 48, 64, 80, 96) and pl in (1, 2, 4, 8, 16, 32) and nl * pl >= n, (n, nl, pl)
        assert nl * pl >= prev
        prev = nl * pl
        waste.append(nl * pl / n)
    assert max(waste[63:1536]) <= 1.34  # 64 .. 1536 documents (power-of-two classes alone would reach 2.0)
    assert L.fr_debug_fullrank_class(65) == (80 << 16) | 1 and L.fr_debug_fullrank_class(129) == (80 << 16) | 2


def test_json_nesting_limit_like_serde():
    """serde_json stops at 128 nested containers with an error envelope; a model 200 000 levels deep must not
    take the process down (it used to overflow the stack in the recursive parser / tree conversions)."""
    deep = '{"DecisionTree":' + '{"FeatureSplit":{"fid":0,"split":0.5,"rhs":{"LeafNode":1.0},"lhs":' * 200000 + '{"LeafNode":0.0}' + "}}" * 200000 + "}"
    res = clib._load().model_from_json(deep.encode("utf-8"))
    with pytest.raises(Exception, match="recursion limit exceeded"):
        clib._unwrap(res)
    # 60 tree levels = 121 containers: still fine
    t = {"LeafNode": 0.0}
    for _ in range(60):
        t = {"FeatureSplit": {"fid": 0, "split": 0.5, "lhs": t, "rhs": {"LeafNode": 1.0}}}
    assert fr.CModel.from_dict({"DecisionTree": t}).to_dict() == {"DecisionTree": t}


def test_f32_gains_print_with_ryu_f32_layout():
    # ryu's f32 printer leaves plain notation at 1e13 / below 1e-6 (f64: 1e16 / 1e-5)
    q = fr.CQRel.from_dict({"1": {"a": 1e13, "b": 1e-6, "c": 1e12, "d": 1e-7, "e": 2.5, "f": 0.1}})
    raw = clib._take_str(clib._load().cqrel_query_json(q.pointer, b"1"))
    assert json.loads(raw) == pytest.approx({"a": 1e13, "b": 1e-6, "c": 1e12, "d": 1e-7, "e": 2.5, "f": 0.1}, rel=1e-6)
    assert '"a":1e13' in raw and '"b":0.000001' in raw and '"c":1000000000000.0' in raw and '"d":1e-7' in raw
    assert '"e":2.5' in raw and '"f":0.1' in raw


def test_integer_fields_are_range_checked_not_truncated(trec):
    ds = fr.CDataset.from_numpy(trec["train_X"], trec["train_y"], trec["train_qid"])
    req = fr.TrainRequest.coordinate_ascent()
    req.params.num_restarts = 2 ** 32 + 1  # would silently become 1 restart under a cast
    with pytest.raises(Exception, match="expected u32"):
        ds.train_model(req)
    with pytest.raises(Exception, match="expected u32"):
        fr.CModel.from_dict({"SingleFeature": {"fid": 2 ** 40, "dir": 1.0}})
    for text in ('{"Linear":{"weights":[0x10]}}', '{"Linear":{"weights":[1e999]}}'):
        with pytest.raises(Exception):
            clib._unwrap(clib._load().model_from_json(text.encode("utf-8")))
    m = fr.CModel.from_dict({"Linear": {"weights": [0.0] * 6}})
    with pytest.raises(Exception, match="Couldn't parse after the @"):
        ds.evaluate(m, "ndcg@99999999999999999999999")  # does not fit usize
    # 19 digits fit usize: parsed like the reference, the failure is then the missing GPU or nothing at all
    try:
        ds.evaluate(m, "ndcg@0000000000000000000005")
    except Exception as exc:
        assert "Couldn't parse" not in str(exc)


def test_nan_label_does_not_unwind_through_the_c_abi():
    X = np.zeros((3, 2), dtype=np.float32)
    y = np.array([1.0, np.nan, 0.0])
    qid = np.array([1, 1, 2], dtype=np.int64)
    ds = fr.CDataset.from_numpy(X, y, qid)
    with pytest.raises(ValueError, match="NaN label"):  # no abort: SIZE_MAX flags the bad dataset
        native.num_queries(ds)
