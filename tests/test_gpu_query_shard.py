"""Query-sharded coordinate ascent (SURVEY.md 8e fallback; include/fastrank.h fr_ca_begin_query_shard):
two processes share one GPU over gloo, each holding half of the queries, and must both arrive at
the restarts the CPU oracle finds on the whole dataset with the same summation shape.

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import json
import os
import socket

import numpy as np
import pytest

from oracle import pyoracle as o
from tests.conftest import synth_dataset

pytestmark = pytest.mark.gpu

DATA = dict(seed=53, n=9000, d=10, q=700, max_len=50)
PARAMS = dict(num_restarts=2, num_max_iterations=5, step_base=0.05, step_scale=2.0, tolerance=0.001,
              seed=11, normalize=True, init_random=True, output_ensemble=False, quiet=True)
MEASURES = ["ndcg@10", "map", "ndcg"]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _split(qid, world):
    """Contiguous blocks of queries (first-appearance order), balanced by query count."""
    _, first = np.unique(qid, return_index=True)
    order = qid[np.sort(first)]
    bounds = [len(order) * r // world for r in range(world + 1)]
    return order, bounds


def _data(long_query):
    X, y, qid = synth_dataset(DATA["seed"], DATA["n"], DATA["d"], DATA["q"], max_len=DATA["max_len"])
    if long_query:
        # one 2 300-document query at the end: it lands in the last rank's block and is longer than the
        # rank-counting kernels take (2 048), so that rank alone cannot run the fused full-ranking path
        X2, y2, _ = synth_dataset(DATA["seed"] + 1, 2300, DATA["d"], 1, max_len=2300)
        X = np.concatenate([X, X2])
        y = np.concatenate([y, y2])
        qid = np.concatenate([qid, np.full(2300, qid.max() + 1, dtype=np.int64)])
    return X, y, qid


def _worker(rank, world, port, out_dir, long_query=False, measures=None):
    import torch.distributed as dist

    import fastrank_amd as fr
    from fastrank_amd import native

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    native.set_device(0)
    X, y, qid = _data(long_query)
    order, bounds = _split(qid, world)
    mask = np.isin(qid, order[bounds[rank]:bounds[rank + 1]])
    shard = fr.CDataset.from_numpy(np.ascontiguousarray(X[mask]), np.ascontiguousarray(y[mask]),
                                   np.ascontiguousarray(qid[mask]))
    out = {}
    for measure in (measures or MEASURES):
        req = fr.TrainRequest.coordinate_ascent()
        req.measure = measure
        req.params = fr.CoordinateAscentParams(**PARAMS)
        run = native.QueryShardedRun(shard, req)
        assert run.total_queries == len(order)
        while not run.finished:
            run.step(1000)
        st = run.state()
        run.close()
        model = native.train_model_query_sharded(shard, req)
        out[measure] = {"restarts": st["restarts"], "path": st["stats"]["path"], "model": model.to_dict()}
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as fh:
        json.dump(out, fh)
    dist.barrier()
    dist.destroy_process_group()


def test_two_query_shards_match_the_oracle_on_the_whole_dataset(tmp_path):
    import torch.multiprocessing as mp

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(world)]
    assert got[0] == got[1], "every rank must end with the same restarts and model"
    X, y, qid = synth_dataset(DATA["seed"], DATA["n"], DATA["d"], DATA["q"], max_len=DATA["max_len"])
    order, bounds = _split(qid, world)
    c = o.Dataset(X, y, qid)
    assert np.array_equal(c.query_ids(), order)
    try:
        o.set_mean_segment(o.DEVICE_MEAN_SEGMENT)
        o.set_mean_shards(bounds[:-1])
        for measure, want_path in zip(MEASURES, ["fused_linesearch", "fused_fullrank", "fused_fullrank"]):
            exp_s, exp_w, _, err = c.ca_learn(measure, PARAMS, threads=2)
            assert err == 0
            res = got[0][measure]
            assert res["path"] == want_path
            for r in res["restarts"]:
                assert r["score"] == exp_s[r["restart_id"]], measure
                assert r["weights"] == exp_w[r["restart_id"]].tolist(), measure
            assert res["model"] == {"Linear": {"weights": exp_w[o.select_best(exp_s)].tolist()}}
    finally:
        o.set_mean_shards(None)
        o.set_mean_segment(0)


def test_shards_with_different_capabilities_agree_on_one_path(tmp_path):
    """Only the last shard holds a query longer than the fused full-ranking kernels take.  Left to
    itself that rank would exchange one sum per candidate and the other rank 64 per line group; the
    ranks must settle on the path all of them support (here: the general sort evaluator) and still
    reach the oracle's restarts."""
    import torch.multiprocessing as mp

    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), True, ["map"]), nprocs=world, join=True)
    got = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(world)]
    assert got[0] == got[1]
    assert got[0]["map"]["path"] == "generic_sort"
    X, y, qid = _data(True)
    order, bounds = _split(qid, world)
    c = o.Dataset(X, y, qid)
    try:
        o.set_mean_segment(o.DEVICE_MEAN_SEGMENT)
        o.set_mean_shards(bounds[:-1])
        exp_s, exp_w, _, err = c.ca_learn("map", PARAMS, threads=2)
        assert err == 0
        for r in got[0]["map"]["restarts"]:
            assert r["score"] == exp_s[r["restart_id"]]
            assert r["weights"] == exp_w[r["restart_id"]].tolist()
    finally:
        o.set_mean_shards(None)
        o.set_mean_segment(0)
