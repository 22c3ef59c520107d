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


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist

    import fastrank_amd as fr
    from fastrank_amd import native

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    native.set_device(0)
    X, y, qid = synth_dataset(DATA["seed"], DATA["n"], DATA["d"], DATA["q"], max_len=DATA["max_len"])
    order, bounds = _split(qid, world)
    mask = np.isin(qid, order[bounds[rank]:bounds[rank + 1]])
    shard = fr.CDataset.from_numpy(np.ascontiguousarray(X[mask]), np.ascontiguousarray(y[mask]),
                                   np.ascontiguousarray(qid[mask]))
    out = {}
    for measure in MEASURES:
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
