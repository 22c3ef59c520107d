"""world_size-2 gloo test of the N>1 path (restart sharding, the single all_gather, the
deterministic selection).  The per-rank shard results come from the CPU oracle here (the GPU
trainer needs a device); the gather/selection code is the product's own."""
import json
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from fastrank_amd import native
from oracle import pyoracle as o
from tests.conftest import GOLDEN

PARAMS = dict(num_restarts=5, num_max_iterations=3, step_base=0.05, step_scale=2.0, tolerance=0.001,
              seed=42, normalize=True, init_random=True)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out_dir, output_ensemble):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = np.load(os.path.join(GOLDEN, "trec_news_2018.npz"))
    ds = o.Dataset(d["train_X"], d["train_y"], d["train_qid"])
    R = PARAMS["num_restarts"]
    begin, end = native.shard_bounds(R, rank, world)
    scores, weights, _, err = ds.ca_learn("ndcg@5", PARAMS, threads=1, restart_range=(begin, end))
    assert err == 0
    mine = [{"restart_id": r, "score": float(scores[r]), "weights": weights[r].tolist()} for r in range(begin, end)]
    allr = native.gather_restarts(mine, R)
    assert [r["restart_id"] for r in allr] == list(range(R))
    model = native.select_model(allr, output_ensemble)
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as fh:
        json.dump(model.to_dict(), fh)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("output_ensemble", [False, True])
def test_two_rank_gather_and_select_matches_single_process(tmp_path, output_ensemble):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path), output_ensemble), nprocs=2, join=True)
    got = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(2)]
    assert got[0] == got[1], "every rank must select the same model"
    d = np.load(os.path.join(GOLDEN, "trec_news_2018.npz"))
    ds = o.Dataset(d["train_X"], d["train_y"], d["train_qid"])
    scores, weights, _, _ = ds.ca_learn("ndcg@5", PARAMS, threads=2)
    if output_ensemble:
        ens = got[0]["Ensemble"]
        assert ens["weights"] == scores.tolist()
        for k, m in enumerate(ens["models"]):
            s = np.abs(weights[k]).sum()
            assert m["Linear"]["weights"] == (weights[k] / s if s > 0 else weights[k]).tolist()
    else:
        assert got[0] == {"Linear": {"weights": weights[o.select_best(scores)].tolist()}}


def test_gather_without_process_group_is_identity():
    mine = [{"restart_id": 1, "score": 0.2, "weights": [1.0]}, {"restart_id": 0, "score": 0.1, "weights": [2.0]}]
    assert [r["restart_id"] for r in native.gather_restarts(mine, 2)] == [0, 1]


def _sum_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the query-sharded trainer's exchange: per-candidate sums of this rank's queries
    rng = np.random.default_rng(100 + rank)
    mine = rng.uniform(0, 1, 1632) * (1e-9 if rank == 0 else 1e9)  # badly conditioned on purpose
    total = native.rank_ordered_sum(mine)
    np.save(os.path.join(out_dir, "sum%d.npy" % rank), total)
    np.save(os.path.join(out_dir, "mine%d.npy" % rank), mine)
    dist.barrier()
    dist.destroy_process_group()


def test_rank_ordered_sum_is_bitwise_identical_on_every_rank(tmp_path):
    world = 3
    mp.spawn(_sum_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    sums = [np.load(tmp_path / ("sum%d.npy" % r)) for r in range(world)]
    mine = [np.load(tmp_path / ("mine%d.npy" % r)) for r in range(world)]
    assert all(np.array_equal(sums[0], s) for s in sums[1:])
    assert np.array_equal(sums[0], (mine[0] + mine[1]) + mine[2]), "added in rank order"
    assert np.array_equal(native.rank_ordered_sum(mine[0]), mine[0])  # no process group: identity


def test_oracle_sharded_mean_shape():
    """mean = (sum over shards, in rank order, of the shard's segment-shaped sum) / nq -- the shape
    fr_ca_begin_query_shard fixes (include/fastrank.h)."""
    v = np.random.default_rng(3).uniform(0, 1, 1000)
    try:
        o.set_mean_segment(256)
        o.set_mean_shards([0, 300, 900])

        def seg(x):
            t = 0.0
            for s0 in range(0, len(x), 256):
                part = 0.0
                for e in x[s0:s0 + 256]:
                    part += e
                t += part
            return t

        exp = ((seg(v[:300]) + seg(v[300:900])) + seg(v[900:])) / 1000.0
        assert o.mean(v) == exp
        o.set_mean_shards(None)
        assert o.mean(v) == seg(v) / 1000.0
    finally:
        o.set_mean_shards(None)
        o.set_mean_segment(0)


def _steal_worker(rank, world, port, out_dir, R, block):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = np.load(os.path.join(GOLDEN, "trec_news_2018.npz"))
    ds = o.Dataset(d["train_X"], d["train_y"], d["train_qid"])
    params = dict(PARAMS, num_restarts=R)
    mine, blocks = [], []
    for begin, end in native.steal_blocks(R, block):
        if rank == 0 and not blocks:
            import time
            time.sleep(0.3)  # a slow rank: the others must take its share
        scores, weights, _, err = ds.ca_learn("ndcg@5", params, threads=1, restart_range=(begin, end))
        assert err == 0
        mine += [{"restart_id": r, "score": float(scores[r]), "weights": weights[r].tolist()} for r in range(begin, end)]
        blocks.append([begin, end])
    allr = native.gather_restarts(mine, R)
    model = native.select_model(allr, False)
    # a second job on the same process group draws from a fresh counter
    again = [list(b) for b in native.steal_blocks(3, 2)]
    with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as fh:
        json.dump({"model": model.to_dict(), "blocks": blocks, "again": again,
                   "restarts": [[r["restart_id"], r["score"]] for r in allr]}, fh)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_work_stealing_blocks_cover_every_restart_once_and_select_the_same_model(tmp_path, world):
    R, block = 7, 2  # ragged last block
    mp.spawn(_steal_worker, args=(world, _free_port(), str(tmp_path), R, block), nprocs=world, join=True)
    got = [json.load(open(tmp_path / ("rank%d.json" % r))) for r in range(world)]
    taken = sorted(tuple(b) for g in got for b in g["blocks"])
    assert taken == [(0, 2), (2, 4), (4, 6), (6, 7)], "each block pulled by exactly one rank"
    assert sorted(tuple(b) for g in got for b in g["again"]) == [(0, 2), (2, 3)]
    assert all(g["model"] == got[0]["model"] and g["restarts"] == got[0]["restarts"] for g in got[1:])
    d = np.load(os.path.join(GOLDEN, "trec_news_2018.npz"))
    ds = o.Dataset(d["train_X"], d["train_y"], d["train_qid"])
    scores, weights, _, _ = ds.ca_learn("ndcg@5", dict(PARAMS, num_restarts=R), threads=2)
    assert got[0]["model"] == {"Linear": {"weights": weights[o.select_best(scores)].tolist()}}
    assert [s for _, s in got[0]["restarts"]] == scores.tolist()


def test_steal_blocks_without_process_group_is_the_plain_block_list():
    assert list(native.steal_blocks(10, 4)) == [(0, 4), (4, 8), (8, 10)]
