"""Extensions of the C ABI that have no reference counterpart (include/fastrank.h part 2):
dense result buffers, the batched line-search evaluator, restart-sharded and query-sharded multi-GPU
training and HIP-event kernel timing.  numpy is used only to hold buffers that cross the ABI."""
import ctypes as C
import json
from typing import Dict, List, Optional, Tuple

import numpy as np

from .clib import ALLREDUCE_SUM_FN, CDataset, CModel, CQRel, _json_reply, _load, _status, _take_str, _unwrap


def device_count() -> int:
    return int(_load().fr_device_count())


def set_device(ordinal: int) -> None:
    if _load().fr_set_device(int(ordinal)) != 0:
        raise RuntimeError("fr_set_device({}) failed".format(ordinal))


def version() -> str:
    return _load().fr_version().decode("utf-8")


def synchronize() -> None:
    if _load().fr_synchronize() != 0:
        raise RuntimeError("device synchronize failed")


def num_queries(dataset: CDataset) -> int:
    nq = int(_load().fr_dataset_num_queries(dataset.pointer))
    if nq == C.c_size_t(-1).value:
        raise ValueError("dataset cannot be grouped by query (a NaN label?): compute calls report the reason")
    return nq


def device_info(dataset: CDataset) -> Dict:
    """How the dataset lives on the device (builds the device form if needed); see include/fastrank.h."""
    return _json_reply(_load().fr_dataset_device_info(dataset.pointer))


def predict_scores_dense(model: CModel, dataset: CDataset, n_total: Optional[int] = None) -> np.ndarray:
    """Scores indexed by instance id (NaN where the id is not part of a sampled dataset)."""
    n = int(n_total if n_total is not None else _load().fr_dataset_num_instances(dataset.pointer))
    if n_total is None and dataset.is_sampled():
        n = 1 + max(max(ids) for ids in dataset.instances_by_query().values())
    out = np.full(n, np.nan, dtype=np.float64)
    _status(_load().fr_predict_scores_dense(model.pointer, dataset.pointer, out.ctypes.data, n))
    return out


def evaluate_dense(model: CModel, dataset: CDataset, evaluator: str, qrel: Optional[CQRel] = None) -> Tuple[List[str], np.ndarray]:
    nq = num_queries(dataset)
    out = np.zeros(nq, dtype=np.float64)
    qids_ptr = C.c_void_p()
    _status(
        _load().fr_evaluate_dense(
            model.pointer, dataset.pointer, None if qrel is None else qrel.pointer, evaluator.encode("utf-8"),
            out.ctypes.data, nq, C.byref(qids_ptr),
        )
    )
    return json.loads(_take_str(qids_ptr.value)), out


def rank_order(model: CModel, dataset: CDataset) -> Tuple[np.ndarray, np.ndarray]:
    """(instance ids grouped by query, best first; offsets[nq+1]) under the reference's
    (score desc, gain asc, id asc) order."""
    nq = num_queries(dataset)
    n = int(_load().fr_dataset_num_instances(dataset.pointer))
    ids = np.zeros(n, dtype=np.uint32)
    offs = np.zeros(nq + 1, dtype=np.uint64)
    _status(_load().fr_rank_order(model.pointer, dataset.pointer, ids.ctypes.data, n, offs.ctypes.data, nq + 1))
    return ids, offs


def evaluate_candidates(dataset: CDataset, evaluator: str, features, base_weights, candidates: List,
                        qrel: Optional[CQRel] = None, per_query: bool = False):
    """Batched evaluate_mean for line-search candidates.  features[g], base_weights[g][d],
    candidates[g] = list of <=64 values for w[features[g]].  Returns means as a list of arrays
    (and, if per_query, the [nq][G*64] matrix)."""
    G = len(features)
    feats = np.ascontiguousarray(features, dtype=np.uint32)
    base = np.ascontiguousarray(base_weights, dtype=np.float64)
    assert base.ndim == 2 and base.shape[0] == G
    ncand = np.asarray([len(c) for c in candidates], dtype=np.uint32)
    cand = np.zeros((G, 64), dtype=np.float64)
    for g, c in enumerate(candidates):
        cand[g, : len(c)] = c
    means = np.zeros((G, 64), dtype=np.float64)
    pq = np.zeros((num_queries(dataset), G * 64), dtype=np.float64) if per_query else None
    _status(
        _load().fr_evaluate_candidates(
            dataset.pointer, None if qrel is None else qrel.pointer, evaluator.encode("utf-8"), G,
            feats.ctypes.data, base.ctypes.data, ncand.ctypes.data, cand.ctypes.data, means.ctypes.data,
            None if pq is None else pq.ctypes.data,
        )
    )
    out = [means[g, : ncand[g]].copy() for g in range(G)]
    return (out, pq) if per_query else out


def profile_enable(on: bool = True) -> None:
    _load().fr_profile_enable(1 if on else 0)


def profile_reset() -> None:
    _load().fr_profile_reset()


def profile_stats() -> Dict[str, Dict[str, float]]:
    """{kernel: {"launches", "total_ms", "avg_ms"}} measured with HIP events on the launch stream."""
    rows = _json_reply(_load().fr_profile_json())
    return {
        r["kernel"]: {"launches": r["launches"], "total_ms": r["total_ms"],
                      "avg_ms": r["total_ms"] / max(1, r["launches"])}
        for r in rows
    }


def last_train_stats() -> Dict:
    return _json_reply(_load().fr_last_train_stats())


def train_model_shard(dataset: CDataset, train_req, restart_begin: int, restart_end: int) -> Dict:
    """Train restarts [begin, end) of the request on this process's GPU."""
    request = json.dumps(train_req.to_dict()).encode("utf-8")
    return _json_reply(_load().fr_train_model_shard(request, dataset.pointer, restart_begin, restart_end))


class CoordinateAscentRun:
    """Steppable coordinate ascent on this process's GPU (fr_ca_begin / fr_ca_step / fr_ca_state).
    One tick = one fused launch over every candidate of every live restart's line search."""

    def __init__(self, dataset: CDataset, train_req, restart_begin: int = 0, restart_end: Optional[int] = None):
        if restart_end is None:
            restart_end = int(train_req.params.num_restarts)
        err = C.c_void_p()
        request = json.dumps(train_req.to_dict()).encode("utf-8")
        self.pointer = _load().fr_ca_begin(request, dataset.pointer, restart_begin, restart_end, C.byref(err))
        if not self.pointer:
            _status(err.value)
            raise RuntimeError("fr_ca_begin failed")
        self._dataset = dataset  # keep the dataset (and its numpy buffers) alive
        self.finished = False

    def step(self, ticks: int) -> int:
        done = C.c_uint64(0)
        fin = C.c_int(0)
        _status(_load().fr_ca_step(self.pointer, int(ticks), C.byref(done), C.byref(fin)))
        self.finished = bool(fin.value)
        return int(done.value)

    def state(self) -> Dict:
        return _json_reply(_load().fr_ca_state(self.pointer))

    def close(self):
        if self.pointer:
            _load().fr_ca_free(self.pointer)
            self.pointer = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rank_ordered_sum(values: "np.ndarray", group=None) -> "np.ndarray":
    """Sum of a float64 vector over all ranks, added in rank order, so every rank gets the same
    bits (a ring all-reduce does not promise that).  One all_gather of len(values) doubles over
    RCCL/xGMI (backend "nccl") or gloo."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return np.array(values, dtype=np.float64)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    mine = torch.from_numpy(np.ascontiguousarray(values, dtype=np.float64)).to(dev)
    parts = [torch.empty_like(mine) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, mine, group=group)
    total = parts[0].cpu().numpy().copy()
    for t in parts[1:]:
        total = total + t.cpu().numpy()
    return total


class QueryShardedRun(CoordinateAscentRun):
    """Coordinate ascent with the QUERIES sharded over the ranks (the fallback for runs
    with fewer restarts than GPUs): `dataset` is this rank's block of queries, every rank runs all
    restarts in lock step, and after each device evaluation the per-candidate sums are combined with
    rank_ordered_sum (<= 13 KB per tick).  Every rank ends with identical restarts."""

    def __init__(self, dataset: CDataset, train_req, total_queries: Optional[int] = None, group=None):
        local_q = num_queries(dataset)
        if total_queries is None:
            total_queries = int(rank_ordered_sum(np.array([float(local_q)]), group)[0])

        def reduce_cb(_ctx, values, n):
            try:
                arr = np.ctypeslib.as_array(values, shape=(n,))
                arr[:] = rank_ordered_sum(arr, group)
                return 0
            except Exception as exc:  # surfaces as an error envelope from fr_ca_step
                self._callback_error = exc
                return 1

        self._callback_error = None
        self._callback = ALLREDUCE_SUM_FN(reduce_cb)  # keep alive as long as the trainer
        err = C.c_void_p()
        request = json.dumps(train_req.to_dict()).encode("utf-8")
        self.pointer = _load().fr_ca_begin_query_shard(request, dataset.pointer, int(total_queries), self._callback, None, C.byref(err))
        if not self.pointer:
            _status(err.value)
            raise RuntimeError("fr_ca_begin_query_shard failed")
        self._dataset = dataset
        self.finished = False
        self.total_queries = int(total_queries)


def train_model_query_sharded(dataset: CDataset, train_req, total_queries: Optional[int] = None, group=None) -> CModel:
    """Trains on query shards: call on every rank with that rank's block of queries (e.g.
    `full.subsample_queries(my_qids)` or a per-rank file).  Returns the same model on every rank."""
    run = QueryShardedRun(dataset, train_req, total_queries, group)
    try:
        while not run.finished:
            run.step(1 << 20)
        restarts = run.state()["restarts"]
    finally:
        run.close()
    model = select_model(restarts, bool(train_req.params.output_ensemble))
    model.params = train_req
    return model


def select_model(restarts: List[Dict], output_ensemble: bool = False) -> CModel:
    """src/coordinate_ascent.rs:232-252 over gathered restarts (last maximum wins ties)."""
    payload = json.dumps(restarts).encode("utf-8")
    return CModel(_unwrap(_load().fr_select_model(payload, 1 if output_ensemble else 0)))


def device_plan(devices: str, device_count: int, num_restarts: int, primary_device: int = 0) -> Dict:
    """How the library's own train_model would spread a request over the devices of FR_DEVICES=`devices` (no device needed)."""
    return _json_reply(_load().fr_debug_device_plan(devices.encode("utf-8"), device_count, num_restarts, primary_device))


def restart_queue_replay(num_restarts: int, n_workers: int, capacity: int, lengths) -> Dict:
    """The library's restart queue replayed without a device (fr_debug_restart_queue): which worker starts which restart
    ids, in which order, when restart r converges after lengths[r % len(lengths)] ticks."""
    arr = np.ascontiguousarray(lengths, dtype=np.uint32)
    return _json_reply(_load().fr_debug_restart_queue(int(num_restarts), int(n_workers), int(capacity), arr.ctypes.data, len(arr)))


def peer_copy(src_device: int, dst_device: int, nbytes: int = 256 << 20) -> Dict:
    """One timed device-to-device copy made like train_model's dataset copies: {"can_access", "enabled", "ms", "gbps", ...}."""
    return _json_reply(_load().fr_debug_peer_copy(int(src_device), int(dst_device), int(nbytes)))


def pack_restart_records(restarts: List[Dict], cap: int, dim: int) -> "np.ndarray":
    """A rank's restarts as the fixed-size block of the job's one exchange: `cap` records of 3 + dim doubles -- valid, restart
    id, score, weights (include/fastrank.h: fr_pack_restart_records; the library's own multi-device train_model packs with the
    same code)."""
    out = np.zeros((int(cap), 3 + int(dim)), dtype=np.float64)
    _status(_load().fr_pack_restart_records(json.dumps(restarts).encode("utf-8"), int(cap), int(dim), out.ctypes.data))
    return out


def unpack_restart_records(records: "np.ndarray", dim: int) -> List[Dict]:
    """The valid records of any number of blocks, as restarts in restart order."""
    arr = np.ascontiguousarray(records, dtype=np.float64).reshape(-1, 3 + int(dim))
    return _json_reply(_load().fr_unpack_restart_records(arr.ctypes.data, arr.shape[0], int(dim)))


def rccl_allgather_restarts(devices: List[int], parts: List[List[Dict]]) -> Tuple[Dict, Optional[List[Dict]]]:
    """The job's one exchange as ONE single-process RCCL all-gather over `devices` (one rank per entry; parts[i] = the restarts
    rank i trained): (report, restarts).  report["ran"] is False with a "reason" when RCCL does not apply (one rank; ranks that
    share a GPU) and restarts is then None; with N distinct GPUs a failure raises (include/fastrank.h: fr_rccl_allgather)."""
    n = len(devices)
    cap = max([1] + [len(p) for p in parts])
    dim = max([0] + [len(r["weights"]) for p in parts for r in p])
    blocks = np.stack([pack_restart_records(p, cap, dim) for p in parts]) if n else np.zeros((0, cap, 3 + dim))
    out = np.empty_like(blocks)
    devs = np.ascontiguousarray(devices, dtype=np.int32)
    rep = _json_reply(_load().fr_rccl_allgather(devs.ctypes.data, n, blocks.ctypes.data, cap * (3 + dim), out.ctypes.data))
    if not rep.get("ran"):
        return rep, None
    if not np.array_equal(out.view(np.uint64), blocks.view(np.uint64)):
        raise RuntimeError("rccl_allgather_restarts: rank 0's gathered records differ from the blocks the ranks contributed")
    return rep, unpack_restart_records(out, dim)


def walk_tiles(run_pos, run_q0, run_q1, qstart, qlen, np_positions: int) -> Dict:
    """The walk tiles the device form of a dataset gets for this layout of runs and queries (no device needed)."""
    arrs = [np.ascontiguousarray(a, dtype=np.uint32) for a in (run_pos, run_q0, run_q1, qstart, qlen)]
    return _json_reply(_load().fr_debug_walk_tiles(arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, len(arrs[0]),
                                                   arrs[3].ctypes.data, arrs[4].ctypes.data, len(arrs[3]), int(np_positions)))


def rccl_selftest(device: int = 0) -> Dict:
    """A one-rank RCCL communicator on `device` through the library's exchange code (dlopen, ncclCommInitAll, grouped
    ncclAllGather, compare, destroy): the report of fr_rccl_allgather."""
    return _json_reply(_load().fr_debug_rccl_selftest(int(device)))


def release_replicas(dataset: CDataset) -> int:
    """Frees the copies of the dataset train_model made on other devices / in other contexts; returns how many."""
    return int(_load().fr_dataset_release_replicas(dataset.pointer))


def shard_bounds(num_restarts: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block partition of restart ids over ranks (first ranks take the remainder)."""
    base, rem = divmod(num_restarts, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def gather_restarts(mine: List[Dict], num_restarts: int, group=None) -> List[Dict]:
    """The job's single exchange: one all_gather of fixed-size (valid, restart_id, score, weights)
    records over RCCL/xGMI (backend "nccl") or gloo (CPU test rigs).  Returns every rank's
    restarts in restart order, identically on every rank.  Ranks may hold different numbers of
    restarts (block partition with a remainder, or work stealing): the record count is the maximum
    over ranks, agreed in the same small all-reduce that agrees the weight dimension."""
    import torch
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        return sorted(mine, key=lambda r: r["restart_id"])
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    shape_t = torch.tensor([max((len(r["weights"]) for r in mine), default=0), len(mine)], dtype=torch.int64, device=dev)
    dist.all_reduce(shape_t, op=dist.ReduceOp.MAX, group=group)
    dim, per_rank = int(shape_t[0].item()), max(1, int(shape_t[1].item()))
    buf = torch.zeros((per_rank, 3 + dim), dtype=torch.float64)
    for k, r in enumerate(mine):
        buf[k, 0] = 1.0
        buf[k, 1] = float(r["restart_id"])
        buf[k, 2] = r["score"]
        buf[k, 3:3 + len(r["weights"])] = torch.tensor(r["weights"], dtype=torch.float64)
    buf = buf.to(dev)
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf, group=group)
    restarts = []
    for t in gathered:
        for row in t.cpu().tolist():
            if row[0] == 1.0:
                restarts.append({"restart_id": int(row[1]), "score": row[2], "weights": row[3:]})
    restarts.sort(key=lambda r: r["restart_id"])
    if len(restarts) != num_restarts or any(r["restart_id"] != i for i, r in enumerate(restarts)):
        raise RuntimeError("gather_restarts: expected restarts 0..{} exactly once, got {}".format(
            num_restarts - 1, [r["restart_id"] for r in restarts]))
    return restarts


_steal_jobs = 0


def steal_blocks(num_restarts: int, block: int, group=None, store=None):
    """Work stealing over restart blocks (restarts converge after different
    numbers of ticks, so a static partition leaves GPUs idle at the end): yields [begin, end)
    blocks of `block` consecutive restart ids pulled from ONE shared counter -- an atomic add on
    the process group's rendezvous store (TCP), no data-path collective.  Which rank trains which
    block does not matter: a restart's trajectory depends only on its child seed
    (src/coordinate_ascent.rs:211-225)."""
    import torch.distributed as dist

    global _steal_jobs
    key = "fastrank_amd/steal/{}".format(_steal_jobs)  # every rank calls in the same order: same key
    _steal_jobs += 1
    if not dist.is_available() or not dist.is_initialized():
        for b in range(0, num_restarts, block):
            yield b, min(num_restarts, b + block)
        return
    if store is None:
        store = dist.distributed_c10d._get_default_store()
    while True:
        k = int(store.add(key, 1)) - 1
        if k * block >= num_restarts:
            return
        yield k * block, min(num_restarts, (k + 1) * block)


def train_model_work_stealing(dataset: CDataset, train_req, block: int = 8, group=None, store=None, stats: Optional[Dict] = None) -> CModel:
    """Like train_model_distributed, but the ranks pull restart blocks from a shared counter instead of
    taking one static block each.  Same model on every rank, identical to the unsharded run."""
    R = int(train_req.params.num_restarts)
    mine, blocks = [], []
    for begin, end in steal_blocks(R, block, group, store):
        shard = train_model_shard(dataset, train_req, begin, end)
        mine.extend(shard["restarts"])
        blocks.append((begin, end))
    if stats is not None:
        stats["blocks"] = blocks
    restarts = gather_restarts(mine, R, group)
    model = select_model(restarts, bool(train_req.params.output_ensemble))
    model.params = train_req
    return model


def train_model_distributed(dataset: CDataset, train_req, group=None) -> CModel:
    """One process per GPU: every rank holds a replica of the dataset, trains its block of random
    restarts, then ONE all-gather (gather_restarts) and the same deterministic selection on every
    rank.  No data-path collective."""
    import torch.distributed as dist

    if not dist.is_available() or not dist.is_initialized():
        return dataset.train_model(train_req)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    R = int(train_req.params.num_restarts)
    begin, end = shard_bounds(R, rank, world)
    shard = train_model_shard(dataset, train_req, begin, end)
    restarts = gather_restarts(shard["restarts"], R, group)
    model = select_model(restarts, bool(train_req.params.output_ensemble))
    model.params = train_req
    return model
