#!/usr/bin/env python3
"""bench.py -- coordinate-ascent NDCG@10 evaluations/sec on an MSLR-WEB30K-shaped matrix.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  * one rank per GPU.  Under `python -m torch.distributed.run --nproc-per-node N` (WORLD_SIZE set) the ranks are
    processes and the job's exchange is RCCL (torch.distributed, backend "nccl"; gloo if RCCL cannot start).  From a
    plain shell with --gpus N > 1 (no launcher) the ranks are N host threads of THIS process, one per device, each
    driving its own device-side copy of the dataset through the C ABI (ctypes releases the GIL; no PyTorch anywhere),
    and the train-to-convergence leg goes through the library's own train_model fan-out (FR_DEVICES, one shared
    restart queue) -- `--launcher torch` re-executes the script under torch.distributed.run instead;
  * a STEP is one lock-step coordinate-ascent tick: one fused HIP launch that evaluates every
    line-search candidate (1 + 2*25 = 51) of every live restart on this GPU (32 restarts per
    GPU -> 1632 reference `evaluate_mean` results per step), plus the host replay of the
    reference's sequential accept/early-break logic;
  * W untimed steps, then exactly K timed steps bracketed by barrier + device synchronize (torch.cuda.synchronize()
    where torch is loaded); time = max over ranks; value = useful evaluations of all ranks / time;
  * weak scaling (default): 32 restarts per GPU (BASELINE.json configs[2] at N=1, configs[3] at N=8),
    dataset replicated, restarts block-partitioned, no data-path collective;
    strong scaling (--restarts-total R): one fixed job of R restarts split over the ranks.
After the timed steps (never part of `value`):
  * a few lock-step steps (one launch per step, nothing else on the device): the ISOLATED launch the roofline
    object is computed from;
  * one whole job trained to convergence INCLUDING its single all-gather and the model selection (`e2e`:
    time-to-model, per-rank ticks / busy time / idle share -- the load imbalance restart sharding has;
    --steal-block B lets the ranks pull blocks of B restarts from a shared counter instead);
  * the CPU baseline (oracle/, rank 0, N=1 only).
Rank 0 prints ONE JSON line.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

SHAPES = {
    # name: (N docs, D features, Q queries, generator seed)   -- SURVEY.md section 8(d)
    "30k": (3_800_000, 136, 31_000, 20250929),
    "10k": (1_200_000, 136, 10_000, 20250930),
    "tiny": (60_000, 136, 500, 20250931),
}
HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
FP64_VALU_PEAK_TADDS = 39.3    # 78.6 TFLOP/s FP64 vector (FMA = 2 flop) -> 39.3 T adds/s
N_SIMD = 1024                  # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9
# sources whose SHA-1 the PMC figures of profiles/hbm_traffic.json are tied to (tools/pmc_bench.sh records them)
PMC_SOURCES = ("kernels_verify.inc", "kernels_fillnet.inc", "kernels_chain.inc", "kernels_order.inc", "kernels_linesearch.inc", "device_dataset.inc", "host.hpp")
DATA_KINDS = ("mslr", "ties", "tiesmix", "hard", "hardties")


def gen_mslr_shaped(seed, n, d, q, kind="mslr"):
    """Synthetic MSLR-like matrix: lognormal query lengths, 5-grade labels,
    columns cycling uniform / small-integer (ties) / heavy-tail / sparse, label signal in 16.
    kind="hard": the label signal is 0.03*label instead of 0.3*label (rankings stay noisy, many
    documents keep entering the top-k lists).  kind="ties": every column floor()-quantised (small
    integers, like most MSLR columns) and 20 % of each query's documents are exact duplicates
    (features and label) of another document of the query -- score ties the reference resolves
    by its gain/id tie-break only when their gains differ.  kind="tiesmix": like "ties", but a quarter of the duplicates
    keep their OWN label: exact score ties between different gains, which only the reference's tie-break (gain asc, id asc)
    orders -- those (query, group) pairs must be recomputed by the exact kernels.  kind="hardties": the stated
    "realistic" side line -- hard's weak label signal AND tiesmix's quantised columns and mixed-gain duplicates (real
    MSLR-WEB30K is integer-heavy with duplicated rows and trains to NDCG@10 ~ 0.45, not 0.99)."""
    rng = np.random.default_rng(seed)
    lens = np.clip(rng.lognormal(np.log(100.0), 0.6, q), 1, 1300)
    lens = np.maximum(1, np.floor(lens * (n / lens.sum()))).astype(np.int64)
    lens = np.minimum(lens, 1300)
    diff = int(n - lens.sum())
    order = rng.permutation(q)
    i = 0
    while diff != 0:
        k = order[i % q]
        if diff > 0 and lens[k] < 1300:
            lens[k] += 1
            diff -= 1
        elif diff < 0 and lens[k] > 1:
            lens[k] -= 1
            diff += 1
        i += 1
    qid = np.repeat(np.arange(1, q + 1, dtype=np.int64), lens)
    y = rng.choice(5, size=n, p=[0.515, 0.324, 0.134, 0.019, 0.008]).astype(np.float64)
    XT = np.empty((d, n), dtype=np.float32)
    signal = set(range(0, 128, 8)) if d >= 128 else set(range(0, d, 8))
    coef = 0.03 if kind in ("hard", "hardties") else 0.3
    for j in range(d):
        m = j % 4
        if m == 0:
            col = rng.random(n)
        elif m == 1:
            col = np.floor(rng.exponential(2.0, n))
        elif m == 2:
            col = rng.lognormal(0.0, 2.0, n)
        else:
            col = np.where(rng.random(n) < 0.7, 0.0, rng.random(n))
        if j in signal:
            col = col + coef * y
        if kind in ("ties", "tiesmix", "hardties"):
            col = np.floor(col * 4.0)
        XT[j] = col.astype(np.float32)
    X = np.ascontiguousarray(XT.T)
    del XT
    if kind in ("ties", "tiesmix", "hardties"):
        starts = np.concatenate(([0], np.cumsum(lens)[:-1]))
        dup = np.nonzero(rng.random(n) < 0.2)[0]
        qi = np.searchsorted(starts, dup, side="right") - 1
        src = starts[qi] + np.floor(rng.random(len(dup)) * lens[qi]).astype(np.int64)
        Xs, ys = X[src].copy(), y[src].copy()
        if kind in ("tiesmix", "hardties"):
            keep = rng.random(len(dup)) < 0.25
            ys = np.where(keep, y[dup], ys)
        X[dup], y[dup] = Xs, ys
    return X, y, qid


def source_sha1(names=None):
    """SHA-1 of the kernel / trainer sources the PMC constants depend on."""
    out = {}
    for name in (names or PMC_SOURCES):
        path = os.path.join(ROOT, "fastrank_amd", "csrc", name)
        out[name] = hashlib.sha1(open(path, "rb").read()).hexdigest() if os.path.exists(path) else None
    return out


def random_trees(rng, X, ntrees, max_depth):
    """config 5's forest: fid uniform, split = a uniform quantile of that column, leaves U[0,4)."""
    sample = X[rng.integers(0, X.shape[0], 4096)]

    def grow(depth):
        if depth >= max_depth or rng.random() < 0.05:
            return {"LeafNode": float(rng.uniform(0, 4))}
        f = int(rng.integers(0, X.shape[1]))
        return {"FeatureSplit": {"fid": f, "split": float(np.quantile(sample[:, f], rng.random())),
                                 "lhs": grow(depth + 1), "rhs": grow(depth + 1)}}

    return [grow(1) for _ in range(ntrees)]


class SingleComm:
    """One rank, no exchange.  `torch` is used only for torch.cuda.synchronize() (the contract's bracket)."""
    world, rank, backend = 1, 0, "none"

    def __init__(self, torch_mod=None):
        self.torch = torch_mod

    def barrier(self):
        pass

    def sync_device(self):
        from fastrank_amd import native
        native.synchronize()
        if self.torch is not None:
            self.torch.cuda.synchronize()

    def allreduce(self, values, op):
        return list(values)

    def allgather_rows(self, row):
        return [list(row)]

    def gather_restarts(self, mine, num_restarts):
        return sorted(mine, key=lambda r: r["restart_id"])

    def steal_blocks(self, num_restarts, block):
        for b in range(0, num_restarts, block):
            yield b, min(num_restarts, b + block)

    def close(self):
        pass


class TorchComm(SingleComm):
    """One process per GPU under torch.distributed.run; RCCL (backend "nccl") unless it cannot start, then gloo --
    the data path has no collective either way (barriers, three small reductions, one all-gather of restart records)."""

    def __init__(self, torch_mod, dist, backend, device):
        self.torch, self.dist, self.backend = torch_mod, dist, backend
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        self.dev = device if backend == "nccl" else torch_mod.device("cpu")

    def barrier(self):
        self.dist.barrier()

    def allreduce(self, values, op):
        t = self.torch.tensor(list(values), dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX if op == "max" else self.dist.ReduceOp.SUM)
        return t.cpu().tolist()

    def allgather_rows(self, row):
        t = self.torch.tensor(list(row), dtype=self.torch.float64, device=self.dev)
        parts = [self.torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(parts, t)
        return [x.cpu().tolist() for x in parts]

    def gather_restarts(self, mine, num_restarts):
        from fastrank_amd import native
        return native.gather_restarts(mine, num_restarts)

    def steal_blocks(self, num_restarts, block):
        from fastrank_amd import native
        return native.steal_blocks(num_restarts, block)

    def close(self):
        self.dist.barrier()
        self.dist.destroy_process_group()


class ThreadHub:
    """What the N rank threads of one process share: a barrier, a slot per rank, a counter."""

    def __init__(self, world):
        import threading
        self.world = world
        self.bar = threading.Barrier(world)
        self.slots = [None] * world
        self.lock = threading.Lock()
        self.counters = {}
        self.devices = list(range(world))  # device ordinal of every rank (run_threads sets it)
        self.rccl = (None, None)


class ThreadComm(SingleComm):
    """Rank = host thread of this process, one per device (no launcher, no PyTorch): exchanges go through host memory."""
    backend = "threads"

    def __init__(self, hub, rank):
        self.hub, self.rank, self.world, self.torch = hub, rank, hub.world, None
        self._jobs = 0

    def barrier(self):
        self.hub.bar.wait()

    def _exchange(self, item):
        self.hub.slots[self.rank] = item
        self.hub.bar.wait()
        got = list(self.hub.slots)
        self.hub.bar.wait()  # nobody overwrites a slot before everybody has read it
        return got

    def allreduce(self, values, op):
        rows = self._exchange(list(values))
        return [max(col) if op == "max" else sum(col) for col in zip(*rows)]  # (sum: in rank order on every rank)

    def allgather_rows(self, row):
        return [list(r) for r in self._exchange(list(row))]

    def gather_restarts(self, mine, num_restarts):
        """The job's one exchange.  The rank threads share an address space, so the blocks meet in host memory; when every
        rank drives a GPU of its own they ALSO go through ONE single-process RCCL all-gather over those GPUs
        (native.rccl_allgather_restarts: ncclCommInitAll + a grouped ncclAllGather), and the selection then reads what RCCL
        delivered -- which must equal the host gather bit for bit (an error otherwise, and an error, not a fallback, if RCCL
        cannot run on N distinct GPUs).  Ranks that share a GPU keep the host gather, with the reason recorded."""
        parts = self._exchange(list(mine))
        allr = sorted((r for part in parts for r in part), key=lambda r: r["restart_id"])
        if [r["restart_id"] for r in allr] != list(range(num_restarts)):
            raise RuntimeError("gather_restarts: expected restarts 0..{} exactly once".format(num_restarts - 1))
        if self.rank == 0:
            from fastrank_amd import native
            try:
                rep, via = native.rccl_allgather_restarts(list(self.hub.devices), parts)
                if via is not None and via != allr:
                    raise RuntimeError("the restarts RCCL delivered differ from the host gather")
                self.hub.rccl = (rep, None)
            except BaseException as exc:
                self.hub.rccl = (None, exc)
        self.hub.bar.wait()
        rep, exc = self.hub.rccl
        if exc is not None:
            raise RuntimeError("RCCL exchange failed: {}: {}".format(type(exc).__name__, exc))
        self.rccl_report = rep
        return allr

    def steal_blocks(self, num_restarts, block):
        key = self._jobs
        self._jobs += 1
        while True:
            with self.hub.lock:
                k = self.hub.counters.get(key, 0)
                self.hub.counters[key] = k + 1
            if k * block >= num_restarts:
                return
            yield k * block, min(num_restarts, (k + 1) * block)


TREE_PMC_SOURCES = ("kernels_treerank.inc", "kernels_tree.inc", "device_dataset.inc")


def load_tree_pmc(shape):
    """rocprofv3 --pmc figures of the tree-scoring kernel (profiles/hbm_traffic.json[shape]["trees"], captured by
    tools/pmc_trees.sh on `bench.py --measure trees`): static, tied to the SHA-1 of the kernel sources, `stale` when the
    sources built here differ."""
    meta = {"source": "profiles/hbm_traffic.json[{}][trees] (rocprofv3 --pmc passes over `python bench.py --measure trees`, "
                      "tools/pmc_trees.sh); static, not measured in this run".format(shape),
            "current_sha1": source_sha1(TREE_PMC_SOURCES)}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "hbm_traffic.json"))).get(shape, {}).get("trees") or {}
    except Exception as exc:
        meta.update({"stale": True, "note": "unreadable: {}".format(exc)})
        return {}, meta
    cap = tj.get("captured") or {}
    meta["captured_sha1"], meta["captured_round"] = cap.get("sha1"), cap.get("round")
    meta["stale"] = (not tj) or cap.get("sha1") != meta["current_sha1"]
    return tj, meta


def bench_trees(args, comm, fr, native, dataset, X, y, qid, n, d):
    """BASELINE.json configs[4] as a bench line (`--measure trees`): one step = one 500-tree ensemble pass over the
    30K shape with the batched tree-traversal kernel.  Ranks score replicas of the same matrix (document shards of one
    pass would be the production split; no exchange either way), so N > 1 is weak scaling over replicas."""
    world, rank = comm.world, comm.rank
    trees = random_trees(np.random.default_rng(7), X, 500, 8)
    model = fr.CModel.from_dict({"Ensemble": {"weights": [1.0] * len(trees), "models": [{"DecisionTree": t} for t in trees]}})
    for _ in range(max(1, args.warmup)):
        native.predict_scores_dense(model, dataset, 0)  # (0 rows copied back: the pass itself)
    comm.barrier()
    comm.sync_device()
    if rank == 0:
        native.profile_reset()
        native.profile_enable(True)
    comm.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        native.predict_scores_dense(model, dataset, 0)
    comm.barrier()
    comm.sync_device()
    elapsed = time.perf_counter() - t0
    native.profile_enable(False)
    elapsed = comm.allreduce([elapsed], "max")[0]
    if rank != 0:
        return
    stats = native.profile_stats()
    kname = "tree_rank_kernel" if "tree_rank_kernel" in stats else "tree_ensemble_kernel"
    k = stats[kname]
    from oracle import pyoracle as o  # the checker, outside the timed region

    m = min(20000, n)
    got = native.predict_scores_dense(model, dataset, n)[:m]
    ok = bool(np.array_equal(o.Dataset(X[:m], y[:m], qid[:m]).score_ensemble(trees, [1.0] * len(trees)), got))
    b_rf = n * (4 * d + 8)  # read X once, write one f64 score per document
    sec = k["avg_ms"] * 1e-3
    tj, pmc_meta = load_tree_pmc(args.shape)
    traffic = tj.get("bytes_per_pass")
    slots = N_SIMD * CLOCK_HZ * sec
    limiter = {"bound": "valu_issue + lds", "unit": "fraction of 1024 SIMDs x 2.4 GHz issue cycles",
               "frac": (tj["valu_insts_per_pass"] * 4.0 / slots) if tj.get("valu_insts_per_pass") else None,
               "lds_insts_per_pass": tj.get("lds_insts_per_pass"),
               "valu_active_frac_of_busy_cycles": tj.get("valu_active_frac_of_busy_cycles"),
               "lds_address_unit_busy_frac_of_busy_cycles": tj.get("lds_idx_active_frac_of_busy_cycles"),
               "lds_bank_conflict_frac_of_lds_active": tj.get("lds_bank_conflict_frac_of_lds_active"),
               "clock_ghz_under_load": tj.get("clock_ghz")}
    print(json.dumps({
        "metric": "tree-ensemble scoring passes/sec on MSLR-WEB30K shape (BASELINE.json configs[4])",
        "value": world * args.steps / elapsed, "unit": "passes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "500 random trees of depth <= 8 x %d docs x %d features (configs[4]); one step = one ensemble pass" % (n, d),
                   "parallelism": "replicas x{}".format(world), "launcher": comm.backend},
        "doc_trees_per_s": world * n * len(trees) * args.steps / elapsed,
        "roofline": {"bound": "hbm", "achieved": b_rf / sec / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": b_rf / sec / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "kernel": kname, "avg_launch_ms": k["avg_ms"],
                     "hbm_frac_measured": (traffic / sec / 1e9 / HBM_PEAK_GBS) if traffic else None,
                     "note": "traffic: (FETCH_SIZE x 2 + WRITE_SIZE) x 1024 per pass from the PMC capture named in `pmc` (null when none)"},
        "limiter": limiter,
        "pmc": pmc_meta,
        "parity_first_docs_bit_exact": ok,
    }))


def load_pmc(shape, usable):
    """Static rocprofv3 --pmc figures (profiles/hbm_traffic.json, captured by tools/pmc_bench.sh on THIS command).
    They are NOT measured in this run: the object says where they come from, which sources they were captured on,
    and `stale` when those sources differ from the ones built here (or when the capture carries no hashes)."""
    tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    meta = {"source": "profiles/hbm_traffic.json (rocprofv3 --pmc passes over `python bench.py --steps 20 --warmup 3`, "
                      "tools/pmc_bench.sh); static, not measured in this run", "current_sha1": source_sha1()}
    if not usable:
        meta.update({"stale": None, "note": "PMC figures exist only for the headline workload (mslr data, ndcg@10): none applied"})
        return {}, meta
    try:
        tj = json.load(open(tpath)).get(shape, {})
    except Exception as exc:
        meta.update({"stale": True, "note": "unreadable: {}".format(exc)})
        return {}, meta
    cap = tj.get("captured") or {}
    meta["captured_sha1"] = cap.get("sha1")
    meta["captured_round"] = cap.get("round")
    meta["stale"] = (cap.get("sha1") != meta["current_sha1"])
    return tj, meta


def oracle_check(X, y, qid, measure, model_dict, device_score):
    """After the e2e leg, outside everything timed: the selected model's weights through the ORACLE's evaluate_mean
    (oracle/fastrank_oracle.c restating src/evaluators.rs:173-224 + src/dense_dataset.rs:67-76; the HIP path's 256-query
    summation shape) on the whole matrix, against the score the device trained to -- bitwise.  One evaluation, ~2 s of
    one host core.  The oracle is the checker here, never the thing measured."""
    try:
        from oracle import pyoracle as o

        lin = model_dict.get("Linear")
        if lin is None:
            return {"ok": None, "note": "selected model is not Linear: {}".format(list(model_dict))}
        t0 = time.perf_counter()
        ds = o.Dataset(X, y, qid)
        o.set_mean_segment(o.DEVICE_MEAN_SEGMENT)
        try:
            got = float(ds.evaluate_mean(measure, np.asarray(lin["weights"], dtype=np.float64)))
        finally:
            o.set_mean_segment(0)
        return {"ok": bool(got == device_score), "oracle": got, "device": device_score, "abs_diff": abs(got - device_score),
                "seconds": time.perf_counter() - t0,
                "what": "oracle evaluate_mean({}) of the selected model's weights on all {} docs == the device's best_score, bitwise".format(measure, len(y))}
    except Exception as exc:  # (a check outside the timed region: its failure is reported in the line, not hidden)
        return {"ok": False, "error": "{}: {}".format(type(exc).__name__, str(exc)[:300])}


def cpu_baseline(X, y, qid, params, target_seconds, measure="ndcg@10", restarts=32):
    """Times oracle/ (the C restatement of the reference algorithm; kind="port") on the host cores: threads over
    restarts only, like rayon in the reference, threads = min(restarts of the config, cores).
    Bounded sample.  `wide` is a second, shorter sample with one thread per restart on up to 64 cores (what a caller
    with more restarts than the config's would get from the same host)."""
    from oracle import pyoracle as o

    cores = os.cpu_count() or 1          # the node's logical cores (stated next to the number)
    ds = o.Dataset(X, y, qid)

    def sample(threads, seconds, nsamples):
        p = dict(params)
        p["num_restarts"] = threads      # one restart per thread: restart-only parallelism, every thread busy

        def run(max_evals):
            t0 = time.perf_counter()
            _, _, evals, _ = ds.ca_learn(measure, p, threads=threads, max_evals_per_restart=max_evals)
            return int(evals.sum()), time.perf_counter() - t0

        _, t1 = run(2)
        per_round = t1 / 2.0
        m = int(max(2, min(200, round(seconds / nsamples / max(per_round, 1e-6)))))
        runs = [run(m) for _ in range(nsamples)]
        rates = sorted(nn / tt for nn, tt in runs)
        return {"value": rates[len(rates) // 2], "threads": threads, "samples": [nn / tt for nn, tt in runs],
                "min_median_max": [rates[0], rates[len(rates) // 2], rates[-1]],
                "sample": "{} samples, each {} restarts x {} evaluate_mean calls of the same CA run ({} evals in {:.1f} s altogether); "
                          "oracle/fastrank_oracle.c, per-call query regrouping hoisted".format(
                              nsamples, threads, m, sum(nn for nn, _ in runs), sum(tt for _, tt in runs)),
                "evals_per_s_per_core": rates[len(rates) // 2] / threads}

    threads = max(1, min(cores, int(restarts)))
    wide_threads = max(1, min(cores, 64))
    main = sample(threads, target_seconds * (0.65 if wide_threads != threads else 1.0), 3)
    out = {"value": main["value"], "unit": "evals/s", "cores": cores, "threads": threads, "kind": "port",
           "samples": main["samples"], "min_median_max": main["min_median_max"], "sample": main["sample"],
           "evals_per_s_per_core": main["evals_per_s_per_core"]}
    if wide_threads != threads:
        out["wide"] = sample(wide_threads, target_seconds * 0.35, 1)
    return out


def side_kind(args, comm, fr, native, req, kind, begin, end, headline_value):
    """A side line inside the driver's own record: the same timed region (warm-up, then exactly --steps pipelined ticks)
    on `kind` data (hardties = weak label signal + integer columns + duplicated rows: what real MSLR-WEB30K looks like to
    the bound-and-verify kernel), after everything that is timed.  Not part of `value`."""
    n, d, q, seed = SHAPES[args.shape]
    t0 = time.perf_counter()
    X, y, qid = gen_mslr_shaped(seed, n, d, q, kind)
    gen_s = time.perf_counter() - t0
    ds = fr.CDataset.from_numpy(X, y, qid)
    # (as for the headline, main(): a throwaway first job uploads the data and pre-heats the device, which has just idled through
    # `generate_s` of host work; the measured job starts from its own fresh state)
    preheat = 0 if os.environ.get("FR_BENCH_NO_PREHEAT") else 100
    seed = req.params.seed
    req.params.seed = 7
    pre = native.CoordinateAscentRun(ds, req, begin, end)
    try:
        if preheat and end > begin:
            pre.step(preheat)
            comm.sync_device()
    finally:
        pre.close()
        req.params.seed = seed
    run = native.CoordinateAscentRun(ds, req, begin, end)
    try:
        run.step(args.warmup)
        s0 = run.state()["stats"]
        comm.sync_device()
        t0 = time.perf_counter()
        run.step(args.steps)
        comm.sync_device()
        el = time.perf_counter() - t0
        s1 = run.state()["stats"]
        psteps = max(1, min(args.steps, 10))
        native.profile_reset()
        native.profile_enable(True)
        run.step(psteps)
        comm.sync_device()
        native.profile_enable(False)
        prof = native.profile_stats()
    finally:
        run.close()
        del ds
    useful = s1["useful_evals"] - s0["useful_evals"]
    vp, vr = s1["verify_pairs"] - s0["verify_pairs"], s1["verify_redone"] - s0["verify_redone"]
    value = useful / el
    return {"data": kind, "value": value, "unit": "evals/s", "ms_per_step": el * 1e3 / max(1, args.steps), "steps": args.steps,
            "of_headline": (value / headline_value) if headline_value else None,
            "redo_fraction": (vr / vp) if vp else None,
            "exact_group_share": (s1["exact_groups"] - s0["exact_groups"]) / max(1, s1["groups"] - s0["groups"]),
            "chain_runs_per_visit": ((s1["chain_runs"] - s0["chain_runs"]) / (s1["chain_visits"] - s0["chain_visits"])) if s1.get("chain_visits", 0) > s0.get("chain_visits", 0) else None,
            "exact_kernel_ms_per_step": prof.get("linesearch_ndcg_kernel", {"total_ms": 0.0})["total_ms"] / psteps,
            "kernels_ms_per_step": {k: v["total_ms"] / psteps for k, v in prof.items()},
            "generate_s": gen_s, "preheat_ticks": preheat}


def rccl_is_mandatory(backend, world, visible_gpus, pinned_device):
    """--backend nccl on a node where every rank has a GPU of its own: an RCCL failure is an error, not a reason to fall back
    to gloo (the fallback exists for smoke tests that put several ranks on ONE GPU, which RCCL refuses by design)."""
    return backend == "nccl" and world > 1 and pinned_device is None and visible_gpus >= world


def peer_copy_report(native, devices, nbytes=256 << 20):
    """One timed device-to-device copy (made like the dataset copies of train_model's fan-out) from the first device to every
    other device of the job: {can_access, enabled, gbps} per pair -- first-contact evidence for hipMemcpyPeerAsync over xGMI."""
    out, src = [], devices[0]
    for dst in list(dict.fromkeys(devices[1:])) or [src]:
        try:
            r = native.peer_copy(src, dst, nbytes)
            out.append({"src": src, "dst": dst, "can_access": r["can_access"], "enabled": r["enabled"], "gbps": r["gbps"], "ms": r["ms"],
                        "same_device": src == dst})
        except Exception as exc:
            out.append({"src": src, "dst": dst, "error": "{}: {}".format(type(exc).__name__, str(exc)[:200])})
    return out


def parse_rocm_smi(txt, ordinal):
    """(socket power in W, shader clock in MHz) of GPU[ordinal] from `rocm-smi --showpower --showclocks` text; None if absent."""
    import re
    watts = mhz = None
    for line in txt.splitlines():
        if not line.startswith("GPU[%d]" % ordinal):
            continue
        m = re.search(r"(?:Current Socket|Average) Graphics Package Power \(W\):\s*([0-9.]+)", line)
        if m:
            watts = float(m.group(1))
        m = re.search(r"sclk clock level:\s*\S+\s*\((\d+)Mhz\)", line)
        if m:
            mhz = int(m.group(1))
    return None if watts is None or mhz is None else (watts, mhz)


def power_sampler(stop, samples, ordinal):
    """Host thread: rocm-smi in a loop while the device steps (the tool takes ~0.3 s per call).  Never raises."""
    import subprocess
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                 text=True, timeout=10).stdout
        except Exception:
            return
        s = parse_rocm_smi(txt, ordinal)
        if s is not None and not stop.is_set():
            samples.append(s)


def power_cap_watts(ordinal):
    import re
    import subprocess
    try:
        txt = subprocess.run(["rocm-smi", "--showmaxpower"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, timeout=10).stdout
    except Exception:
        return None
    for line in txt.splitlines():
        m = re.search(r"Max Graphics Package Power \(W\):\s*([0-9.]+)", line)
        if m and line.startswith("GPU[%d]" % ordinal):
            return float(m.group(1))
    return None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--measure", default="ndcg@10", help="headline = ndcg@10 (BASELINE.json); others (ndcg, map, mrr, ndcg@k, trees) for side measurements")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--shape", default=os.environ.get("FR_BENCH_SHAPE", "30k"), choices=sorted(SHAPES))
    ap.add_argument("--data", default=os.environ.get("FR_BENCH_DATA", "mslr"), choices=DATA_KINDS,
                    help="mslr = headline; ties / tiesmix / hard / hardties = side measurements (see gen_mslr_shaped)")
    ap.add_argument("--restarts-per-gpu", type=int, default=32)
    ap.add_argument("--restarts-total", type=int, default=0,
                    help="strong scaling: one fixed job of this many restarts split over the ranks (0 = weak scaling)")
    ap.add_argument("--steal-block", type=int, default=0,
                    help="e2e leg: ranks pull blocks of this many restarts from a shared counter (0 = static block partition)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the train-to-convergence leg")
    ap.add_argument("--repeats", type=int, default=int(os.environ.get("FR_BENCH_REPEATS", "5")),
                    help="after the timed region, repeat it this many times on fresh jobs (value_runs: the spread `value` sits in)")
    ap.add_argument("--inprocess-devices", default=os.environ.get("FR_BENCH_INPROCESS", ""),
                    help="also train the e2e job through the library's own train_model with FR_DEVICES set to this list (e.g. "
                         "0,1,2,3,4,5,6,7, or 0,0 for two contexts on one GPU): the in-process fan-out a caller of the reference's "
                         "API gets.  With --launcher threads it defaults to the ranks' devices")
    ap.add_argument("--cpu-seconds", type=float, default=float(os.environ.get("FR_BENCH_CPU_SECONDS", "20")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side", action="store_true", help="skip the hardties side line (headline run on one GPU only)")
    ap.add_argument("--no-power", action="store_true", help="skip the power / clock leg (rocm-smi sampled while more steps run; N = 1 only)")
    ap.add_argument("--backend", default=os.environ.get("FR_BENCH_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="torch launcher: nccl = RCCL over xGMI (default; falls back to gloo if it cannot start); gloo for "
                         "single-GPU smoke tests of the N>1 path")
    ap.add_argument("--launcher", default=os.environ.get("FR_BENCH_LAUNCHER", "auto"), choices=["auto", "torch", "threads"],
                    help="auto: the ranks torch.distributed.run started when WORLD_SIZE is set, otherwise N host threads of this "
                         "process (one per device, no PyTorch); torch: re-execute under torch.distributed.run when no launcher "
                         "started this process; threads: always threads")
    return ap.parse_args()


def main():
    args = parse_args()
    env_world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    launcher = args.launcher
    if launcher == "auto":
        launcher = "torch" if env_world > 0 else ("threads" if args.gpus > 1 else "single")
    if launcher == "torch" and env_world == 0:
        if args.gpus == 1:
            launcher = "single"
        else:  # a plain shell asked for the one-process-per-GPU form: start it the way the driver does
            import socket

            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
                   "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
            os.execv(sys.executable, cmd)
    if launcher == "threads" and args.gpus == 1:
        launcher = "single"

    import fastrank_amd as fr
    from fastrank_amd import native

    n, d, q, seed = SHAPES[args.shape]
    if launcher == "threads":
        run_threads(args, fr, native)
        return

    import torch

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # FR_BENCH_DEVICE pins every rank to one ordinal (2-rank smoke test of the N>1 path on a 1-GPU box)
    dev_ordinal = int(os.environ.get("FR_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_ordinal)
    if launcher == "torch":
        import datetime
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend, note = args.backend, None
        must_be_rccl = rccl_is_mandatory(backend, int(os.environ.get("WORLD_SIZE", "1") or 1), torch.cuda.device_count(),
                                         os.environ.get("FR_BENCH_DEVICE"))
        if backend == "nccl":
            try:
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_ordinal), timeout=datetime.timedelta(seconds=300))
                probe = torch.ones(1, dtype=torch.float64, device=torch.device("cuda", dev_ordinal))
                dist.all_reduce(probe)
                torch.cuda.synchronize()
            except Exception as exc:  # RCCL could not start on this node: the job has no data-path collective, gloo carries the rest
                if must_be_rccl:  # one distinct GPU per rank and RCCL was asked for: a gloo run must not pass for an RCCL run
                    raise SystemExit("bench.py --backend nccl: RCCL failed to start with {} ranks on {} visible GPUs ({}: {}); not falling "
                                     "back to gloo -- rerun with --backend gloo to measure without RCCL".format(
                                         os.environ.get("WORLD_SIZE"), torch.cuda.device_count(), type(exc).__name__, str(exc)[:300]))
                note = "nccl (RCCL) failed to start: {}: {}; using gloo".format(type(exc).__name__, str(exc)[:200])
                print("[bench.py] " + note, file=sys.stderr, flush=True)
                try:
                    if dist.is_initialized():
                        dist.destroy_process_group()
                except Exception:
                    pass
                backend = "gloo"  # (same rendezvous: under torch.distributed.run the launcher's agent hosts the store)
                dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=120))
        else:
            dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=300))
        comm = TorchComm(torch, dist, backend, torch.device("cuda", dev_ordinal))
        comm.note = note
        if comm.world != args.gpus:
            raise SystemExit("bench.py --gpus {} was started with WORLD_SIZE={}: launch with torch.distributed.run "
                             "--nproc-per-node {} (or from a plain shell, which starts its own ranks)".format(args.gpus, comm.world, args.gpus))
    else:
        comm = SingleComm(torch)
    native.set_device(dev_ordinal)
    t0 = time.perf_counter()
    X, y, qid = gen_mslr_shaped(seed, n, d, q, args.data)
    gen_s = time.perf_counter() - t0
    try:
        rank_main(args, comm, fr, native, X, y, qid, gen_s, dev_ordinal, None)
    finally:
        comm.close()


def run_threads(args, fr, native):
    """--gpus N from a plain shell: N rank threads of this process, one per device (FR_BENCH_DEVICE pins all of them to
    one ordinal: N contexts on one GPU, the smoke test of this path on a one-GPU box)."""
    import threading

    if native.device_count() < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (no CPU fallback)")
    pin = os.environ.get("FR_BENCH_DEVICE")
    devices = [int(pin)] * args.gpus if pin is not None else list(range(args.gpus))
    if max(devices) >= native.device_count():
        raise SystemExit("bench.py --gpus {}: only {} device(s) visible (FR_BENCH_DEVICE=k puts every rank on device k)".format(
            args.gpus, native.device_count()))
    n, d, q, seed = SHAPES[args.shape]
    t0 = time.perf_counter()
    X, y, qid = gen_mslr_shaped(seed, n, d, q, args.data)  # one host copy, borrowed by every rank's dataset
    gen_s = time.perf_counter() - t0
    hub = ThreadHub(args.gpus)
    hub.devices = list(devices)
    errors = []

    def body(rank):
        try:
            native.set_device(devices[rank])
            rank_main(args, ThreadComm(hub, rank), fr, native, X, y, qid, gen_s, devices[rank], devices)
        except BaseException as exc:  # release the other ranks from their barriers, then report
            errors.append((rank, exc))
            hub.bar.abort()

    threads = [threading.Thread(target=body, args=(r,), name="rank%d" % r) for r in range(args.gpus)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    real = [(r, e) for r, e in errors if not isinstance(e, threading.BrokenBarrierError)] or errors
    if real:
        rank, exc = real[0]
        raise SystemExit("bench.py rank {} failed: {}: {}".format(rank, type(exc).__name__, exc))


def rank_main(args, comm, fr, native, X, y, qid, gen_s, dev_ordinal, thread_devices):
    """What one rank does (a process under torch.distributed.run, or a host thread of this process)."""
    world, rank = comm.world, comm.rank
    n, d = X.shape
    q = SHAPES[args.shape][2]
    dataset = fr.CDataset.from_numpy(X, y, qid)

    if args.measure == "trees":
        bench_trees(args, comm, fr, native, dataset, X, y, qid, n, d)
        return

    strong = args.restarts_total > 0
    R = args.restarts_total if strong else args.restarts_per_gpu * world
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = args.measure
    p = req.params
    p.num_restarts, p.num_max_iterations, p.step_base, p.step_scale = R, 25, 0.05, 2.0
    p.tolerance, p.normalize, p.init_random, p.seed, p.quiet = 0.001, True, True, 42, True
    begin, end = native.shard_bounds(R, rank, world)
    my_restarts = end - begin
    solo = rank == 0  # legs that use process-wide switches (HIP-event instrumentation, FR_LS_PIPELINE) run on rank 0 alone when the ranks are threads
    threads_mode = thread_devices is not None

    def barrier():
        comm.barrier()
        comm.sync_device()

    # Set-up, outside everything timed.  The first trainer on the dataset uploads it (tiles, column-major copy, order tables) and
    # forms the restarts' first evaluations and resident sums: `upload_and_init_s`.  That first trainer is a THROWAWAY job
    # (another seed) which also pre-heats the device: the process has just spent seconds generating data on the host with the
    # device idle, and W = 5 warm-up ticks (12 ms) do not always bring clocks and power management back (one first region
    # in ~20 measured 20 % low, its repeats normal).  It steps PREHEAT_TICKS ticks and is discarded; the measured job below
    # starts from its own fresh state (a dataset serves one trainer's resident sums at a time).  Reported in `setup`.
    PREHEAT_TICKS = 0 if os.environ.get("FR_BENCH_NO_PREHEAT") else 100
    t0 = time.perf_counter()
    p.seed = 7
    pre = native.CoordinateAscentRun(dataset, req, begin, end)  # uploads + initial evaluate_mean per restart
    comm.sync_device()
    upload_s = time.perf_counter() - t0
    t0 = time.perf_counter()
    if PREHEAT_TICKS and my_restarts:
        pre.step(PREHEAT_TICKS)
        comm.sync_device()
    pre.close()
    preheat_s = time.perf_counter() - t0
    p.seed = 42
    run = native.CoordinateAscentRun(dataset, req, begin, end)
    comm.sync_device()

    totals = {"useful_evals": 0, "raw_evals": 0, "verify_pairs": 0, "verify_redone": 0, "exact_ticks": 0, "ticks": 0,
              "line_searches": 0, "groups": 0, "exact_groups": 0, "verify_redo_entries": 0, "chain_runs": 0, "chain_visits": 0, "rank_slots_on": 0, "rank_slots_off": 0}
    jobs = {"finished": 0}

    def add_stats(stats):
        for k in totals:
            totals[k] += int(stats.get(k, 0) or 0)

    def advance(nsteps):
        """Runs exactly nsteps ticks; when every restart of the current job has converged a new job
        (next seed) is started so that long --steps requests still time real training work."""
        nonlocal run
        done = 0
        while done < nsteps:
            done += run.step(nsteps - done)
            if run.finished and done < nsteps:
                add_stats(run.state()["stats"])
                jobs["finished"] += 1
                p.seed += 1
                run.close()
                run = native.CoordinateAscentRun(dataset, req, begin, end)

    def snapshot():
        """Counters so far (finished jobs + the current one); reads the trainer's state, so it is called outside
        the timed region only."""
        st = run.state()["stats"]
        return {k: totals[k] + int(st.get(k, 0) or 0) for k in totals}

    advance(args.warmup)
    s0 = snapshot()
    # The timed region runs WITHOUT the library's HIP-event instrumentation (two events created and recorded around every
    # launch: ~50 us of host time per set and tick, which is on the critical path of the pipeline -- FR_BENCH_PROFILE_TIMED=1
    # puts it back); the per-kernel numbers come from two instrumented legs right after it, which are not part of `value`.
    profile_timed = os.environ.get("FR_BENCH_PROFILE_TIMED", "0") == "1"
    if solo:
        native.profile_reset()
        native.profile_enable(profile_timed)
    barrier()
    t0 = time.perf_counter()
    advance(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    if solo:
        native.profile_enable(False)
    s1 = snapshot()
    # The driver fixes --steps, so the timed region above is ~50 ms of device time.  The same region -- warm-up, then exactly
    # K pipelined steps between barriers -- is repeated on fresh jobs (new master seeds): `value_runs` in the line shows
    # the spread `value` sits in.  Not part of `value`.
    value_runs = [{"job_seed": 42, "useful": float(s1["useful_evals"] - s0["useful_evals"]), "elapsed": elapsed, "headline_region": True}]
    for rep in range(max(0, args.repeats)):
        add_stats(run.state()["stats"])
        run.close()
        p.seed = 1000 + rep
        run = native.CoordinateAscentRun(dataset, req, begin, end)
        advance(args.warmup)
        r0 = snapshot()
        barrier()
        tr = time.perf_counter()
        advance(args.steps)
        barrier()
        er = time.perf_counter() - tr
        r1 = snapshot()
        value_runs.append({"job_seed": 1000 + rep, "useful": float(r1["useful_evals"] - r0["useful_evals"]), "elapsed": er, "headline_region": False})
    s1b = snapshot()
    prof, prof_steps, prof_raw, iso, iso_evals = {}, 0, 0, None, 0
    if solo or not threads_mode:
        if profile_timed:
            prof, prof_steps, prof_raw = native.profile_stats(), args.steps, s1["raw_evals"] - s0["raw_evals"]
        else:  # the same pipelined stepping, instrumented (overlapping launches: durations of the sets' kernels in flight together)
            prof_steps = max(1, min(args.steps, 10))
            native.profile_reset()
            native.profile_enable(True)
            advance(prof_steps)
            comm.sync_device()
            native.profile_enable(False)
            prof = native.profile_stats()
            prof_raw = snapshot()["raw_evals"] - s1b["raw_evals"]
        # The timed steps keep several launches of the dominant kernel in flight (the restarts are stepped as three sets
        # on three streams), so their HIP-event durations overlap.  A few more steps in plain lock step (one launch per
        # step, nothing else on the device) give the duration of an ISOLATED launch: the roofline object below is computed
        # from it.  They are not part of `value`.  (tools/pmc_bench.sh counts the instructions and HBM bytes of both kinds
        # of launches of this same command.)
        ISO_STEPS = 4
        if not os.environ.get("FR_LS_PIPELINE"):
            os.environ["FR_LS_PIPELINE"] = "0"
            try:
                i0 = snapshot()
                native.profile_reset()
                native.profile_enable(True)
                advance(ISO_STEPS)
                comm.sync_device()
                native.profile_enable(False)
                iso = native.profile_stats()
                iso_evals = snapshot()["raw_evals"] - i0["raw_evals"]
            finally:
                del os.environ["FR_LS_PIPELINE"]
    # Socket power and shader clock while the device steps (the hot kernel runs the socket at its power cap, which
    # is why bytes and lanes not moved show up as time).  One GPU only, after everything that is timed: more pipelined steps
    # (until four samples are in, six seconds at most) with rocm-smi sampled by a host thread.  Not part of `value`.
    power = None
    smi_index = int(dev_ordinal)  # rocm-smi lists the node's GPUs: a visibility list renumbers them for this process
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        ids = [t.strip() for t in os.environ.get(var, "").split(",") if t.strip()]
        if ids:
            smi_index = int(ids[smi_index]) if smi_index < len(ids) and ids[smi_index].isdigit() else -1
    if world == 1 and solo and args.measure == "ndcg@10" and not args.no_power and smi_index >= 0:
        import threading
        samples, stop = [], threading.Event()
        th = threading.Thread(target=power_sampler, args=(stop, samples, smi_index), daemon=True)
        tp = time.perf_counter()
        th.start()
        while len(samples) < 4 and time.perf_counter() - tp < 6.0:  # (the tool's first call alone takes about a second)
            advance(200)
        comm.sync_device()
        stop.set()
        th.join(timeout=15)
        if samples:
            ws, cs = sorted(w for w, _ in samples), sorted(c for _, c in samples)
            power = {"socket_w_min_median_max": [ws[0], ws[len(ws) // 2], ws[-1]], "sclk_mhz_min_median_max": [cs[0], cs[len(cs) // 2], cs[-1]],
                     "cap_w": power_cap_watts(smi_index), "rocm_smi_index": smi_index, "samples": len(samples), "seconds": time.perf_counter() - tp,
                     "during": "pipelined steps after everything that is timed (until four samples are in); rocm-smi --showpower --showclocks sampled by a host thread"}
    if threads_mode:
        comm.barrier()  # (the other rank threads wait here while rank 0 runs its instrumented legs)

    useful = s1["useful_evals"] - s0["useful_evals"]
    raw = s1["raw_evals"] - s0["raw_evals"]
    elapsed_max = comm.allreduce([elapsed], "max")[0]
    useful_all, raw_all = comm.allreduce([float(useful), float(raw)], "sum")
    vr_elapsed = comm.allreduce([r["elapsed"] for r in value_runs], "max")
    vr_useful = comm.allreduce([r["useful"] for r in value_runs], "sum")
    best_so_far = max(r["score"] for r in run.state()["restarts"]) if my_restarts else float("nan")
    run.close()

    # ---- second leg: one whole job to convergence, INCLUDING the job's single exchange and the selection ----------
    e2e = None
    if not args.no_e2e:
        p.seed = 42
        barrier()
        t0 = time.perf_counter()
        mine, blocks, ticks, e_stats = [], [], 0, dict.fromkeys(totals, 0)
        chunks = comm.steal_blocks(R, args.steal_block) if args.steal_block > 0 else ([(begin, end)] if my_restarts else [])
        for b, e in chunks:
            job = native.CoordinateAscentRun(dataset, req, b, e)
            while not job.finished:
                ticks += job.step(1 << 20)
            st = job.state()
            job.close()
            mine.extend(st["restarts"])
            blocks.append([b, e])
            for k in e_stats:
                e_stats[k] += int(st["stats"].get(k, 0) or 0)
        comm.sync_device()
        busy_s = time.perf_counter() - t0
        allr = comm.gather_restarts(mine, R)
        model = native.select_model(allr, False)
        barrier()
        e2e_s = time.perf_counter() - t0
        rows = comm.allgather_rows([float(ticks), busy_s, float(e_stats["useful_evals"]), float(e_stats["raw_evals"]),
                                    float(len(mine)), float(e_stats["verify_pairs"]), float(e_stats["verify_redone"]),
                                    float(e_stats["exact_ticks"]), e2e_s, float(e_stats["line_searches"]),
                                    float(e_stats["exact_groups"]), float(e_stats["groups"]), float(dev_ordinal), float(rank)])
        wall = max(r[8] for r in rows)
        busy = [r[1] for r in rows]
        best = max(allr, key=lambda r: r["score"])
        e2e = {
            "what": "one job of {} restarts trained to convergence, {} of (restart, score, weights) and last-max "
                    "selection included (time-to-model); {}".format(
                        R, {"nccl": "RCCL all-gather", "gloo": "gloo all-gather", "threads": "gather in host memory", "none": "no exchange (one rank)"}[comm.backend],
                        "work stealing in blocks of {}".format(args.steal_block) if args.steal_block > 0 else "static block partition"),
            "wall_s": wall,
            "useful_evals": sum(r[2] for r in rows),
            "e2e_evals_per_s": sum(r[2] for r in rows) / wall,
            "per_rank_ticks": [int(r[0]) for r in rows],
            "per_rank_busy_s": busy,
            "per_rank_restarts": [int(r[4]) for r in rows],
            "idle_fraction": 1.0 - (sum(busy) / len(busy)) / max(max(busy), 1e-12),
            "exchange_and_select_s": wall - max(busy),
            "redo_fraction": (sum(r[6] for r in rows) / sum(r[5] for r in rows)) if sum(r[5] for r in rows) else None,
            "exact_line_search_share": (sum(r[7] for r in rows) / max(1.0, sum(r[9] for r in rows))),
            "exact_group_share": (sum(r[10] for r in rows) / max(1.0, sum(r[11] for r in rows))),
            # who took part, from the all-gather itself: one row per rank, its rank id and the device ordinal it drove
            "ranks_seen": len(rows),
            "rank_ids_seen": sorted(int(r[13]) for r in rows),
            "devices_seen": [int(r[12]) for r in rows],
            "best_score": best["score"],
            "model_sha1": hashlib.sha1(json.dumps(model.to_dict(), sort_keys=True).encode()).hexdigest(),
            # did the exchange go through RCCL?  thread launcher: one single-process all-gather over the ranks' GPUs, compared
            # with the host gather (null + reason when the ranks share a GPU); torch launcher: torch.distributed's backend
            "rccl": (getattr(comm, "rccl_report", None) if threads_mode else
                     {"ran": comm.backend == "nccl", "ranks": world, "via": "torch.distributed all_gather, backend {}".format(comm.backend),
                      "reason": None if comm.backend == "nccl" else ("one rank: nothing to exchange" if world == 1 else getattr(comm, "note", None))}),
        }
        e2e_top = e2e["e2e_evals_per_s"]
        if rank == 0:
            e2e["oracle_check"] = oracle_check(X, y, qid, args.measure, model.to_dict(), best["score"])
    else:
        e2e_top = None

    # ---- third leg: the same job through train_model itself, fanned out inside the library over FR_DEVICES (one shared
    #      restart queue, one host thread + trainer + device-to-device dataset copy per entry); with the thread launcher this
    #      is the product's own multi-GPU path over the ranks' devices, run by rank 0 while the other ranks wait
    inproc = None
    inproc_list = args.inprocess_devices or (",".join(str(x) for x in thread_devices) if threads_mode and not args.no_e2e else "")
    if inproc_list and rank == 0 and (world == 1 or threads_mode):
        p.seed = 42
        old_env = os.environ.get("FR_DEVICES")
        os.environ["FR_DEVICES"] = inproc_list
        try:
            runs = []
            for attempt in range(2):  # the first call also makes the device-to-device copies of the dataset
                t0 = time.perf_counter()
                m2 = dataset.train_model(req)
                wall = time.perf_counter() - t0
                st = native.last_train_stats()
                per_dev = st.get("per_device") or []
                secs = [x["seconds"] for x in per_dev]
                runs.append({"wall_s": wall, "useful_evals": st["useful_evals"], "e2e_evals_per_s": st["useful_evals"] / wall,
                             "devices": st["devices"], "ticks_longest_device": st["ticks"], "refills": st.get("refills"),
                             "per_device_ticks": [x["ticks"] for x in per_dev], "per_device_restarts": [x["restarts"] for x in per_dev],
                             "per_device_busy_s": secs,
                             "idle_fraction": (1.0 - (sum(secs) / len(secs)) / max(max(secs), 1e-12)) if secs else None})
            sha = hashlib.sha1(json.dumps(m2.to_dict(), sort_keys=True).encode()).hexdigest()
            inproc = {"what": "dataset.train_model(request) with FR_DEVICES={}: the listed devices pull the {} restarts from one queue "
                              "inside the call, one host thread + trainer + device-to-device dataset copy each".format(inproc_list, R),
                      "first_call_with_replication": runs[0], "second_call": runs[1], "model_sha1": sha,
                      "same_model_as_e2e_leg": (sha == e2e["model_sha1"]) if e2e is not None else None}
            native.release_replicas(dataset)
        except Exception as exc:  # (a side leg: its failure must not cost the bench line -- it is reported in it)
            inproc = {"what": "dataset.train_model(request) with FR_DEVICES={}".format(inproc_list),
                      "error": "{}: {}".format(type(exc).__name__, str(exc)[:500])}
        finally:
            if old_env is None:
                del os.environ["FR_DEVICES"]
            else:
                os.environ["FR_DEVICES"] = old_env
    peer = None
    if rank == 0 and world > 1:
        devs = list(thread_devices) if threads_mode else ([dev_ordinal] * world if os.environ.get("FR_BENCH_DEVICE") is not None else list(range(world)))
        peer = peer_copy_report(native, devs)
    if threads_mode:
        comm.barrier()

    if rank == 0:
        b_eval = n * (4 * d + 8)  # algorithmic bytes per evaluate_mean
        # dominant kernel: the bound-and-verify line search (the exact kernels only recompute the pairs it
        # could not verify); FR_LS_EXACT=1 runs measure the exact kernel instead
        dom = next((k for k in ("linesearch_verify_kernel", "fullrank_verify_kernel", "rr_verify_kernel", "rank_metric_kernel")
                    if k in prof), "linesearch_ndcg_kernel")
        ls = prof.get(dom, {"launches": 0, "total_ms": 0.0, "avg_ms": 0.0})
        exact = prof.get("linesearch_ndcg_kernel", {"launches": 0, "total_ms": 0.0, "avg_ms": 0.0})
        evals_per_launch = (prof_raw / max(1, ls["launches"])) if ls["launches"] else 0.0
        groups_per_launch = evals_per_launch / 51.0
        step_s = elapsed_max / max(1, args.steps)
        evals_per_step = raw / max(1, args.steps)
        headline = (args.data == "mslr" and args.measure == "ndcg@10")
        tj, pmc_meta = load_pmc(args.shape, headline and dom == "linesearch_verify_kernel")

        # -- roofline from the ISOLATED launch (nothing else on the device): algorithmic bytes of the evaluations one
        #    launch produces / its HIP-event duration
        iso_k = iso.get(dom) if iso else None
        if iso_k and iso_k["launches"]:
            iso_evals_per_launch = iso_evals / iso_k["launches"]
            iso_s = iso_k["avg_ms"] * 1e-3
            timing = "isolated lock-step launches after the timed region (one per step, nothing else on the device)"
        else:  # FR_LS_PIPELINE was set by the caller: the timed launches are what there is
            iso_evals_per_launch, iso_s = evals_per_launch, ls["avg_ms"] * 1e-3
            timing = "timed launches (FR_LS_PIPELINE set by the caller)"
        iso_groups = iso_evals_per_launch / 51.0
        achieved = (b_eval * iso_evals_per_launch / iso_s / 1e9) if iso_s > 0 else 0.0
        effective = b_eval * evals_per_step / step_s / 1e9 if step_s > 0 else 0.0
        traffic_iso = tj.get("bench_isolated_bytes_per_group")
        traffic_iso = traffic_iso * iso_groups if traffic_iso else None
        traffic_step = tj.get("bench_timed_bytes_per_group")
        traffic_step = traffic_step * evals_per_step / 51.0 if traffic_step else None
        roofline = {
            "bound": "hbm",
            "achieved": achieved,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS,
            "traffic": traffic_iso,
            "kernel": dom,
            "timing": timing,
            "avg_launch_ms": iso_s * 1e3,
            "launches": iso_k["launches"] if iso_k else ls["launches"],
            "evals_per_launch": iso_evals_per_launch,
            "algorithmic_bytes_per_launch": b_eval * iso_evals_per_launch,
            # SURVEY 8(d) formula on the whole timed step (all launches, gaps, host share): value x B_eval / peak
            "effective_frac": effective / HBM_PEAK_GBS,
            # real HBM bytes (PMC, static -- see `pmc`) over the launch / over the step, against the peak
            "hbm_frac_measured": (traffic_iso / iso_s / 1e9 / HBM_PEAK_GBS) if (traffic_iso and iso_s > 0) else None,
            "hbm_frac_measured_step": (traffic_step / step_s / 1e9 / HBM_PEAK_GBS) if (traffic_step and step_s > 0) else None,
            "traffic_per_step": traffic_step,
            "note": "batched: one pass over the resident sums and ONE feature column serves the 51 candidates of a line "
                    "group, so the algorithmic (per-eval, whole-matrix) bytes exceed real HBM traffic by construction and "
                    "frac >> 1 is not evidence of kernel quality; hbm_frac_measured and limiter are",
        }

        # -- what actually binds the kernel: VALU issue.  Wave-level instruction counts are PMC constants (see `pmc`).
        def valu_block(insts_per_group, mix, groups, seconds):
            """utilisation of the VALU issue slots: every wave instruction priced at 4 cycles.  That IS the rate of this
            chip for everything these kernels issue (tools/ubench/valu_rate.hip, profiles/r03_valu_rate.txt: v_min/max
            u32 4.1, v_min/fma/add f64 4.3, v_cndmask / v_lshl_or / v_mad_u32_u24 / v_mul_lo_u32 / DPP moves 4.2-4.3
            cycles per wave instruction and SIMD; only v_add_f32 / v_add_u32 issue faster, 2.4-2.6).  The second figure
            (f64-class at 4, the rest at 2) is kept for comparison with round 2's lines; it is a lower bound, not an estimate."""
            if not insts_per_group or seconds <= 0:
                return None
            total = insts_per_group * groups
            slots = N_SIMD * CLOCK_HZ * seconds
            out = {"wave_insts": total, "frac_all_at_4_cycles": total * 4.0 / slots}
            if mix:
                f64 = mix.get("f64_class_per_group", 0.0) * groups
                out["f64_class_wave_insts"] = f64
                out["frac_f64_at_4_rest_at_2"] = (f64 * 4.0 + (total - f64) * 2.0) / slots
            return out

        limiter = None
        if dom == "linesearch_verify_kernel":
            limiter = {
                "bound": "valu_issue",
                "unit": "fraction of 1024 SIMDs x 2.4 GHz issue cycles",
                "isolated_launch": valu_block(tj.get("bench_isolated_valu_insts_per_group"), tj.get("bench_isolated_valu_mix"),
                                              iso_groups, iso_s),
                "timed_step": valu_block(tj.get("bench_timed_valu_insts_per_group"), tj.get("bench_timed_valu_mix"),
                                         evals_per_step / 51.0, step_s),
                # measured by the SQ counters of the PMC pass itself (no pricing model): VALU-active quad-cycles over the
                # cycles the SIMDs were busy, and the clock those busy cycles imply (the chip runs this FP64 load at
                # ~1.9 GHz, so the 2.4 GHz slots above are not all reachable)
                "valu_active_frac_of_busy_cycles": tj.get("bench_isolated_valu_active_frac_of_busy_cycles"),
                "valu_issue_frac_of_busy_cycles_all_at_4": tj.get("bench_isolated_valu_issue_frac_of_busy_cycles"),
                "clock_ghz_under_load": tj.get("bench_isolated_clock_ghz"),
                "frac": None,
                "note": "bound-and-verify kernel on resident sums: three operations per document and restart for the "
                        "base dot product, then a per-document loop over the tile in candidate lanes (LDS broadcast, "
                        "FMA, compare; min/max chain for documents that enter a list); VALU-issue bound, "
                        "see DESIGN.md section 4",
            }
            # `frac`: share of the 1024 SIMDs x 2.4 GHz issue cycles of an isolated launch in which the VALU was active.
            # Measured: SQ_ACTIVE_INST_VALU (quad-cycles) over SQ_BUSY_CYCLES, scaled by the clock the busy cycles
            # imply.  It agrees with the instruction count priced at 4 cycles each (frac_all_at_4_cycles); the
            # instruction-class counters classify only 13 % of the kernel's VALU instructions (v_min/max/cmp_f64 are
            # in none of them), so frac_f64_at_4_rest_at_2 is a LOWER bound on issue utilisation, not an estimate.
            act, clk = limiter["valu_active_frac_of_busy_cycles"], limiter["clock_ghz_under_load"]
            il = limiter["isolated_launch"]
            if act and clk:
                limiter["frac"] = act * clk * 1e9 / CLOCK_HZ
                limiter["frac_source"] = "SQ_ACTIVE_INST_VALU x 4 / SQ_BUSY_CYCLES x clock_under_load / 2.4 GHz"
            elif il:
                limiter["frac"] = il["frac_all_at_4_cycles"]
                limiter["frac_source"] = "SQ_INSTS_VALU x 4 cycles"
        elif dom in ("linesearch_ndcg_kernel",):
            adds_per_launch = n * iso_groups * 51 * (d - 1) / 2.0  # avg shared prefix = half the features
            limiter = {
                "bound": "fp64_valu_add",
                "achieved": (adds_per_launch / iso_s / 1e12) if iso_s > 0 else 0.0,
                "peak": FP64_VALU_PEAK_TADDS,
                "unit": "Tadd/s",
                "frac": (adds_per_launch / iso_s / 1e12 / FP64_VALU_PEAK_TADDS) if iso_s > 0 else 0.0,
                "measured_ceiling": 35.4,  # pure v_add_f64 stream on this chip, tools/ubench/dpadd.hip
            }
        vp = float(s1["verify_pairs"] - s0["verify_pairs"])
        vr = float(s1["verify_redone"] - s0["verify_redone"])
        runs_eps = [u / e for u, e in zip(vr_useful, vr_elapsed)]
        srt = sorted(runs_eps)
        out = {
            "metric": "coordinate-ascent NDCG@10 evals/sec on MSLR-WEB30K shape" if (headline and args.shape == "30k")
            else "coordinate-ascent {} evals/sec on MSLR-WEB{} shape, data={} (side measurement)".format(args.measure, args.shape.upper(), args.data),
            "value": useful_all / elapsed_max,
            "unit": "evals/s",
            # the timed region again on fresh jobs (entry 0 IS `value`): K steps are ~50 ms, so one region alone is noise-prone
            "value_runs": runs_eps,
            "value_runs_job_seeds": [r["job_seed"] for r in value_runs],
            "value_runs_min_median_max": [srt[0], srt[len(srt) // 2] if len(srt) % 2 else 0.5 * (srt[len(srt) // 2 - 1] + srt[len(srt) // 2]), srt[-1]],
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" if args.data == "mslr" else "synthetic ({})".format(args.data),
            "config": {
                "workload": "synthetic MSLR-WEB{} shape: {} docs x {} features x {} queries, coordinate ascent "
                            "{}, {} x 25 steps/coord (configs[{}])".format(
                                args.shape.upper(), n, d, q, args.measure.upper() if args.measure.startswith("ndcg") else args.measure,
                                "{} restarts in total".format(R) if strong else "{} restarts/GPU".format(args.restarts_per_gpu),
                                1 if args.shape == "10k" else (3 if R == 256 and world == 8 else 2)),
                "restarts_total": R,
                "parallelism": "restart-sharded x{} (dataset replicated)".format(world),
                "launcher": {"none": "single process", "threads": "one host thread per device in one process (no PyTorch)"}.get(
                    comm.backend, "torch.distributed.run, backend {}".format(comm.backend)),
                "collective_note": getattr(comm, "note", None),
                "evals_per_step_per_gpu": evals_per_step,
                "launches_per_step": ls["launches"] / max(1, prof_steps),
                "groups_per_launch": groups_per_launch,
                "jobs_finished_inside_timed_region": jobs["finished"],
            },
            "raw_evals_per_s": raw_all / elapsed_max,
            "useful_fraction": useful_all / max(1.0, raw_all),
            "e2e_evals_per_s": e2e_top,
            "roofline": roofline,
            "limiter": limiter,
            "pmc": pmc_meta,
            "power": power,
            "verify": {
                "pairs": vp, "redone": vr, "redo_fraction": (vr / vp) if vp else None,
                "ticks": s1["ticks"] - s0["ticks"],
                "line_searches": s1["line_searches"] - s0["line_searches"],
                # line searches evaluated by the exact kernels alone (after one with > 25 % redone pairs)
                "exact_line_search_share": (s1["exact_ticks"] - s0["exact_ticks"]) / max(1, s1["line_searches"] - s0["line_searches"]),
                # single restarts' line searches routed to the exact kernel (their last verified one left > 25 % undecided)
                "exact_group_share": (s1["exact_groups"] - s0["exact_groups"]) / max(1, s1["groups"] - s0["groups"]),
                # 16-candidate slices the exact kernel recomputed per listed (query, group) pair (4 = all of a 51-candidate group)
                "redo_slices_per_pair": ((s1["verify_redo_entries"] - s0["verify_redo_entries"]) / vr) if vr else None,
                "exact_kernel_ms_per_step": exact["total_ms"] / max(1, prof_steps),
                # documents that made a wave of the verify kernel run its insertion chain, per (document, group) visit (device counter)
                "chain_runs_per_visit": ((s1["chain_runs"] - s0["chain_runs"]) / (s1["chain_visits"] - s0["chain_visits"])) if s1["chain_visits"] > s0["chain_visits"] else None,
                # restarts (all jobs of this process so far) whose R-rank tables were found worth keeping / not worth it by that counter
                "rank_slots_on_off": [s1b["rank_slots_on"], s1b["rank_slots_off"]],
            },
            "per_launch_overlapped": {"avg_launch_ms": ls["avg_ms"], "launches": ls["launches"],
                                      "note": "HIP-event durations of pipelined launches (three sets in flight, so they overlap), from {} "
                                              "instrumented steps {}".format(prof_steps, "= the timed region" if profile_timed else "after the timed region")},
            "kernels_ms": {k: v["total_ms"] for k, v in prof.items()},
            "instrumented_steps": prof_steps,
            "timed_region_instrumented": profile_timed,
            "e2e": e2e,
            "inprocess": inproc,
            "peer_copy": peer,
            "setup": {"generate_s": gen_s, "upload_and_init_s": upload_s, "best_score_so_far": best_so_far,
                      "preheat_ticks": PREHEAT_TICKS, "preheat_s": preheat_s},
        }
        if world == 1 and headline and not args.no_side:
            p.seed = 42
            # (the headline's dataset goes first -- free_dataset, as a caller done with it would: its tiles, order tables and
            # line-search contexts with their streams otherwise stay beside the side line's)
            dataset = None
            import gc
            gc.collect()
            try:
                out["side"] = {"hardties": side_kind(args, comm, fr, native, req, "hardties", begin, end, out["value"])}
            except Exception as exc:  # (a side leg: its failure must not cost the bench line -- it is reported in it)
                out["side"] = {"hardties": {"error": "{}: {}".format(type(exc).__name__, str(exc)[:300])}}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(X, y, qid, p.to_dict(), args.cpu_seconds, args.measure, args.restarts_per_gpu)
        print(json.dumps(out))
        sys.stdout.flush()


if __name__ == "__main__":
    main()
