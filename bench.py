#!/usr/bin/env python3
"""bench.py -- coordinate-ascent NDCG@10 evaluations/sec on an MSLR-WEB30K-shaped matrix.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  * one process per GPU (torch.distributed / RCCL when WORLD_SIZE > 1);
  * a STEP is one lock-step coordinate-ascent tick: one fused HIP launch that evaluates every
    line-search candidate (1 + 2*25 = 51) of every live restart on this GPU (32 restarts per
    GPU -> 1632 reference `evaluate_mean` results per step), plus the host replay of the
    reference's sequential accept/early-break logic;
  * W untimed steps, then exactly K timed steps bracketed by barrier + torch.cuda.synchronize();
    time = max over ranks; value = useful evaluations of all ranks / time;
  * weak scaling: 32 restarts per GPU (BASELINE.json configs[2] at N=1, configs[3] at N=8),
    dataset replicated, restarts block-partitioned, one all-gather of (restart, score, weights)
    at the end of training (not part of a step).
Rank 0 prints ONE JSON line.
"""
import argparse
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


def gen_mslr_shaped(seed, n, d, q):
    """Synthetic MSLR-like matrix: lognormal query lengths, 5-grade labels,
    columns cycling uniform / small-integer (ties) / heavy-tail / sparse, label signal in 16."""
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
            col = col + 0.3 * y
        XT[j] = col.astype(np.float32)
    X = np.ascontiguousarray(XT.T)
    del XT
    return X, y, qid


def cpu_baseline(X, y, qid, params, target_seconds, measure="ndcg@10"):
    """Times oracle/ (the C restatement of the reference algorithm; kind="port") on the host
    cores: threads over restarts only, like rayon in the reference.  Bounded sample."""
    from oracle import pyoracle as o

    cores = os.cpu_count() or 1
    threads = max(1, min(cores, 64))
    ds = o.Dataset(X, y, qid)
    p = dict(params)
    p["num_restarts"] = threads

    def run(max_evals):
        t0 = time.perf_counter()
        _, _, evals, _ = ds.ca_learn(measure, p, threads=threads, max_evals_per_restart=max_evals)
        return int(evals.sum()), time.perf_counter() - t0

    n1, t1 = run(2)
    per_round = t1 / 2.0
    m = int(max(2, min(200, round(target_seconds / max(per_round, 1e-6)))))
    if m > 2:
        n2, t2 = run(m)
    else:
        n2, t2 = n1, t1
    return {
        "value": n2 / t2,
        "unit": "evals/s",
        "cores": threads,
        "kind": "port",
        "sample": "{} restarts x {} evaluate_mean calls each of the same CA run ({} evals in {:.1f} s); "
                  "oracle/fastrank_oracle.c, per-call query regrouping hoisted".format(threads, m, n2, t2),
        "evals_per_s_per_core": n2 / t2 / threads,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--measure", default="ndcg@10", help="headline = ndcg@10 (BASELINE.json); others (ndcg, map, mrr, ndcg@k) for side measurements")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--shape", default=os.environ.get("FR_BENCH_SHAPE", "30k"), choices=sorted(SHAPES))
    ap.add_argument("--restarts-per-gpu", type=int, default=32)
    ap.add_argument("--cpu-seconds", type=float, default=float(os.environ.get("FR_BENCH_CPU_SECONDS", "20")))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default=os.environ.get("FR_BENCH_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (default); gloo only for single-GPU smoke tests of the N>1 path")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # FR_BENCH_DEVICE pins every rank to one ordinal (2-rank smoke test of the N>1 path on a 1-GPU box)
    dev_ordinal = int(os.environ.get("FR_BENCH_DEVICE", local_rank))
    torch.cuda.set_device(dev_ordinal)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev_ordinal))
        else:
            dist.init_process_group(backend="gloo")
    coll_dev = torch.device("cuda", dev_ordinal) if args.backend == "nccl" else torch.device("cpu")
    assert world == args.gpus, "launch with torch.distributed.run --nproc-per-node {}".format(args.gpus)

    import fastrank_amd as fr
    from fastrank_amd import native

    native.set_device(dev_ordinal)
    n, d, q, seed = SHAPES[args.shape]
    t0 = time.perf_counter()
    X, y, qid = gen_mslr_shaped(seed, n, d, q)
    gen_s = time.perf_counter() - t0
    dataset = fr.CDataset.from_numpy(X, y, qid)

    R = args.restarts_per_gpu * world
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = args.measure
    p = req.params
    p.num_restarts, p.num_max_iterations, p.step_base, p.step_scale = R, 25, 0.05, 2.0
    p.tolerance, p.normalize, p.init_random, p.seed, p.quiet = 0.001, True, True, 42, True
    begin, end = native.shard_bounds(R, rank, world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t0 = time.perf_counter()
    run = native.CoordinateAscentRun(dataset, req, begin, end)  # uploads + initial evaluate_mean per restart
    torch.cuda.synchronize()
    upload_s = time.perf_counter() - t0

    totals = {"useful_evals": 0, "raw_evals": 0}  # of the jobs already finished (see advance)
    all_restarts = []

    def advance(nsteps):
        """Runs exactly nsteps ticks; when every restart of the current job has converged a new job
        (next seed) is started so that long --steps requests still time real training work."""
        nonlocal run
        done = 0
        while done < nsteps:
            done += run.step(nsteps - done)
            if run.finished and done < nsteps:
                after = run.state()
                totals["useful_evals"] += after["stats"]["useful_evals"]
                totals["raw_evals"] += after["stats"]["raw_evals"]
                all_restarts.extend(after["restarts"])
                p.seed += 1
                run.close()
                run = native.CoordinateAscentRun(dataset, req, begin, end)

    def snapshot():
        """Evaluations so far (finished jobs + the current one); reads the trainer's state, so it is called outside
        the timed region only."""
        st = run.state()["stats"]
        return {k: totals[k] + st[k] for k in totals}

    advance(args.warmup)
    s0 = snapshot()
    native.profile_reset()
    native.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    advance(args.steps)
    barrier()
    elapsed = time.perf_counter() - t0
    native.profile_enable(False)
    s1 = snapshot()
    prof = native.profile_stats()
    # The timed steps keep several launches of the dominant kernel in flight (the restarts are stepped as three sets
    # on three streams), so their HIP-event durations overlap.  A few more steps in plain lock step (one launch per
    # step, nothing else on the device) give the duration of an isolated launch; they are not part of `value`.
    # (tools/pmc_bench.sh counts the instructions and HBM bytes of both kinds of launches of this same command.)
    iso = None
    ISO_STEPS = 4
    if not os.environ.get("FR_LS_PIPELINE"):
        os.environ["FR_LS_PIPELINE"] = "0"
        try:
            native.profile_reset()
            native.profile_enable(True)
            advance(ISO_STEPS)
            torch.cuda.synchronize()
            native.profile_enable(False)
            iso = native.profile_stats()
        finally:
            del os.environ["FR_LS_PIPELINE"]

    useful = s1["useful_evals"] - s0["useful_evals"]
    raw = s1["raw_evals"] - s0["raw_evals"]
    tvals = torch.tensor([elapsed, float(useful), float(raw)], dtype=torch.float64, device=coll_dev)
    if world > 1:
        tmax = tvals.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = tvals.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed_max, useful_all, raw_all = float(tmax[0]), float(tsum[1]), float(tsum[2])
    else:
        elapsed_max, useful_all, raw_all = elapsed, float(useful), float(raw)

    # the job's single exchange: all-gather (restart, score, weights) + deterministic selection
    t0 = time.perf_counter()
    st = run.state()
    mine = st["restarts"]
    if world > 1:
        dim = d
        buf = torch.zeros((args.restarts_per_gpu + 1, 3 + dim), dtype=torch.float64, device=coll_dev)
        for k, r in enumerate(mine):
            buf[k, 0], buf[k, 1], buf[k, 2] = 1.0, float(r["restart_id"]), r["score"]
            buf[k, 3:3 + len(r["weights"])] = torch.tensor(r["weights"], dtype=torch.float64)
        gathered = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(gathered, buf)
        torch.cuda.synchronize()
        allr = [{"restart_id": int(row[1]), "score": row[2], "weights": row[3:]}
                for t in gathered for row in t.cpu().tolist() if row[0] == 1.0]
    else:
        allr = mine
    allr.sort(key=lambda r: r["restart_id"])
    best_model = native.select_model(allr, False)
    collective_ms = (time.perf_counter() - t0) * 1e3
    best_score = max(r["score"] for r in allr)

    if rank == 0:
        b_eval = n * (4 * d + 8)  # algorithmic bytes per evaluate_mean
        # dominant kernel: the bound-and-verify line search (the exact kernel only recomputes the pairs it
        # could not verify); FR_LS_EXACT=1 runs measure the exact kernel instead
        dom = next((k for k in ("linesearch_verify_kernel", "rr_verify_kernel", "rank_metric_kernel") if k in prof),
                   "linesearch_ndcg_kernel")
        ls = prof.get(dom, {"launches": 0, "total_ms": 0.0, "avg_ms": 0.0})
        exact = prof.get("linesearch_ndcg_kernel", {"launches": 0, "total_ms": 0.0, "avg_ms": 0.0})
        evals_per_launch = (raw / max(1, ls["launches"])) if ls["launches"] else 0.0
        avg_s = ls["avg_ms"] * 1e-3
        achieved = (b_eval * evals_per_launch / avg_s / 1e9) if avg_s > 0 else 0.0
        # PMC figures (profiles/hbm_traffic.json, collected by tools/pmc_bench.sh on THIS command): per line group of
        # a launch, separately for the timed (pipelined, ~restarts/3 groups per launch) and the isolated launches
        # -- the instruction count falls as training proceeds (fewer documents enter a top-k list once the model
        # ranks the relevant ones first), so each kind of launch is priced with its own count.
        traffic = None
        valu_insts = None
        iso_valu_insts = None
        pmc_busy_frac = pmc_clock = None
        groups_per_launch = evals_per_launch / 51.0
        tpath = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath)).get(args.shape, {})
                pmc_busy_frac = tj.get("bench_timed_valu_issue_frac_of_busy_cycles")
                pmc_clock = tj.get("bench_timed_clock_ghz")
                if dom == "linesearch_verify_kernel" and "bench_timed_valu_insts_per_group" in tj:
                    valu_insts = tj["bench_timed_valu_insts_per_group"] * groups_per_launch
                    iso_valu_insts = tj["bench_isolated_valu_insts_per_group"] * args.restarts_per_gpu
                    traffic = tj["bench_timed_bytes_per_group"] * groups_per_launch
                else:
                    scale = groups_per_launch / float(tj.get("pmc_groups_per_launch", 32))
                    traffic = tj.get(dom + "_bytes_per_launch")
                    traffic = traffic * scale if traffic else None
                    valu_insts = tj.get(dom + "_valu_insts_per_launch")
                    valu_insts = valu_insts * scale if valu_insts else None
                    iso_valu_insts = valu_insts / groups_per_launch * args.restarts_per_gpu if valu_insts else None
            except Exception:
                traffic = valu_insts = iso_valu_insts = None
        vstats = st.get("stats", {})
        vp, vr = float(vstats.get("verify_pairs", 0)), float(vstats.get("verify_redone", 0))
        # FP64 adds the exact ordered dot products would need (what the exact kernel is bound by)
        adds_per_launch = n * args.restarts_per_gpu * 51 * (d - 1) / 2.0  # avg shared prefix = half the features
        out = {
            "metric": "coordinate-ascent NDCG@10 evals/sec on MSLR-WEB30K shape" if args.measure == "ndcg@10"
            else "coordinate-ascent {} evals/sec on MSLR-WEB30K shape (side measurement)".format(args.measure),
            "value": useful_all / elapsed_max,
            "unit": "evals/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max * 1e3 / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "synthetic MSLR-WEB{} shape: {} docs x {} features x {} queries, coordinate ascent "
                            "{}, {} restarts/GPU x 25 steps/coord (configs[2])".format(
                                args.shape.upper(), n, d, q, args.measure.upper() if args.measure.startswith("ndcg") else args.measure, args.restarts_per_gpu),
                "restarts_total": R,
                "parallelism": "restart-sharded x{} (dataset replicated)".format(world),
                "evals_per_step_per_gpu": raw / max(1, args.steps),
                "launches_per_step": ls["launches"] / max(1, args.steps),
                "groups_per_launch": groups_per_launch,
            },
            "raw_evals_per_s": raw_all / elapsed_max,
            "useful_fraction": useful_all / max(1.0, raw_all),
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": dom,
                "avg_launch_ms": ls["avg_ms"],
                "launches": ls["launches"],
                "algorithmic_bytes_per_launch": b_eval * evals_per_launch,
                "note": "batched: one pass over X serves every candidate of a launch, so the algorithmic "
                        "(per-eval) bytes exceed real HBM traffic and frac can exceed 1; see limiter.  Launches of "
                        "the timed region overlap (three streams): avg_launch_ms includes time shared with the "
                        "neighbouring launches",
            },
            "limiter": ({
                "bound": "valu_issue",
                # wave-level VALU instructions of the dominant kernel (rocprofv3 SQ_INSTS_VALU per (run, group) pair,
                # profiles/hbm_traffic.json, first ticks of a run) issued over the WHOLE timed region -- launch gaps,
                # the small kernels and the host's share included -- against 1024 SIMDs x one VALU instruction per
                # 4 cycles at 2.4 GHz.  The timed launches overlap at their ends (three streams), so their own
                # HIP-event durations (per_launch_overlapped) add up to more than the elapsed time; isolated_launch
                # is the same kernel alone on the device.
                "achieved": (valu_insts * ls["launches"] / elapsed_max / 1e9) if valu_insts else None,
                "peak": 1024 * 2.4e9 / 4 / 1e9,
                "unit": "G wave-instr/s",
                "frac": (valu_insts * ls["launches"] / elapsed_max / (1024 * 2.4e9 / 4)) if valu_insts else None,
                # from the PMC pass (static, profiles/hbm_traffic.json): the kernel's VALU instructions x 4 cycles over the
                # SIMD cycles it was actually busy for (SQ_BUSY_CYCLES), and the clock those cycles imply -- the chip
                # runs this FP64 load at ~1.9 GHz, so the 2.4 GHz peak above is not reachable by any instruction mix
                "valu_issue_frac_of_busy_cycles": pmc_busy_frac,
                "clock_ghz_under_load": pmc_clock,
                "per_launch_overlapped": {
                    "avg_launch_ms": ls["avg_ms"],
                    "groups_per_launch": groups_per_launch,
                    "frac": (valu_insts / avg_s / (1024 * 2.4e9 / 4)) if (valu_insts and avg_s > 0) else None,
                },
                "isolated_launch": ({
                    "avg_launch_ms": iso[dom]["avg_ms"],
                    "launches": iso[dom]["launches"],
                    "groups_per_launch": args.restarts_per_gpu,
                    "frac": (iso_valu_insts / (iso[dom]["avg_ms"] * 1e-3) / (1024 * 2.4e9 / 4))
                    if (iso_valu_insts and iso[dom]["avg_ms"] > 0) else None,
                    "note": "lock-step steps after the timed region (FR_LS_PIPELINE=0): one launch per step, no overlap",
                } if (iso and dom in iso) else None),
                "note": "bound-and-verify kernel on resident sums: three operations per document and restart for the "
                        "base dot product, then a per-document loop over the tile in candidate lanes (LDS broadcast, "
                        "FMA, compare; min/max chain for documents that enter a list); VALU-issue bound, "
                        "see DESIGN.md section 4",
                "verify_pairs": vp,
                "verify_redone": vr,
                "redo_fraction": (vr / vp) if vp else None,
                "exact_kernel_ms_per_step": exact["total_ms"] / max(1, args.steps),
            } if dom == "linesearch_verify_kernel" else {
                "bound": "fp64_valu_add",
                "achieved": (adds_per_launch / avg_s / 1e12) if avg_s > 0 else 0.0,
                "peak": FP64_VALU_PEAK_TADDS,
                "unit": "Tadd/s",
                "frac": (adds_per_launch / avg_s / 1e12 / FP64_VALU_PEAK_TADDS) if avg_s > 0 else 0.0,
                "measured_ceiling": 35.4,  # pure v_add_f64 stream on this chip, tools/ubench/dpadd.hip
            }),
            "kernels_ms": {k: v["total_ms"] for k, v in prof.items()},
            "setup": {"generate_s": gen_s, "upload_and_init_s": upload_s, "final_allgather_select_ms": collective_ms,
                      "best_score_so_far": best_score},
        }
        if world == 1 and not args.no_cpu_baseline:
            del run
            out["cpu_baseline"] = cpu_baseline(X, y, qid, p.to_dict(), args.cpu_seconds, args.measure)
        print(json.dumps(out))
        sys.stdout.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
