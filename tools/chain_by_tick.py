#!/usr/bin/env python3
"""Chain runs per (document, group) visit of the NDCG@k verify kernel, tick by tick, at the start of a job (the kernel's own
counter; run once with FR_VERIFY_ORDER=0 in the environment for storage order).  usage: python tools/chain_by_tick.py [kind] [ticks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, fastrank_amd as fr
from fastrank_amd import native
kind = sys.argv[1] if len(sys.argv) > 1 else "mslr"
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 30
n, d, q, seed = bench.SHAPES["30k"]
X, y, qid = bench.gen_mslr_shaped(seed, n, d, q, kind)
g = fr.CDataset.from_numpy(X, y, qid)
req = fr.TrainRequest.coordinate_ascent(); req.measure = "ndcg@10"
p = req.params; p.num_restarts, p.seed, p.quiet = 32, 42, True
run = native.CoordinateAscentRun(g, req)
prev = run.state()["stats"]; rates = []; times = []; redo = []
for t in range(ticks):
    t0 = time.perf_counter(); run.step(1); native.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
    st = run.state()["stats"]
    dv = st["chain_visits"] - prev["chain_visits"]
    rates.append((st["chain_runs"] - prev["chain_runs"]) / dv if dv else float("nan"))
    dp = st["verify_pairs"] - prev["verify_pairs"]
    redo.append((st["verify_redone"] - prev["verify_redone"]) / dp if dp else float("nan")); prev = st
run.close()
print(kind, "order" if not os.environ.get("FR_VERIFY_ORDER") else "storage", "chain runs per visit by tick:", " ".join("%.3f" % r for r in rates))
print(kind, "ms per single-stepped tick:", " ".join("%.2f" % x for x in times))
print(kind, "share of (query, group) pairs redone by the exact kernel:", " ".join("%.4f" % x for x in redo))
