#!/usr/bin/env python3
"""Milliseconds per single-stepped tick at the start of a job, any measure: python tools/ms_by_tick.py [measure] [kind] [ticks]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, fastrank_amd as fr
from fastrank_amd import native
measure = sys.argv[1] if len(sys.argv) > 1 else "ndcg"
kind = sys.argv[2] if len(sys.argv) > 2 else "mslr"
ticks = int(sys.argv[3]) if len(sys.argv) > 3 else 24
n, d, q, seed = bench.SHAPES["30k"]
X, y, qid = bench.gen_mslr_shaped(seed, n, d, q, kind)
g = fr.CDataset.from_numpy(X, y, qid)
req = fr.TrainRequest.coordinate_ascent(); req.measure = measure
p = req.params; p.num_restarts, p.seed, p.quiet = 32, 42, True
run = native.CoordinateAscentRun(g, req)
times, redo, prev = [], [], run.state()["stats"]
for t in range(ticks):
    t0 = time.perf_counter(); run.step(1); native.synchronize(); times.append((time.perf_counter() - t0) * 1e3)
    st = run.state()["stats"]; dp = st["verify_pairs"] - prev["verify_pairs"]
    redo.append((st["verify_redone"] - prev["verify_redone"]) / dp if dp else float("nan")); prev = st
native.profile_reset(); native.profile_enable(True); run.step(4); native.synchronize(); native.profile_enable(False)
print(measure, kind, "kernels, ms per tick over 4 more ticks:", {k: round(v["total_ms"] / 4, 2) for k, v in native.profile_stats().items()})
print(measure, kind, "path", run.state()["stats"]["path"], "ms per tick:", " ".join("%.1f" % x for x in times))
print(measure, kind, "pairs redone:", " ".join("%.4f" % x for x in redo))
run.close()
