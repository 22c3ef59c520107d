#!/usr/bin/env python3
"""Tick time of the trainer against the number of restarts stepped together (G line groups per tick), at the bench's
shape: the numbers behind train_model's per-device floor (FR_MIN_RESTARTS_PER_DEVICE) and the per-trainer bound on live
restarts (FR_RESTART_SLOTS).  One JSON line: [{"groups", "ms_per_tick", "evals_per_s", "init_s"} ...] and the affine fit
ms_per_tick ~ a + b * groups over the points with at least 4 groups."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import fastrank_amd as fr  # noqa: E402
from fastrank_amd import native  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="30k")
    ap.add_argument("--measure", default="ndcg@10")
    ap.add_argument("--groups", default="1,2,3,4,5,6,8,12,16,24,32,48,64")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--ticks", type=int, default=40)
    args = ap.parse_args()
    n, d, q, seed = bench.SHAPES[args.shape]
    X, y, qid = bench.gen_mslr_shaped(seed, n, d, q)
    ds = fr.CDataset.from_numpy(X, y, qid)
    rows = []
    for G in [int(x) for x in args.groups.split(",")]:
        req = fr.TrainRequest.coordinate_ascent()
        req.measure = args.measure
        p = req.params
        p.num_restarts, p.seed, p.quiet = G, 42, True
        t0 = time.perf_counter()
        run = native.CoordinateAscentRun(ds, req)
        native.synchronize()
        init_s = time.perf_counter() - t0
        run.step(args.warmup)
        s0 = run.state()["stats"]
        native.synchronize()
        t0 = time.perf_counter()
        done = run.step(args.ticks)
        native.synchronize()
        wall = time.perf_counter() - t0
        s1 = run.state()["stats"]
        run.close()
        rows.append({"groups": G, "ms_per_tick": wall * 1e3 / max(1, done), "ticks": done, "init_s": init_s,
                     "evals_per_s": (s1["useful_evals"] - s0["useful_evals"]) / wall})
        print("  G=%d  %.3f ms/tick  %.0f evals/s  init %.3f s" % (G, rows[-1]["ms_per_tick"], rows[-1]["evals_per_s"], init_s),
              file=sys.stderr, flush=True)
    big = [r for r in rows if r["groups"] >= 4]
    fit = None
    if len(big) >= 2:
        b, a = np.polyfit([r["groups"] for r in big], [r["ms_per_tick"] for r in big], 1)
        fit = {"a_ms": float(a), "b_ms_per_group": float(b), "a_over_b_groups": float(a / b) if b > 0 else None}
    print(json.dumps({"shape": args.shape, "measure": args.measure, "pipeline": os.environ.get("FR_LS_PIPELINE", "3 (default)"),
                      "rows": rows, "fit_ge_4_groups": fit}))


if __name__ == "__main__":
    main()
