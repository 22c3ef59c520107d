#!/usr/bin/env python3
"""End-to-end train_model to convergence on a synthetic MSLR shape (BASELINE.json configs[1]):
wall time, ticks, useful/raw evaluations, final NDCG@10."""
import argparse
import hashlib
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
    ap.add_argument("--shape", default="10k")
    ap.add_argument("--restarts", type=int, default=8)
    ap.add_argument("--measure", default="ndcg@10")
    ap.add_argument("--data", default="mslr", choices=bench.DATA_KINDS)
    ap.add_argument("--max-ticks", type=int, default=100000)
    ap.add_argument("--profile", action="store_true", help="HIP-event time per kernel (use with FR_LS_PIPELINE=0: overlapping launches inflate it)")
    args = ap.parse_args()
    n, d, q, seed = bench.SHAPES[args.shape]
    X, y, qid = bench.gen_mslr_shaped(seed, n, d, q, args.data)
    ds = fr.CDataset.from_numpy(X, y, qid)
    req = fr.TrainRequest.coordinate_ascent()
    req.measure = args.measure
    p = req.params
    p.num_restarts, p.seed, p.quiet = args.restarts, 42, True
    t0 = time.perf_counter()
    run = native.CoordinateAscentRun(ds, req)
    t_init = time.perf_counter() - t0
    if args.profile:
        native.profile_reset()
        native.profile_enable(True)
    t0 = time.perf_counter()
    ticks = 0
    while not run.finished and ticks < args.max_ticks:
        ticks += run.step(min(136, args.max_ticks - ticks))
        st = run.state()
        print("  ticks=%d best=%.6f useful=%d raw=%d elapsed=%.1fs" % (
            ticks, max(r["score"] for r in st["restarts"]), st["stats"]["useful_evals"], st["stats"]["raw_evals"],
            time.perf_counter() - t0), flush=True)
    wall = time.perf_counter() - t0
    if args.profile:
        native.profile_enable(False)
        for name, v in sorted(native.profile_stats().items()):
            print("   %-30s launches=%d avg=%.3f ms total=%.1f ms" % (name, v["launches"], v["avg_ms"], v["total_ms"]))
    st = run.state()
    model = native.select_model(st["restarts"], False)
    mean = float(np.mean(native.evaluate_dense(model, ds, args.measure)[1]))
    vp, vr = st["stats"].get("verify_pairs") or 0, st["stats"].get("verify_redone") or 0
    print(json.dumps({"shape": args.shape, "data": args.data, "redo_fraction": (vr / vp) if vp else None,
                      "exact_line_search_share": (st["stats"].get("exact_ticks") or 0) / max(1, st["stats"].get("line_searches") or 1), "restarts": args.restarts, "measure": args.measure, "finished": run.finished,
                      "ticks": ticks, "train_wall_s": wall, "upload_init_s": t_init,
                      "useful_evals": st["stats"]["useful_evals"], "raw_evals": st["stats"]["raw_evals"],
                      "useful_evals_per_s": st["stats"]["useful_evals"] / wall, "final_mean": mean,
                      "path": st["stats"]["path"], "verify_pairs": st["stats"].get("verify_pairs"),
                      "verify_redone": st["stats"].get("verify_redone"),
                      "audit_values": st["stats"].get("audit_values"), "audit_mismatches": st["stats"].get("audit_mismatches"),
                      "restarts_sha1": hashlib.sha1(json.dumps(st["restarts"], sort_keys=True).encode()).hexdigest()}))


if __name__ == "__main__":
    main()
