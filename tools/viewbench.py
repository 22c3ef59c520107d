#!/usr/bin/env python3
"""Sampled views at the 30K shape: HBM owned by an 80 % query sample, time to its first compute call,
and NDCG@10 training throughput on it -- sharing the parent's tiles (default) vs tiling its own copy (FR_VIEW_COPIES=1)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import fastrank_amd as fr  # noqa: E402
from fastrank_amd import native  # noqa: E402


def main():
    n, d, q, seed = bench.SHAPES["30k"]
    X, y, qid = bench.gen_mslr_shaped(seed, n, d, q)
    parent = fr.CDataset.from_numpy(X, y, qid)
    pinfo = native.device_info(parent)
    names = [str(v) for v in range(1, q + 1)]
    train = [s for i, s in enumerate(names) if i % 5 != 0]
    out = {"parent_hbm_bytes": pinfo["hbm_bytes_owned"], "view": "4 of every 5 queries (%d)" % len(train)}
    for mode in ("shared", "copy"):
        if mode == "copy":
            os.environ["FR_VIEW_COPIES"] = "1"
        t0 = time.perf_counter()
        view = parent.subsample_queries(train)
        t_sample = time.perf_counter() - t0
        t0 = time.perf_counter()
        info = native.device_info(view)
        t_dev = time.perf_counter() - t0
        req = fr.TrainRequest.coordinate_ascent()
        req.measure = "ndcg@10"
        req.params.num_restarts, req.params.quiet, req.params.seed = 32, True, 42
        run = native.CoordinateAscentRun(view, req)
        run.step(5)
        native.synchronize()
        t0 = time.perf_counter()
        run.step(40)
        native.synchronize()
        dt = time.perf_counter() - t0
        st = run.state()["stats"]
        out[mode] = {"hbm_bytes_owned": info["hbm_bytes_owned"], "shares_parent_matrix": info["shares_parent_matrix"],
                     "host_sampling_s": t_sample, "device_form_s": t_dev, "ms_per_tick": dt / 40 * 1e3,
                     "raw_evals_per_s_40_ticks": 40 * 32 * 51 / dt, "path": st["path"]}
        run.close()
        del view
    print(json.dumps(out))


if __name__ == "__main__":
    main()
