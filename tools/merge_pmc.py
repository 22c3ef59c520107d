#!/usr/bin/env python3
"""Merges the keys tools/pmc_bench.sh captured (summary.json) into profiles/hbm_traffic.json[shape]:
python tools/merge_pmc.py gpurun_out/pmc_bench/summary.json [--shape 30k]"""
import argparse
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("summary")
    ap.add_argument("--shape", default="30k")
    args = ap.parse_args()
    keys = json.load(open(args.summary))["hbm_traffic_keys"]
    path = os.path.join(ROOT, "profiles", "hbm_traffic.json")
    tj = json.load(open(path))
    missing = [k for k, v in keys.items() if v is None]
    tj.setdefault(args.shape, {}).update({k: v for k, v in keys.items() if v is not None})
    json.dump(tj, open(path, "w"), indent=1)
    print("merged {} keys into {} ({} missing: {})".format(len(keys) - len(missing), path, len(missing), missing))


if __name__ == "__main__":
    main()
