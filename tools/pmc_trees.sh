#!/bin/bash
# PMC passes over `python bench.py --measure trees` (BASELINE.json configs[4]; rocprofv3 --pmc with --kernel-trace only, one
# counter set per pass): HBM bytes, VALU / LDS instructions, VALU-active and LDS-address-unit cycles of the tree-scoring
# kernel per pass.  Writes $OUT/summary.json and merges it into profiles/hbm_traffic.json[30k]["trees"] together with the
# SHA-1 of the sources it was captured on (bench.py marks the figures `stale` when they differ).
# usage (GPU box, repo root): bash tools/pmc_trees.sh gpurun_out/pmc_trees r04
set -u
OUT=${1:-gpurun_out/pmc_trees}
ROUND=${2:-r04}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
CMD=(python bench.py --measure trees --steps 5 --warmup 1)
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d "$OUT/sq1" -o p -- "${CMD[@]}" > "$OUT/sq1.log" 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d "$OUT/sq2" -o p -- "${CMD[@]}" > "$OUT/sq2.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o p -- "${CMD[@]}" > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o p -- "${CMD[@]}" > "$OUT/write.log" 2>&1
python - "$OUT" "$ROUND" <<'PY'
import collections, csv, glob, json, os, sys
out, rnd = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.getcwd())
import bench
TAGS = ("sq1", "sq2", "fetch", "write")
per = collections.defaultdict(lambda: collections.defaultdict(dict))  # tag -> dispatch -> counter -> value
for f in sorted(glob.glob(os.path.join(out, "*", "*counter_collection.csv")) + glob.glob(os.path.join(out, "*", "*", "*counter_collection.csv"))):
    parts = f.split(os.sep)
    tag = parts[-2] if parts[-2] in TAGS else parts[-3]
    for r in csv.DictReader(open(f)):
        if "tree_ensemble" not in r["Kernel_Name"]:
            continue
        dsp = per[tag][int(r["Dispatch_Id"])]
        dsp[r["Counter_Name"]] = dsp.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        dsp["_dur_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        dsp["_name"] = r["Kernel_Name"]
def avg(tag, ctr):
    # (the largest launches of the kernel = whole passes over the matrix; the 20 000-document parity check is a small one)
    rows = list(per[tag].values())
    if not rows:
        return None
    big = max(r["_dur_ns"] for r in rows)
    rows = [r for r in rows if r["_dur_ns"] > 0.5 * big and ctr in r]
    return sum(r[ctr] for r in rows) / len(rows) if rows else None
def dur(tag):
    rows = list(per[tag].values())
    if not rows:
        return None
    big = max(r["_dur_ns"] for r in rows)
    rows = [r for r in rows if r["_dur_ns"] > 0.5 * big]
    return sum(r["_dur_ns"] for r in rows) / len(rows) * 1e-9
f, w = avg("fetch", "FETCH_SIZE"), avg("write", "WRITE_SIZE")
busy1, busy2 = avg("sq1", "SQ_BUSY_CYCLES"), avg("sq2", "SQ_BUSY_CYCLES")
keys = {
    "kernel": next(iter(per["sq1"].values()))["_name"].split("(")[0] if per["sq1"] else None,
    "bytes_per_pass": (f * 2.0 + w) * 1024.0 if (f is not None and w is not None) else None,  # KB units; x2: gfx950 wide-read correction
    "fetch_size_kb": f, "write_size_kb": w,
    "valu_insts_per_pass": avg("sq1", "SQ_INSTS_VALU"), "lds_insts_per_pass": avg("sq1", "SQ_INSTS_LDS"),
    "salu_insts_per_pass": avg("sq1", "SQ_INSTS_SALU"), "waves_per_pass": avg("sq1", "SQ_WAVES"),
    "launch_ms_under_pmc": dur("sq1") * 1e3 if dur("sq1") else None,
    "clock_ghz": busy1 / 32.0 / dur("sq1") / 1e9 if busy1 and dur("sq1") else None,
    "valu_active_frac_of_busy_cycles": avg("sq1", "SQ_ACTIVE_INST_VALU") * 4.0 / (1024.0 * busy1 / 32.0) if busy1 and avg("sq1", "SQ_ACTIVE_INST_VALU") else None,
    "lds_idx_active_frac_of_busy_cycles": avg("sq2", "SQ_LDS_IDX_ACTIVE") / (256.0 * busy2 / 32.0) if busy2 and avg("sq2", "SQ_LDS_IDX_ACTIVE") else None,
    "lds_bank_conflict_frac_of_lds_active": avg("sq2", "SQ_LDS_BANK_CONFLICT") / avg("sq2", "SQ_LDS_IDX_ACTIVE") if avg("sq2", "SQ_LDS_IDX_ACTIVE") else None,
    "captured": {"round": rnd, "sha1": bench.source_sha1(bench.TREE_PMC_SOURCES), "command": "python bench.py --measure trees --steps 5 --warmup 1"},
}
json.dump(keys, open(os.path.join(out, "summary.json"), "w"), indent=1)
path = os.path.join("profiles", "hbm_traffic.json")
tj = json.load(open(path))
tj.setdefault("30k", {})["trees"] = keys
json.dump(tj, open(path, "w"), indent=1)
json.dump(tj, open(os.path.join(out, "hbm_traffic.json"), "w"), indent=1)
print(json.dumps(keys, indent=1))
PY
