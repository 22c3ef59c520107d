#!/bin/bash
# PMC passes over the bench command itself (rocprofv3 --pmc with --kernel-trace only; one counter set per pass):
# wave-level VALU instructions and HBM bytes of the dominant kernel's launches, per line group, separately for
# the timed (pipelined) launches and for the isolated lock-step launches bench.py appends.  Prints the keys
# bench.py reads from profiles/hbm_traffic.json.
# usage (GPU box, repo root): bash tools/pmc_bench.sh gpurun_out/pmc_bench
set -u
OUT=${1:-gpurun_out/pmc_bench}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
CMD=(python bench.py --steps 20 --warmup 3 --no-cpu-baseline)
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d "$OUT/sq" -o p -- "${CMD[@]}" > "$OUT/sq.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o p -- "${CMD[@]}" > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o p -- "${CMD[@]}" > "$OUT/write.log" 2>&1
python - "$OUT" <<'PY'
import csv, glob, json, os, sys, collections
out = sys.argv[1]
WARM_LAUNCHES = 9   # 3 warm-up ticks x 3 sets
per = collections.defaultdict(dict)   # (pass, dispatch) -> counter -> value
grid = {}
for f in sorted(glob.glob(os.path.join(out, "*", "*counter_collection.csv")) + glob.glob(os.path.join(out, "*", "*", "*counter_collection.csv"))):
    tag = f.split(os.sep)[-2] if f.split(os.sep)[-2] in ("sq", "fetch", "write") else f.split(os.sep)[-3]
    for r in csv.DictReader(open(f)):
        if "linesearch_verify_kernel" not in r["Kernel_Name"]:
            continue
        k = (tag, int(r["Dispatch_Id"]))
        per[k][r["Counter_Name"]] = per[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        per[k]["_dur_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        grid[k] = int(r["Grid_Size"])
res = {}
for tag in ("sq", "fetch", "write"):
    ks = sorted(k for k in per if k[0] == tag)
    if not ks:
        continue
    gmax = max(grid[k] for k in ks)
    iso = [k for k in ks if grid[k] == gmax]
    timed = [k for k in ks if grid[k] != gmax][WARM_LAUNCHES:]
    # groups of a launch: grid = runs_padded * groups * 64 threads; the isolated launches carry 32
    unit = gmax / 32.0
    for name, sel in (("timed", timed), ("isolated", iso)):
        groups = sum(grid[k] / unit for k in sel)
        for c in per[ks[0]]:
            if c.startswith("_"):
                continue
            res["%s_%s_per_group" % (name, c)] = sum(per[k].get(c, 0.0) for k in sel) / max(groups, 1e-9)
        res["%s_launches_%s" % (name, tag)] = len(sel)
        if tag == "sq" and sel:
            # SQ_BUSY_CYCLES is summed over the 32 shader engines: busy / 32 / duration = the clock the kernel ran at
            busy = sum(per[k].get("SQ_BUSY_CYCLES", 0.0) for k in sel)
            dur = sum(per[k]["_dur_ns"] for k in sel) * 1e-9
            valu = sum(per[k].get("SQ_INSTS_VALU", 0.0) for k in sel)
            res["%s_clock_ghz" % name] = busy / 32.0 / dur / 1e9
            res["%s_valu_issue_frac_of_busy_cycles" % name] = valu * 4.0 / (1024.0 * busy / 32.0)
            res["%s_launch_ms_serialised" % name] = dur / len(sel) * 1e3
keys = {}
for name in ("timed", "isolated"):
    keys["bench_%s_valu_insts_per_group" % name] = res.get("%s_SQ_INSTS_VALU_per_group" % name)
    f, w = res.get("%s_FETCH_SIZE_per_group" % name), res.get("%s_WRITE_SIZE_per_group" % name)
    # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports half the bytes of 16-byte-per-lane coalesced reads (x2, see _doc)
    keys["bench_%s_bytes_per_group" % name] = (f * 2.0 + w) * 1024.0 if (f is not None and w is not None) else None
    keys["bench_%s_clock_ghz" % name] = res.get("%s_clock_ghz" % name)
    keys["bench_%s_valu_issue_frac_of_busy_cycles" % name] = res.get("%s_valu_issue_frac_of_busy_cycles" % name)
    keys["bench_%s_fetch_size_kb_per_group" % name] = f
    keys["bench_%s_write_size_kb_per_group" % name] = w
summary = {"raw": res, "hbm_traffic_keys": keys}
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
