#!/bin/bash
# PMC passes over the bench command itself (rocprofv3 --pmc with --kernel-trace only; one counter set per pass):
# wave-level VALU instructions (total and by class), VALU-active cycles and HBM bytes of the dominant kernel's
# launches, per line group, separately for the timed (pipelined) launches and for the isolated lock-step launches
# bench.py appends.  Writes $OUT/summary.json: the keys bench.py reads from profiles/hbm_traffic.json plus the SHA-1
# of the sources they were captured on (merge with `python tools/merge_pmc.py $OUT/summary.json`).
# usage (GPU box, repo root): bash tools/pmc_bench.sh gpurun_out/pmc_bench r02
set -u
OUT=${1:-gpurun_out/pmc_bench}
ROUND=${2:-r02}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
CMD=(python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e)
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d "$OUT/sq" -o p -- "${CMD[@]}" > "$OUT/sq.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_SALU --kernel-trace --output-format csv -d "$OUT/mix" -o p -- "${CMD[@]}" > "$OUT/mix.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o p -- "${CMD[@]}" > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o p -- "${CMD[@]}" > "$OUT/write.log" 2>&1
python - "$OUT" "$ROUND" <<'PY'
import csv, glob, json, os, sys, collections
out, rnd = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.getcwd())
import bench
WARM_LAUNCHES = 9   # 3 warm-up ticks x 3 sets
TAGS = ("sq", "mix", "fetch", "write")
per = collections.defaultdict(dict)   # (pass, dispatch) -> counter -> value
grid = {}
for f in sorted(glob.glob(os.path.join(out, "*", "*counter_collection.csv")) + glob.glob(os.path.join(out, "*", "*", "*counter_collection.csv"))):
    tag = f.split(os.sep)[-2] if f.split(os.sep)[-2] in TAGS else f.split(os.sep)[-3]
    for r in csv.DictReader(open(f)):
        if "linesearch_verify_kernel" not in r["Kernel_Name"]:
            continue
        k = (tag, int(r["Dispatch_Id"]))
        per[k][r["Counter_Name"]] = per[k].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        per[k]["_dur_ns"] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        grid[k] = int(r["Grid_Size"])
res = {}
for tag in TAGS:
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
            act = sum(per[k].get("SQ_ACTIVE_INST_VALU", 0.0) for k in sel)  # quad-cycles, summed over SIMDs
            res["%s_clock_ghz" % name] = busy / 32.0 / dur / 1e9
            res["%s_valu_issue_frac_of_busy_cycles" % name] = valu * 4.0 / (1024.0 * busy / 32.0)
            res["%s_valu_active_frac_of_busy_cycles" % name] = act * 4.0 / (1024.0 * busy / 32.0) if act else None
            res["%s_launch_ms_serialised" % name] = dur / len(sel) * 1e3
keys = {}
for name in ("timed", "isolated"):
    keys["bench_%s_valu_insts_per_group" % name] = res.get("%s_SQ_INSTS_VALU_per_group" % name)
    f, w = res.get("%s_FETCH_SIZE_per_group" % name), res.get("%s_WRITE_SIZE_per_group" % name)
    # FETCH_SIZE / WRITE_SIZE are in KB; gfx950 reports half the bytes of 16-byte-per-lane coalesced reads (x2, see _doc)
    keys["bench_%s_bytes_per_group" % name] = (f * 2.0 + w) * 1024.0 if (f is not None and w is not None) else None
    keys["bench_%s_clock_ghz" % name] = res.get("%s_clock_ghz" % name)
    keys["bench_%s_valu_issue_frac_of_busy_cycles" % name] = res.get("%s_valu_issue_frac_of_busy_cycles" % name)
    keys["bench_%s_valu_active_frac_of_busy_cycles" % name] = res.get("%s_valu_active_frac_of_busy_cycles" % name)
    keys["bench_%s_fetch_size_kb_per_group" % name] = f
    keys["bench_%s_write_size_kb_per_group" % name] = w
    mix = {c[len("SQ_INSTS_"):].lower() + "_per_group": res.get("%s_%s_per_group" % (name, c)) for c in (
        "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64",
        "SQ_INSTS_VALU_INT32", "SQ_INSTS_VALU_INT64", "SQ_INSTS_VALU_CVT", "SQ_INSTS_SALU")}
    if any(v is not None for v in mix.values()):
        # instructions that run at the FP64 rate (4 cycles per wave instruction); the unclassified remainder of
        # SQ_INSTS_VALU (v_min/max_f64, v_cmp_f64, moves ...) is not in these counters: bench.py prices it at 2 cycles
        # for the lower bound and at 4 for the upper one
        mix["f64_class_per_group"] = sum(mix.get(k) or 0.0 for k in ("valu_add_f64_per_group", "valu_mul_f64_per_group",
                                                                    "valu_fma_f64_per_group", "valu_trans_f64_per_group"))
        keys["bench_%s_valu_mix" % name] = mix
keys["captured"] = {"round": rnd, "sha1": bench.source_sha1(), "command": "python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e"}
summary = {"raw": res, "hbm_traffic_keys": keys}
json.dump(summary, open(os.path.join(out, "summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
PY
