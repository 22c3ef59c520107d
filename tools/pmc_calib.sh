#!/bin/bash
# FETCH_SIZE / WRITE_SIZE against known byte counts (tools/ubench/pmc_calib.hip): one --pmc pass each, --kernel-trace only.
# Prints, per access pattern, counter (KB) x 1024 / bytes moved.  usage (GPU box, repo root): bash tools/pmc_calib.sh gpurun_out/pmc_calib
set -u
OUT=${1:-gpurun_out/pmc_calib}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
BIN=tools/ubench/pmc_calib
[ -x "$BIN" ] || hipcc -O2 --offload-arch=gfx950 -o "$BIN" tools/ubench/pmc_calib.hip || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -o p -- "$BIN" > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -o p -- "$BIN" > "$OUT/write.log" 2>&1
python - "$OUT" <<'PY' | tee "$OUT/calibration.txt"
import csv, glob, os, sys, collections
out = sys.argv[1]
BYTES = float(1 << 30)
print("# counter (reported in KB) x 1024 / bytes the kernel moved (1 GiB each; tools/ubench/pmc_calib.hip)")
print("%-64s %-10s %10s %8s" % ("kernel", "counter", "KB", "ratio"))
for tag in ("fetch", "write"):
    agg = collections.OrderedDict()
    for f in sorted(glob.glob(os.path.join(out, tag, "**", "*counter_collection.csv"), recursive=True)):
        for r in csv.DictReader(open(f)):
            if "calib_" not in r["Kernel_Name"]:
                continue
            k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"], r["Dispatch_Id"])
            agg[k] = agg.get(k, 0.0) + float(r["Counter_Value"])
    for (name, ctr, _), v in agg.items():
        print("%-64s %-10s %10.0f %8.3f" % (name[:64], ctr, v, v * 1024.0 / BYTES))
PY
