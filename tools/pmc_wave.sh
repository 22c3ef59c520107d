#!/bin/bash
# usage: tools/pmc_wave.sh <outdir> <kernel-substring> -- <command...>
# Where do a kernel's wave-cycles go?  WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES.
set -u
OUT=$1; KSUB=$2; shift 3
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { local name=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- "$@" > "$OUT/$name.log" 2>&1; }
run w1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM -- "$@"
run w2 SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT -- "$@"
python - "$OUT" "$KSUB" <<'PY'
import csv, glob, os, sys, collections
out, ksub = sys.argv[1], sys.argv[2]
rows = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(out, "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            rows.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for f in sorted(glob.glob(os.path.join(out, "*", "*kernel_trace.csv")))[:1]:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in csv.DictReader(open(f)) if ksub in r["Kernel_Name"]]
    print("kernel ms:", [round(x, 3) for x in d])
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for c, v in rows.items():
        fh.write("%-28s n=%d avg=%.6g\n" % (c, len(v), sum(v) / len(v)))
print(open(os.path.join(out, "summary.txt")).read())
PY
