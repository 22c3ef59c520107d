#!/bin/bash
# rocprofv3 PMC passes for the hot kernel (one --pmc set per run; never combined with sys/hip traces).
# usage: tools/pmc_passes.sh <outdir> [lsbench args]
set -u
OUT=$1; shift
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { # name counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- python tools/lsbench.py --reps 2 "${EXTRA[@]}" > "$OUT/$name.log" 2>&1
}
EXTRA=(--calib "$@")
run sq1 SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU
run sq2 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
run sq3 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VMEM SQ_INSTS_BRANCH SQ_LEVEL_WAVES SQ_CYCLES SQ_BUSY_CU_CYCLES
run mem1 GRBM_GUI_ACTIVE FETCH_SIZE
run mem2 GRBM_GUI_ACTIVE WRITE_SIZE
run mem3 TCC_HIT_sum TCC_MISS_sum
python - "$OUT" <<'PY'
import csv, glob, os, sys, collections
out = sys.argv[1]
rows = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(out, "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        rows.setdefault(k, collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for k, cs in rows.items():
        fh.write(k + "\n")
        for c, v in cs.items():
            fh.write("   %-28s n=%d avg=%.6g\n" % (c, len(v), sum(v) / len(v)))
print(open(os.path.join(out, "summary.txt")).read())
PY
