#!/bin/bash
# usage: tools/pmc_kernels.sh <outdir> <kernel-substring> -- <command...>
# rocprofv3 --pmc passes (one counter set per run, only with --kernel-trace) and a summary PER KERNEL INSTANTIATION
# whose name contains the substring: average duration, counters, and the derived VALU-issue / HBM fractions.
set -u
OUT=$1; KSUB=$2; shift 3
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
run() { local name=$1; shift; local ctrs=(); while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
  rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d "$OUT/$name" -o p -- "$@" > "$OUT/$name.log" 2>&1; }
run sq1 SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS -- "$@"
run sq2 SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- "$@"
run sq3 SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LEVEL_WAVES -- "$@"
run sq4 SQ_INSTS_VMEM_WR SQ_INSTS_FLAT -- "$@"   # (scratch spills show up as VMEM writes beyond the kernel's own stores)
run ic SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL -- "$@"   # (fully unrolled kernels: does the code fit the 64 KB instruction cache?)
run mem1 GRBM_GUI_ACTIVE FETCH_SIZE -- "$@"
run mem2 GRBM_GUI_ACTIVE WRITE_SIZE -- "$@"
python - "$OUT" "$KSUB" <<'PY'
import csv, glob, os, re, sys, collections
out, ksub = sys.argv[1], sys.argv[2]
def short(n):
    return re.sub(r"\(.*", "", n.replace("void ", "").replace("frdev::", ""))
ctr = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(out, "*", "*counter_collection.csv"))):
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            ctr.setdefault(short(r["Kernel_Name"]), collections.OrderedDict()).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
dur = collections.OrderedDict()
for f in sorted(glob.glob(os.path.join(out, "mem1", "*kernel_trace.csv")))[:1]:
    for r in csv.DictReader(open(f)):
        if ksub in r["Kernel_Name"]:
            dur.setdefault(short(r["Kernel_Name"]), []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
with open(os.path.join(out, "summary.txt"), "w") as fh:
    for k, cs in ctr.items():
        d = dur.get(k, [])
        ms = sum(d) / len(d) if d else float("nan")
        fh.write("%s   launches=%d avg_ms=%.4f (under the mem1 pass)\n" % (k, len(d), ms))
        avg = {c: sum(v) / len(v) for c, v in cs.items()}
        for c, v in cs.items():
            fh.write("   %-26s n=%d avg=%.6g\n" % (c, len(v), avg[c]))
        if "SQ_INSTS_VALU" in avg and d:
            # 1024 SIMDs, 4 cycles per wave-instruction, 2.4 GHz peak clock
            fh.write("   derived: VALU issue fraction of peak (insts*4 / (1024 SIMD * 2.4 GHz * t)) = %.3f\n" % (avg["SQ_INSTS_VALU"] * 4 / (1024 * 2.4e9 * ms * 1e-3)))
        if "SQ_ACTIVE_INST_VALU" in avg and "SQ_BUSY_CYCLES" in avg:
            fh.write("   derived: VALU active share of the busy cycles (ACTIVE_INST_VALU*4 / (BUSY_CYCLES summed over 32 shader engines -> per SE x 32 SIMDs)) = %.3f\n" % (avg["SQ_ACTIVE_INST_VALU"] * 4 / avg["SQ_BUSY_CYCLES"] / 32))
            fh.write("   derived: shader clock while busy = %.3f GHz (BUSY_CYCLES / 32 SEs / duration)\n" % (avg["SQ_BUSY_CYCLES"] / 32 / (ms * 1e-3) / 1e9))
        if "SQC_ICACHE_REQ" in avg and avg["SQC_ICACHE_REQ"] > 0:
            fh.write("   derived: instruction cache miss rate = %.4f (misses %.4g + duplicate %.4g of %.4g requests)\n" % ((avg.get("SQC_ICACHE_MISSES", 0) + avg.get("SQC_ICACHE_MISSES_DUPLICATE", 0)) / avg["SQC_ICACHE_REQ"], avg.get("SQC_ICACHE_MISSES", 0), avg.get("SQC_ICACHE_MISSES_DUPLICATE", 0), avg["SQC_ICACHE_REQ"]))
        if "FETCH_SIZE" in avg and "WRITE_SIZE" in avg and d:
            b = (avg["FETCH_SIZE"] * 2 + avg["WRITE_SIZE"]) * 1024  # KB units; x2 gfx950 wide-read correction on FETCH_SIZE
            fh.write("   derived: HBM bytes/launch = %.4g (FETCH x2 + WRITE) -> %.3f TB/s = %.3f of 8 TB/s\n" % (b, b / (ms * 1e-3) / 1e12, b / (ms * 1e-3) / 8e12))
print(open(os.path.join(out, "summary.txt")).read())
PY
