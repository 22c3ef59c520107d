#!/bin/bash
# VALU wave-instructions of linesearch_verify_kernel per (document, group) with and without phase K (FR_LS_DEBUG=1 skips the
# per-document loop): how much of the kernel is the per-tile and per-query work around it?
# (round 6: the tuning / ablation switches this script sets exist only in a pricing build -- csrc/device.hpp pricing_env)
export FR_BUILD_FLAGS="${FR_BUILD_FLAGS:--DFR_PRICING}"; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04d
one() {
  rm -rf gpurun_out/r04d/bd_$1
  FR_LS_PIPELINE=0 FR_LS_DEBUG=$2 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD --output-format csv -d gpurun_out/r04d/bd_$1 -o b -- python bench.py --steps $3 --warmup 2 --no-cpu-baseline --no-e2e > /dev/null 2>&1
  python - "$1" <<'PY'
import csv,glob,sys,collections
tag=sys.argv[1]
f=glob.glob("gpurun_out/r04d/bd_%s/**/*counter_collection.csv"%tag, recursive=True)[0]
per=collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "linesearch_verify_kernel" in r["Kernel_Name"]:
        per[int(r["Dispatch_Id"])][r["Counter_Name"]]=float(r["Counter_Value"])
        per[int(r["Dispatch_Id"])]["dur"]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
ks=sorted(per)[(2 if len(per) > 4 else 0):]
n=max(1,len(ks)); dv=3.8e6*32
print(tag, "launches", n, "avg ms %.3f"%(sum(per[k]["dur"] for k in ks)/n), {c: round(sum(per[k].get(c,0) for k in ks)/n/dv,3) for c in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_INSTS_VMEM_RD")}, "per (document, group)")
PY
}
one full 0 20
one nok 1 2
