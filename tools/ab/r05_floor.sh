#!/bin/bash
# round 5: what the verify kernel costs WITHOUT its per-document loop (FR_LS_DEBUG=1: phase S, the tile steps, the queries' ends,
# the prologue; nothing is listed for redo), next to the whole kernel -- lock step
# (round 6: the tuning / ablation switches this script sets exist only in a pricing build -- csrc/device.hpp pricing_env)
export FR_BUILD_FLAGS="${FR_BUILD_FLAGS:--DFR_PRICING}"; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; TAG=${1:-r05floor}; O=gpurun_out/$TAG; mkdir -p $O
one() {
  local lab=$1; shift
  rm -rf $O/kt_$lab
  env "$@" FR_LS_PIPELINE=0 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $O/kt_$lab -o b -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-e2e --repeats 0 > /dev/null 2>&1
  python - "$lab" "$O" <<'PY'
import csv,glob,sys,collections
lab,O=sys.argv[1],sys.argv[2]
f=glob.glob("%s/kt_%s/**/*counter_collection.csv"%(O,lab), recursive=True)[0]
rows=collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "linesearch_verify_kernel" in r["Kernel_Name"]:
        k=r["Dispatch_Id"]; rows[k]["d"]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6; rows[k][r["Counter_Name"]]=float(r["Counter_Value"])
v=list(rows.values())[3:23]
avg=lambda k: sum(x.get(k,0) for x in v)/max(1,len(v))
busy=avg("SQ_BUSY_CYCLES"); docs=3.8e6*32
print("%-10s launches %d ms %.4f per doc-group: VALU %.3f SALU %.3f VMEM_RD %.3f LDS %.3f | VALU active %.3f" % (lab, len(v), avg("d"), avg("SQ_INSTS_VALU")/docs, avg("SQ_INSTS_SALU")/docs, avg("SQ_INSTS_VMEM_RD")/docs, avg("SQ_INSTS_LDS")/docs, avg("SQ_ACTIVE_INST_VALU")*4/busy/32))
PY
}
one full
one nok FR_LS_DEBUG=1
one full_off FR_VERIFY_ORDER=0
one nok_off FR_LS_DEBUG=1 FR_VERIFY_ORDER=0
