#!/bin/bash
# round 5: where the verify kernel's cycles go with / without the visiting-order tables: SQ activity counters, lock step
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; TAG=${1:-r05pmc}; shift; O=gpurun_out/$TAG; mkdir -p $O
KINDS=${@:-"mslr"}
one() {
  local lab=$1 kind=$2; shift 2
  rm -rf $O/kt_$lab
  env "$@" FR_LS_PIPELINE=0 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/kt_$lab -o b -- python bench.py --steps 40 --warmup 5 --data $kind --no-cpu-baseline --no-e2e --repeats 0 > /dev/null 2>&1
  python - "$lab" "$O" <<'PY'
import csv,glob,sys,collections
lab,O=sys.argv[1],sys.argv[2]
f=glob.glob("%s/kt_%s/**/*counter_collection.csv"%(O,lab), recursive=True)[0]
rows=collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "linesearch_verify_kernel" in r["Kernel_Name"]:
        k=r["Dispatch_Id"]; rows[k]["d"]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6; rows[k][r["Counter_Name"]]=float(r["Counter_Value"])
v=list(rows.values())[5:45]
avg=lambda k: sum(x.get(k,0) for x in v)/max(1,len(v))
busy=avg("SQ_BUSY_CYCLES")
print("%-12s ms %.4f VALU insts %.4g | of busy cycles (x32 SIMD per SE counted): VALU active %.3f LDS active %.3f LDS idx active %.3f wait-LDS %.3f wait-any %.3f | clock %.3f GHz" % (
  lab, avg("d"), avg("SQ_INSTS_VALU"), avg("SQ_ACTIVE_INST_VALU")*4/busy/32, avg("SQ_ACTIVE_INST_LDS")*4/busy/32, avg("SQ_LDS_IDX_ACTIVE")/busy/8, avg("SQ_WAIT_INST_LDS")/busy/32/8,
  avg("SQ_WAIT_INST_ANY")/busy/32/8, busy/32/(avg("d")*1e-3)/1e9))
PY
}
for k in $KINDS; do
  one ${k}_off $k FR_VERIFY_ORDER=0
  one ${k}_on $k FR_VERIFY_ORDER=1
done
