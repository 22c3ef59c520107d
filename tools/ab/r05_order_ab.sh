#!/bin/bash
# round 5: the verify kernel with and without the visiting-order tables (FR_VERIFY_ORDER=0), lock step, by the kernel's own
# duration and its instruction counters.  usage: tools/ab/r05_order_ab.sh <tag> [kinds...]   (extra env is passed through)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; TAG=${1:-r05ab}; shift; O=gpurun_out/$TAG; mkdir -p $O
KINDS=${@:-"mslr hard"}
one() {  # $1 = label, $2 = data kind, rest = env
  local lab=$1 kind=$2; shift 2
  rm -rf $O/kt_$lab
  env "$@" FR_LS_PIPELINE=0 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d $O/kt_$lab -o b -- python bench.py --steps 40 --warmup 5 --data $kind --no-cpu-baseline --no-e2e --repeats 0 > /dev/null 2>&1
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
docs=3.8e6*32
print("%-22s launches %d avg ms %.4f VALU/doc-group %.3f LDS insts/doc-group %.3f bank-conflict cycles/LDS inst %.3f" % (lab, len(v), avg("d"), avg("SQ_INSTS_VALU")/docs, avg("SQ_INSTS_LDS")/docs, avg("SQ_LDS_BANK_CONFLICT")/max(1,avg("SQ_INSTS_LDS"))))
PY
}
for k in $KINDS; do
  one ${k}_off $k FR_VERIFY_ORDER=0
  one ${k}_on $k FR_VERIFY_ORDER=1
done
