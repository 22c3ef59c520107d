#!/bin/bash
# round 5: variants of the visiting order by environment (no rebuild): kappa = inf (every lane walks the R copy: one address per
# read), kappa = 0 (no lane does), ranks made once
# (round 6: the tuning / ablation switches this script sets exist only in a pricing build -- csrc/device.hpp pricing_env)
export FR_BUILD_FLAGS="${FR_BUILD_FLAGS:--DFR_PRICING}"; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; TAG=${1:-r05var}; shift; O=gpurun_out/$TAG; mkdir -p $O
KIND=${1:-mslr}
one() {
  local lab=$1; shift
  rm -rf $O/kt_$lab
  env "$@" FR_LS_PIPELINE=0 rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --output-format csv -d $O/kt_$lab -o b -- python bench.py --steps 40 --warmup 5 --data $KIND --no-cpu-baseline --no-e2e --repeats 0 > /dev/null 2>&1
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
busy=avg("SQ_BUSY_CYCLES"); docs=3.8e6*32
print("%-14s ms %.4f VALU/doc-group %.3f VALU active %.3f LDS idx active %.3f conflicts/LDS inst %.3f" % (lab, avg("d"), avg("SQ_INSTS_VALU")/docs, avg("SQ_ACTIVE_INST_VALU")*4/busy/32, avg("SQ_LDS_IDX_ACTIVE")/busy/8, avg("SQ_LDS_BANK_CONFLICT")/max(1,avg("SQ_INSTS_LDS"))))
PY
}
one off FR_VERIFY_ORDER=0
one on
one all_near FR_ORDER_KAPPA=1e30
one none_near FR_ORDER_KAPPA=0
one rank_once FR_RANK_PERIOD=1000000
one kappa4 FR_ORDER_KAPPA=4
one kappa025 FR_ORDER_KAPPA=0.25
