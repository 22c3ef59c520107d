#!/bin/bash
# round 5: where the wave-instructions of config 5's kernel go, per NODE STEP (one tree level of one
# document): the whole kernel, without the walks (FR_TREE_NOWALK: staging the threshold ranks + streaming the forest), without
# the staging (FR_TREE_NOSTAGE: garbage codes, same walks).  One counter pass each, kernel trace only.
# (round 6: the tuning / ablation switches this script sets exist only in a pricing build -- csrc/device.hpp pricing_env)
export FR_BUILD_FLAGS="${FR_BUILD_FLAGS:--DFR_PRICING}"; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; O=gpurun_out/${1:-r05tree}; mkdir -p $O
one() {
  local lab=$1; shift
  rm -rf $O/$lab
  env "$@" rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O/$lab -o p -- python bench.py --measure trees --steps 5 --warmup 1 > $O/$lab.log 2>&1
  python - "$lab" "$O" <<'PY'
import csv,glob,sys,collections
lab,O=sys.argv[1],sys.argv[2]
f=glob.glob("%s/%s/**/*counter_collection.csv"%(O,lab), recursive=True)[0]
rows=collections.defaultdict(dict)
for r in csv.DictReader(open(f)):
    if "tree_ensemble" in r["Kernel_Name"]:
        k=r["Dispatch_Id"]; rows[k]["d"]=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6; rows[k][r["Counter_Name"]]=float(r["Counter_Value"]); rows[k]["n"]=r["Kernel_Name"].split("(")[0]
v=list(rows.values())[1:]
avg=lambda k: sum(x.get(k,0) for x in v)/max(1,len(v))
steps=3.8e6*500*7/64.0   # wave-level node steps of one pass (500 trees, 7 levels below the root level)
busy=avg("SQ_BUSY_CYCLES")
print("%-8s %s launches %d ms %.3f | per wave node step: VALU %.2f LDS %.2f SALU %.2f VMEM_RD %.3f | VALU active %.3f LDS idx active %.3f" % (
  lab, v[0]["n"][-40:], len(v), avg("d"), avg("SQ_INSTS_VALU")/steps, avg("SQ_INSTS_LDS")/steps, avg("SQ_INSTS_SALU")/steps, avg("SQ_INSTS_VMEM_RD")/steps,
  avg("SQ_ACTIVE_INST_VALU")*4/busy/32, avg("SQ_LDS_IDX_ACTIVE")/busy/8))
PY
}
one full
one nowalk FR_TREE_NOWALK=1
one nostage FR_TREE_NOSTAGE=1
