#!/bin/bash
# A/B of two kernels_verify.inc variants by the kernel's own duration over 80 lock-step launches (rocprofv3 kernel trace) and by
# its VALU instruction count: B = the tree's variant, A = tools/ab/kernels_verify_A.inc (git-ignored: put the other variant there, e.g. `git show HEAD~1:fastrank_amd/csrc/kernels_verify.inc`)
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/r04d
one() {
  rm -rf gpurun_out/r04d/kt_$1
  FR_LS_PIPELINE=0 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d gpurun_out/r04d/kt_$1 -o b -- python bench.py --steps 80 --warmup 5 --no-cpu-baseline --no-e2e > /dev/null 2>&1
  python - "$1" <<'PY'
import csv,glob,sys
tag=sys.argv[1]
f=glob.glob("gpurun_out/r04d/kt_%s/**/*counter_collection.csv"%tag, recursive=True)[0]
d=[];v=[]
for r in csv.DictReader(open(f)):
    if "linesearch_verify_kernel" in r["Kernel_Name"]:
        d.append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6); v.append(float(r["Counter_Value"]))
d=d[5:85]; v=v[5:85]
print(tag, "launches", len(d), "avg ms %.4f"%(sum(d)/len(d)), "median %.4f"%sorted(d)[len(d)//2], "VALU insts/launch %.4g"%(sum(v)/len(v)))
PY
}
one B
cp fastrank_amd/csrc/kernels_verify.inc /tmp/B.inc; cp tools/ab/kernels_verify_A.inc fastrank_amd/csrc/kernels_verify.inc
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
one A
cp /tmp/B.inc fastrank_amd/csrc/kernels_verify.inc; python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1
one B2
