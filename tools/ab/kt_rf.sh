cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt_rf -o p -- python tools/rfbench.py --shape 30k --trees 11 --cpu-trees 0 > gpurun_out/kt_rf.log 2>&1
python - <<'PY'
import csv,glob
f=(glob.glob("gpurun_out/kt_rf/*kernel_trace.csv")+glob.glob("gpurun_out/kt_rf/*/*kernel_trace.csv"))[0]
rows=[r for r in csv.DictReader(open(f))]
for name in ["rf_eval_kernel","rf_blk_build","rf_blk_scan"]:
    d=[((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6, r["Grid_Size_X"]) for r in rows if name in r["Kernel_Name"]]
    print(name, " ".join("%.1f(%s)"%x for x in d))
PY
rm -rf gpurun_out/kt_rf
