set -u
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
A="--workload c3x --steps 1 --warmup 1 --max-warmup 0 --no-cpu-baseline"
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/q_mfma -o m -- python3 $R/bench.py $A > $O/q_m.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $O/q_lds -o l -- python3 $R/bench.py $A > $O/q_l.log 2>&1 < /dev/null
timeout 600 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU --output-format csv -d $O/q_wait -o w -- python3 $R/bench.py $A > $O/q_w.log 2>&1 < /dev/null
cd $R
python3 - <<'PY'
import csv, glob, os, collections
O=os.path.join(os.environ["GRAFT_REPO_ROOT"],"gpurun_out","r06")
for d in ("q_mfma","q_lds","q_wait"):
    for f in glob.glob(os.path.join(O,d,"**","*counter_collection.csv"),recursive=True):
        agg=collections.defaultdict(lambda:[0.0,0,0])
        for r in csv.DictReader(open(f)):
            n=r["Kernel_Name"]
            if "row_hess" not in n and "eig_tridiag" not in n: continue
            k=(n[:60],r["Counter_Name"]); a=agg[k]; a[0]+=float(r["Counter_Value"]); a[1]+=1; a[2]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
        for k,a in sorted(agg.items()): print(d,k,"avg",a[0]/a[1],"n",a[1],"avg_ns",a[2]/a[1])
        os.remove(f)
PY
rm -rf $O/q_mfma $O/q_lds $O/q_wait
