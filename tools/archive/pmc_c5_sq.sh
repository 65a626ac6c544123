# SQ instruction mix / wait breakdown of the C5 SpMM kernel (one PMC pass, SQ counters only)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_c5_sq
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS --output-format csv -d $O -o sq -- python3 $R/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline > $O/log.txt 2>&1
cd $R
python3 - <<'PY'
import csv, glob, os
f = glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/pmc_c5_sq/**/*counter_collection.csv"), recursive=True)[0]
acc = {}
for r in csv.DictReader(open(f)):
    if "spmm_blocked" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    print(k, sum(v) / len(v), len(v))
os.remove(f)
PY
