# C3 row kernels (class launches vs logit launches): matrix-pipe busy, wait share and L2 hit rate, two PMC passes
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_c3_rows
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/a -o a -- python3 $R/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline > $O/a.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $O/b -o b -- python3 $R/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline > $O/b.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, os
for f in glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/pmc_c3_rows/**/*counter_collection.csv"), recursive=True):
    acc = {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "row_hess_kernel" not in n:
            continue
        key = ("class" if "3, 1>" in n else "rows", r["Counter_Name"])
        acc.setdefault(key, []).append(float(r["Counter_Value"]))
    for k, v in sorted(acc.items()):
        print(k[0], k[1], "%.4g" % (sum(v) / len(v)), len(v))
    os.remove(f)
PY
