# HBM-side traffic of the C3 row kernels: FETCH_SIZE and WRITE_SIZE in separate PMC passes (MI355X_MICROARCH.md, HBM section)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_c3_traffic
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/f -o f -- python3 $R/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline > $O/f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/w -o w -- python3 $R/bench.py --workload c3 --steps 1 --warmup 1 --no-cpu-baseline > $O/w.log 2>&1
cd $R
python3 - <<'PY'
import csv, glob, json, os
root = os.environ["GRAFT_REPO_ROOT"]
acc = {}
for f in glob.glob(os.path.join(root, "gpurun_out/pmc_c3_traffic/**/*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "row_hess_kernel" not in n:
            continue
        key = ("class" if "3, 1>" in n else "rows", r["Counter_Name"])
        acc.setdefault(key, []).append(float(r["Counter_Value"]))
    os.remove(f)
out = {}
for kind in ("class", "rows"):
    fe = acc.get((kind, "FETCH_SIZE"), []); wr = acc.get((kind, "WRITE_SIZE"), [])
    if fe and wr:
        # FETCH_SIZE: KiB, wide reads tallied at half size on gfx950 (x2); WRITE_SIZE: KiB
        out[kind] = {"launches": len(fe), "fetch_bytes_per_launch": sum(fe) / len(fe) * 1024 * 2, "write_bytes_per_launch": sum(wr) / len(wr) * 1024}
print(json.dumps(out, indent=1))
json.dump(out, open(os.path.join(root, "gpurun_out/pmc_c3_traffic/summary.json"), "w"), indent=1)
PY
