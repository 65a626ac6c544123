#!/bin/bash
# Round 4: batched refinement test, SpMM column-block sweep at C5, c3z / c3x bench lines
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_step5
mkdir -p "$O"
cd "$R"
python3 -m pytest tests/test_gpu_conditioning.py -x -q -m gpu -k "batched or redone or across" > "$O/pytest_cond.txt" 2>&1
tail -n 4 "$O/pytest_cond.txt"
for w in c3z c3x; do
  python3 bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline > "$O/bench_$w.json" 2> "$O/bench_$w.err"
done
for bc in 0 1024 1536 3072; do
  python3 bench.py --workload c5 --steps 10 --warmup 3 --no-cpu-baseline --option spmm_block_cols=$bc > "$O/bench_c5_bc$bc.json" 2> "$O/bench_c5_bc$bc.err"
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "it/s %.3f ms %.2f"%(d["value"], d["ms_per_step"]), {k:round(v,2) for k,v in d["roofline"]["per_class_ms_per_step"].items()}, d.get("conditioning",{}).get("rows_refined_in_float64"), d["rel_residual"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace('.json','.err')).read()[-600:])
PY
