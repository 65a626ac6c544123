#!/bin/bash
# Round 4: MFMA-blocked Cholesky wired into the per-row Newton solves: tests + C3 bench, A/B against the rank-1 kernel
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_step4
mkdir -p "$O"
cd "$R"
python3 -m pytest tests/test_gpu_newton.py tests/test_gpu_midrange.py tests/test_gpu_shared64.py tests/test_gpu_fuzz.py -x -q -m gpu > "$O/pytest_newton.txt" 2>&1
tail -n 6 "$O/pytest_newton.txt"
python3 bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline > "$O/bench_c3.json" 2> "$O/bench_c3.err"
python3 bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --option chol_mfma=0 > "$O/bench_c3_old.json" 2> "$O/bench_c3_old.err"
python3 - <<PY
import json
for f in ["bench_c3","bench_c3_old"]:
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
        print(f, "ms %.2f"%d["ms_per_step"], {k:round(v,2) for k,v in d["roofline"]["per_class_ms_per_step"].items()}, d["rel_residual"])
    except Exception as e:
        print(f, "ERR", e, open("$O/%s.err"%f).read()[-800:])
PY
