#!/bin/bash
# Round 4: class images [group][block][class] -- class tests, then the C3 line
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_step11
mkdir -p "$O"
cd "$R"
python3 -m pytest tests/test_gpu_newton.py tests/test_gpu_fullsize.py -x -q -m gpu -k "shared_partial or classes or class or certific or sub_problem or fullsize" > "$O/pytest_cls.txt" 2>&1
tail -n 4 "$O/pytest_cls.txt"
python3 bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline > "$O/bench_c3.json" 2> "$O/bench_c3.err"
python3 - <<PY
import json
d=json.loads(open("$O/bench_c3.json").read().strip().splitlines()[-1])
print("c3 it/s %.3f ms %.2f"%(d["value"], d["ms_per_step"]), {k:round(v,2) for k,v in d["roofline"]["per_class_ms_per_step"].items()}, d["rel_residual"])
PY
