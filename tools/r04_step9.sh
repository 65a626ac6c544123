#!/bin/bash
# Round 4: class_sum_blocks_kernel with 4 / 8 / 16 class blocks in flight per thread (C3), then the whole GPU suite
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_step9
mkdir -p "$O"
cd "$R"
for depth in 4 16 8 4 16; do
  timeout 600 python3 bench.py --workload c3 --steps 5 --warmup 2 --no-cpu-baseline --option class_sum_depth=$depth > "$O/bench_c3_d${depth}_$RANDOM.json" 2> "$O/bench_c3_d$depth.err"
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_c3_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "it/s %.3f ms %.2f" % (d["value"], d["ms_per_step"]), {k: round(v, 2) for k, v in d["roofline"]["per_class_ms_per_step"].items()})
    except Exception as e:
        print(f, "ERR", e)
PY
python3 -m pytest tests -q -m gpu --durations=15 > "$O/pytest_gpu.txt" 2>&1
tail -n 30 "$O/pytest_gpu.txt"
