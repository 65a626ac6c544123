#!/bin/bash
# Round 4: U and Z updates in one launch behind the paired data passes -- parity, C2 lines (pair_passes 1 / 2 / 0 alternating)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_step10
mkdir -p "$O"
cd "$R"
timeout 600 python3 -m pytest tests/test_gpu_mu.py tests/test_gpu_run_loop.py -x -q -m gpu > "$O/pytest_mu.txt" 2>&1; tail -n 5 "$O/pytest_mu.txt"
for rep in 1 2 3; do
for pair in 1 2 0; do
  timeout 300 python3 bench.py --workload c2 --steps 300 --warmup 30 --no-cpu-baseline --option pair_passes=$pair > "$O/bench_pair${pair}_$rep.json" 2> "$O/bench_pair${pair}_$rep.err"
done
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "it/s %.1f ms %.4f" % (d["value"], d["ms_per_step"]), {k: round(v, 4) for k, v in d["roofline"]["per_class_ms_per_step"].items()}, "frac %.3f" % d["roofline"]["frac"], d["rel_residual"]["x"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace('.json', '.err')).read()[-400:])
PY
