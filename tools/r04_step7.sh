#!/bin/bash
# Round 4: k_pad = 128 data passes -- pinned TN fragment bases + two-tile fill, A/B against the build before them, C2 lines
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_step7
mkdir -p "$O"
cd "$R"
python3 tools/ab_lib_versions.py 16384,8192,4096,128 60 > "$O/ab_c2.txt" 2>&1; tail -n 3 "$O/ab_c2.txt"
python3 tools/ab_lib_versions.py 16384,8192,4096,128 60 > "$O/ab_c2_b.txt" 2>&1; tail -n 3 "$O/ab_c2_b.txt"
python3 -m pytest tests/test_gpu_mu.py tests/test_gpu_fullsize.py -x -q -m gpu -k "mu or c2 or C2" > "$O/pytest_mu.txt" 2>&1; tail -n 3 "$O/pytest_mu.txt"
for rep in 1 2; do
for pipe in 4 5; do
  python3 bench.py --workload c2 --steps 300 --warmup 30 --no-cpu-baseline --option gemm_pipe=$pipe > "$O/bench_pipe${pipe}_$rep.json" 2> "$O/bench_pipe${pipe}_$rep.err"
done
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "it/s %.1f ms %.4f" % (d["value"], d["ms_per_step"]), {k: round(v, 4) for k, v in d["roofline"]["per_class_ms_per_step"].items()})
    except Exception as e:
        print(f, "ERR", e, open(f.replace('.json', '.err')).read()[-400:])
PY
