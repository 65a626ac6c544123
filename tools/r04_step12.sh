#!/bin/bash
# Round 4: C4 with staging schedule 10 (direct-to-LDS loads) against schedule 4, alternating, per-class times
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_step12
mkdir -p "$O"
cd "$R"
for rep in 1 2; do
for pipe in 4 10 5; do
  timeout 600 python3 bench.py --steps 12 --warmup 3 --no-cpu-baseline --option gemm_pipe=$pipe > "$O/bench_c4_p${pipe}_$rep.json" 2> "$O/bench_c4_p${pipe}_$rep.err"
done
done
python3 - <<PY
import json, glob
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], "it/s %.3f ms %.3f" % (d["value"], d["ms_per_step"]), {k: round(v, 3) for k, v in d["roofline"]["per_class_ms_per_step"].items()}, "frac %.4f" % d["roofline"]["frac"])
    except Exception as e:
        print(f, "ERR", e, open(f.replace('.json', '.err')).read()[-400:])
PY
