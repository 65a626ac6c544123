#!/bin/bash
# Round 4: C2 with the paired launch -- small Grams on the side stream, row shares, graph replay (one box, alternating)
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_step13
mkdir -p "$O"
cd "$R"
i=0
for opt in "pair_passes=1" "side_gram=1" "small_gram_shares=64" "graph=1" "pair_passes=1" "side_gram=1" "small_gram_shares=16" "graph=1"; do
  i=$((i+1))
  timeout 300 python3 bench.py --workload c2 --steps 300 --warmup 30 --no-cpu-baseline --option $opt > "$O/bench_${i}_$opt.json" 2> "$O/bench_${i}_$opt.err"
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
