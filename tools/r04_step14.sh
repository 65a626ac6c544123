#!/bin/bash
# Round 4: differential campaign on the final build -- the paired launch in focus (dense MU, k_pad = 128), then the general mix
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/r04_step14
mkdir -p "$O"
cd "$R"
timeout 300 python3 -m pytest tests/test_gpu_run_loop.py tests/test_gpu_mu.py -x -q -m gpu > "$O/pytest.txt" 2>&1; tail -n 3 "$O/pytest.txt"
timeout 700 python3 tests/tools/fuzz_campaign.py --minutes 8 --seed 41 --focus pair > "$O/fuzz_pair.jsonl" 2> "$O/fuzz_pair.err"
timeout 900 python3 tests/tools/fuzz_campaign.py --minutes 12 --seed 42 > "$O/fuzz_mix.jsonl" 2> "$O/fuzz_mix.err"
python3 - <<PY
import json
for n in ("fuzz_pair", "fuzz_mix"):
    rows = [json.loads(l) for l in open("$O/%s.jsonl" % n) if l.startswith("{")]
    bad = [r for r in rows if r.get("bad") or "error" in r]
    worst = max((max(r["err"]) for r in rows if "err" in r), default=0)
    pair = sum(1 for r in rows if r["case"]["solver"] == "mu" and 64 < r["case"]["k"] <= 128 and not r["case"]["csr"])
    print(n, "cases", len(rows), "bad", len(bad), "worst %.2e" % worst, "dense MU k_pad=128 cases", pair)
    for r in bad[:5]: print("   BAD", json.dumps(r)[:400])
PY
