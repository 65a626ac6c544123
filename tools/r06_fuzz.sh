#!/bin/bash
# differential campaign on the final build of round 6 (general mix, then dense MU at k_pad = 128), summarised to gpurun_out/fuzz_r06_summary.json
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"
python3 tests/tools/fuzz_campaign.py --minutes ${1:-10} --seed ${3:-505} > gpurun_out/fuzz_r06.jsonl 2> gpurun_out/fuzz_r06.err
python3 tests/tools/fuzz_campaign.py --minutes ${2:-4} --seed ${4:-55} --focus pair > gpurun_out/fuzz_r06_pair.jsonl 2> gpurun_out/fuzz_r06_pair.err
python3 - <<'PY'
import json, collections
out = {}
for tag in ("fuzz_r06", "fuzz_r06_pair"):
    rows = [json.loads(l) for l in open("gpurun_out/%s.jsonl" % tag) if l.startswith("{")]
    cases = [r for r in rows if "case" in r]
    groups = collections.OrderedDict()
    for r in cases:
        c = r["case"]
        if c["solver"] == "mu":
            g = "mu"
        elif c["x_link"] == "linear" and c["y_link"] == "linear" and c["ratio"] >= 1.0:
            g = "newton_shared"
        else:
            g = "newton_per_row"
        e = max(r["err"]) if "err" in r else float("inf")
        d = groups.setdefault(g, {"cases": 0, "bad": 0, "largest_error": 0.0, "largest_residual_rel": 0.0, "k129_256": 0, "refined_cases": 0})
        d["cases"] += 1; d["bad"] += bool(r["bad"]); d["largest_error"] = max(d["largest_error"], e)
        d["largest_residual_rel"] = max(d["largest_residual_rel"], r.get("residual_rel", 0.0))
        d["k129_256"] += 128 < c["k"] <= 256
        d["refined_cases"] += r.get("refined_rows", 0) > 0
    opts = collections.Counter()
    for r in cases:
        for n, v in r["case"]["options"].items():
            if n in ("eig_clamp", "refine_rows_tol_ppm", "spmm_split", "trace_error", "refine_rows_batched", "chol_mfma", "newton_schulz", "rank1_clamp", "row_symmetric"):
                opts["%s=%d" % (n, v)] += 1
    worst = sorted(cases, key=lambda r: -(max(r["err"]) if "err" in r else 9e9))[:3]
    out[tag] = {"groups": groups, "option_draws": dict(opts), "bad_cases": [r for r in cases if r["bad"]][:10],
                "worst": [{"err": r.get("err"), "case": {k: v for k, v in r["case"].items() if k != "options"}, "clamp_ratio": r.get("clamp_ratio"), "refined_rows": r.get("refined_rows")} for r in worst]}
json.dump(out, open("gpurun_out/fuzz_r06_summary.json", "w"), indent=1)
print(json.dumps({t: {g: dict(v) for g, v in o["groups"].items()} for t, o in out.items()}, indent=1))
PY
