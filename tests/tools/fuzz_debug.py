"""Replays chosen lines of fuzz_flagged.jsonl under a few option sets and prints the worst V rows (debugging aid)."""
import json
import sys

import fuzz_campaign as FC

lines = open("fuzz_flagged.jsonl").read().splitlines()
for ln in [int(a) for a in sys.argv[1:]]:
    c = json.loads(lines[ln])["case"]
    print({k: v for k, v in c.items() if k != "options"})
    for opts in ({}, {"refine_rows_ratio": 1}, {"sparse_mode": 1}, {"sparse_mode": 1, "refine_rows_ratio": 1}, {"row_classes": 0},
                 {"row_classes": 0, "refine_rows_ratio": 1}, {"refine_rows": 0}):
        cc = dict(c); cc["options"] = opts
        info = {"want_rows": True}
        try:
            e = FC.run_case(cc, c["seed"], info)
            print("   ", opts, ["%.1e" % v for v in e], {k: info[k] for k in ("refined_rows", "clamp_rows", "clamp_ratio", "v_rows_above_1e-3", "worst_v_rows") if k in info})
        except Exception as ex:
            print("   ", opts, repr(ex)[:200])
        sys.stdout.flush()
