"""Replays the flagged cases of a tests/tools/fuzz_campaign.py log: as drawn, with default options, and with each drawn option removed in
turn -- which option (if any) carries the discrepancy.   python tests/tools/fuzz_bisect.py gpurun_out/fuzz1.jsonl [...]"""
import json
import sys

import fuzz_campaign as FC

for path in [a for a in sys.argv[1:] if not a.startswith('--')]:
    for line in open(path):
        d = json.loads(line)
        if not d.get("bad") or "case" not in d:
            continue
        c = d["case"]
        rows = []
        def run(tag, opts):
            cc = dict(c); cc["options"] = opts
            try:
                info = {}
                e = FC.run_case(cc, c["seed"], info)
                rows.append((tag, ["%.1e" % v for v in e] + ["%s=%.1e" % kv for kv in info.items()]))
            except Exception as ex:
                rows.append((tag, repr(ex)[:120]))
        run("as drawn", c["options"])
        run("defaults", {})
        if "--full" in sys.argv:
            for n in c["options"]:
                run("without " + n, {k: v for k, v in c["options"].items() if k != n})
        for extra in ({"newton_schulz": 0}, {"row_certificates": 0}, {"row_kernel": 0}, {"safe_inverse_cholesky": 0}):
            run("defaults + %s" % extra, extra)
        print(json.dumps({k: v for k, v in c.items() if k != "options"}), c["options"])
        for r in rows:
            print("    ", r[0], r[1])
        sys.stdout.flush()
