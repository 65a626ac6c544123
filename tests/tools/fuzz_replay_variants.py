#!/usr/bin/env python3
"""Replay the cases of a JSON list [[tag, case], ...] (tests/tools/fuzz_r05_replay.json: the two cases round 5's second campaign flagged,
each under four option sets) through fuzz_campaign.py --replay and print the factor errors."""
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, "fuzz_r05_replay.json")
for tag, c in json.load(open(path)):
    q = subprocess.run([sys.executable, os.path.join(HERE, "fuzz_campaign.py"), "--replay", json.dumps(c)], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    try:
        r = json.loads(q.stdout.decode().strip().splitlines()[-1])
        print(tag, "k", c["k"], "err", [float("%.2e" % e) for e in r["err"]], "clamp_rows", r.get("clamp_rows"), "ratio_left", r.get("clamp_ratio"),
              "refined", r.get("refined_rows"), "residual_rel", r.get("residual_rel"))
    except Exception:
        print(tag, "ERR", q.stderr.decode()[-300:])
