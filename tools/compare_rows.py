#!/usr/bin/env python3
"""Compare two sets of factor-row dumps of bench.py --dump-rows (e.g. the N = 8 dress rehearsal against the N = 1 run).

    python tools/compare_rows.py PREFIX_A PREFIX_B [--tol 1e-5] [--out result.json]

Every set is PREFIX.rank<r>.npz for r = 0 .. N-1; a rank holds the rows it owns out of 16 fixed global rows of U, V and Z.  The
sets are merged by global row index and compared row by row: max |a - b| / max |factor| must be <= tol for each factor, and
both sets must cover the same rows.  Prints one JSON line; exit code 1 on a mismatch."""
import argparse
import glob
import json
import sys

import numpy as np


def load(prefix):
    files = sorted(glob.glob(prefix + ".rank*.npz"))
    if not files:
        raise SystemExit("no dumps under %s.rank*.npz" % prefix)
    rows = {"U": {}, "V": {}, "Z": {}}
    absmax = {"U": 0.0, "V": 0.0, "Z": 0.0}
    for f in files:
        z = np.load(f)
        for name in rows:
            for i, r in zip(z[name + "_rows"], z[name]):
                rows[name][int(i)] = r
            absmax[name] = max(absmax[name], float(z[name + "_absmax"][0]))
    return rows, absmax, len(files)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("a")
    ap.add_argument("b")
    ap.add_argument("--tol", type=float, default=1e-5)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    ra, ma, na = load(args.a)
    rb, mb, nb = load(args.b)
    res = {"a": args.a, "b": args.b, "ranks_a": na, "ranks_b": nb, "tol": args.tol, "factors": {}}
    ok = True
    for name in ("U", "V", "Z"):
        same_rows = sorted(ra[name]) == sorted(rb[name])
        err = max((float(np.abs(ra[name][i] - rb[name][i]).max()) for i in ra[name] if i in rb[name]), default=float("nan"))
        scale = max(ma[name], mb[name])
        rel = err / scale if scale > 0 else float("nan")
        good = same_rows and len(ra[name]) > 0 and np.isfinite(rel) and rel <= args.tol
        ok = ok and good
        res["factors"][name] = {"rows_compared": len(ra[name]), "same_rows": same_rows, "max_abs_diff": err, "max_abs_factor": scale,
                                "rel": rel, "ok": bool(good)}
    res["ok"] = bool(ok)
    line = json.dumps(res)
    print(line)
    if args.out:
        with open(args.out, "w") as f:
            f.write(line + "\n")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
