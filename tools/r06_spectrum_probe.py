#!/usr/bin/env python3
"""r06 probe: what does `_safe_invert`'s clamp (pycmf/cmf_solvers.py:346-356) actually see at C3 with the reference's default
l2 = 0 (and at C3X)?  Runs the workload to iteration N on the device, then recomputes the per-row Hessians of a few rows of every
sweep in float64 on the host from the device's factors and the device sampler's lists, and prints their spectra against pert:
how many eigenvalues sit below the threshold, how far, and how the rows' invariant subspaces relate to the shared Gram's.

    python tools/r06_spectrum_probe.py [--workload c3z] [--iters 10] [--rows 4] [--option refine_rows=0]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3z")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rows", type=int, default=4)
    ap.add_argument("--scale", type=int, default=1, help="divide m, d, p by this (debug)")
    ap.add_argument("--option", action="append", default=[])
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    import numpy as np
    import bench
    from pycmf_amd import _lib
    w = dict(bench.WORKLOADS[args.workload])
    m, d, p, k = w["m"] // args.scale, w["d"] // args.scale, w["p"] // args.scale, w["k"]
    ctx = _lib.Context(0)
    for kv in args.option:
        name, _, val = kv.partition("=")
        ctx.set_option(name, int(val))
    ctx.set_problem(m, d, p, k)
    ctx.fill_data_synthetic(0, 42, 0, 0, w.get("x_kind", 0))
    ctx.fill_data_synthetic(1, 43, 0, 0, w.get("y_kind", 0), w.get("y_param", 0.0))
    scale = (0.7979 / k) ** 0.5
    ctx.fill_factor_synthetic(_lib.CMF_U, 101, 0, scale)
    ctx.fill_factor_synthetic(_lib.CMF_V, 102, 0, scale)
    ctx.fill_factor_synthetic(_lib.CMF_Z, 103, 0, scale)
    l1, l2 = w.get("l1", 0.0), w.get("l2", 0.1)
    pert, alpha, ratio = 0.2, 0.5, w["ratio"]
    out = {"workload": args.workload, "m": m, "d": d, "p": p, "k": k, "l2": l2, "iters": []}
    for it in range(args.iters):
        t0 = time.perf_counter()
        ctx.newton_step_device_sampled(alpha, l1, l2, w["x_link"], w["y_link"], 0, 7, pert, ratio, 1000 + it)
        ctx.sync()
        st = ctx.newton_clamp_stats(full=True)
        print("iter %d: %.1f ms  clamp_stats %s" % (it, (time.perf_counter() - t0) * 1e3, st), flush=True)
    U = ctx.get_factor(_lib.CMF_U).astype(np.float64)
    V = ctx.get_factor(_lib.CMF_V).astype(np.float64)
    Z = ctx.get_factor(_lib.CMF_Z).astype(np.float64)
    seed = 1000 + args.iters   # the lists the NEXT iteration would draw
    sig = lambda t: 1.0 / (1.0 + np.exp(-t))

    def spectrum(name, H):
        ev = np.linalg.eigvalsh(H)
        nb = int((ev < pert).sum())
        q = np.quantile(ev, [0, 0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 1.0])
        near = int((np.abs(ev - pert) < 0.1 * pert).sum())
        print("  %-10s below pert: %3d / %d   within 10%% of pert: %d   min %.3e  q1%% %.3e q10 %.3e q25 %.3e med %.3e q75 %.3e q90 %.3e max %.3e  fro %.3e"
              % ((name, nb, len(ev), near) + tuple(q) + (np.linalg.norm(H),)), flush=True)
        return {"name": name, "below": nb, "near": near, "quantiles": q.tolist(), "ev": ev.tolist()}

    def norms(name, F):
        s = np.linalg.svd(F, compute_uv=False)
        print("%s: shape %s  |F|max %.3e  sv max %.3e  sv[k/2] %.3e  sv min %.3e" % (name, F.shape, np.abs(F).max(), s[0], s[len(s) // 2], s[-1]))
        return s
    norms("U", U); sV = norms("V", V); norms("Z", Z)
    G = alpha * V.T @ V
    print("shared alpha V^T V:")
    gs = spectrum("G", G)
    evG, QG = np.linalg.eigh(G)
    res = {"G": gs, "rows": []}
    rows = np.linspace(0, m - 1, args.rows).astype(int)
    print("U sweep (x %s): H_i = alpha V_S^T D V_S" % w["x_link"])
    for i in rows:
        lst = ctx.sample_lists(0, seed, ratio, int(i), 1)[0] if ratio < 1 else np.arange(d)
        Vs = V[lst]
        if w["x_link"] == "logit":
            s_ = sig(Vs @ U[i])
            wgt = s_ * (1 - s_)
            H = alpha * (Vs * wgt[:, None]).T @ Vs
        else:
            H = alpha * Vs.T @ Vs
        H = H + (l2 if w["x_link"] != "logit" else 0.0) * np.eye(k)
        r = spectrum("U[%d]" % i, H)
        # how diagonal is H in the eigenbasis of the shared Gram?
        Hq = QG.T @ H @ QG
        dg = np.diag(Hq)
        off = Hq - np.diag(dg)
        sc = off / np.sqrt(np.abs(np.outer(dg, dg)) + 1e-300)
        print("      in G's eigenbasis: |offdiag|_F / |diag|_2 = %.3e, scaled offdiag max %.3e, 2-norm %.3e"
              % (np.linalg.norm(off) / np.linalg.norm(dg), np.abs(sc).max(), np.linalg.norm(sc, 2)))
        # subspace below pert of H vs of G * s/d
        ev, Q = np.linalg.eigh(H)
        nb = int((ev < pert).sum())
        if 0 < nb < k:
            evs = evG * (len(lst) / float(d))
            nbG = int((evs < pert).sum())
            Pg = QG[:, :max(nbG, 1)]
            Ph = Q[:, :nb]
            sv = np.linalg.svd(Pg.T @ Ph, compute_uv=False)
            print("      G-predicted below: %d; principal cosines between the two low subspaces: min %.4f median %.4f" % (nbG, sv.min(), np.median(sv)))
        res["rows"].append(r)
    print("Z sweep (y %s):" % w["y_link"])
    for i in np.linspace(0, p - 1, args.rows).astype(int):
        lst = ctx.sample_lists(1, seed, ratio, int(i), 1)[0] if ratio < 1 else np.arange(d)
        Vs = V[lst]
        if w["y_link"] == "logit":
            s_ = sig(Vs @ Z[i])
            wgt = s_ * (1 - s_)
            H = (1 - alpha) * (Vs * wgt[:, None]).T @ Vs
        else:
            H = (1 - alpha) * Vs.T @ Vs
        H = H + l2 * np.eye(k)
        res["rows"].append(spectrum("Z[%d]" % i, H))
    print("V sweep:")
    for i in np.linspace(0, d - 1, args.rows).astype(int):
        l1_ = ctx.sample_lists(2, seed, ratio, int(i), 1)[0] if ratio < 1 else np.arange(m)
        l2_ = ctx.sample_lists(3, seed, ratio, int(i), 1)[0] if ratio < 1 else np.arange(p)
        Us, Zs = U[l1_], Z[l2_]
        if w["x_link"] == "logit":
            s_ = sig(Us @ V[i]); Hx = (Us * (s_ * (1 - s_))[:, None]).T @ Us
        else:
            Hx = Us.T @ Us
        if w["y_link"] == "logit":
            s_ = sig(Zs @ V[i]); Hy = (Zs * (s_ * (1 - s_))[:, None]).T @ Zs
        else:
            Hy = Zs.T @ Zs
        H = alpha * Hx + (1 - alpha) * Hy + l2 * np.eye(k)
        res["rows"].append(spectrum("V[%d]" % i, H))
    if args.out:
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(res, f)
    ctx.close()


if __name__ == "__main__":
    main()
