"""Time-bounded differential campaign at MID sizes: random shapes (multi-tile, chunked, ragged), k up to 300, every link pair,
sampled / unsampled, dense / CSR X, random internal options -- one Newton step (or three MU steps) through the C ABI against the
float64 oracle.  tests/test_gpu_fuzz.py is the committed small-shape slice of this; the campaign is the long-running form used to
look for latent defects on a GPU box:

    python tests/tools/fuzz_campaign.py --minutes 10 --seed 0 > gpurun_out/fuzz.jsonl

One JSON line per case: the configuration, max |device - oracle| / max |oracle| per factor, and "bad" when above the threshold.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import cmf_oracle as O          # noqa: E402  (test infrastructure: the checker)
from pycmf_amd import _lib                  # noqa: E402

OPTIONS = {"row_classes": [-1, 0, 2, 3, 5], "row_certificates": [0, 1], "lowrank_rows": [0, 1, 2], "row_split": [0, 1],
           "row_chunk": [0, 256, 512], "small_gram": [0, 1], "gemm_split": [0, 1, 3], "row_symmetric": [0, 1, 3, 4],
           "row_kernel": [0, 1], "direct_newton_step": [0, 1], "small_tile_update": [0, 1], "fused_mu_update": [0, 1],
           "split_reduce_in_kernel": [0, 1], "spmm_blocked": [0, 1], "newton_schulz": [0, 1], "safe_inverse_cholesky": [0, 1],
           "factor_times_tile": [64, 128, 256], "graph": [0, 1], "chol_mfma": [0, 1], "refine_rows_batched": [0, 1], "narrow_update": [0, 1],
           "side_gram": [0, 1], "pair_passes": [0, 1, 2],
           "nt_tile16": [0, 1], "nt_bn256": [0, 1], "nt_raster": [0, 1], "gemm64_tile128": [0, 1], "refine_spectral_map": [0, 1],   # round 5
           "eig_clamp": [0, 1, 3], "refine_rows_tol_ppm": [0, 20, 1000], "spmm_split": [0, 1], "trace_error": [0, 1],               # round 6
           "rank1_clamp": [0, 1]}


def log_int(rng, lo, hi):
    return int(round(np.exp(rng.uniform(np.log(lo), np.log(hi)))))


def draw_case(rng, solver, focus=None):
    k = int(rng.choice([1, 2, 3, 7, 10, 20, 33, 64, 65, 100, 128, 129, 200, 256, 300],
                       p=[.04, .04, .06, .08, .12, .12, .08, .1, .06, .06, .08, .04, .04, .06, .02]))
    if focus == "pair":                     # dense MU at k_pad = 128, larger shapes: the balanced two-product launch (cmf_gemm_pair.hip.h)
        k = int(rng.choice([65, 70, 100, 127, 128]))
    hi = 2500 if k <= 64 else (900 if k <= 130 else 400)
    if focus == "pair":
        hi = 6000
    m, d, p = (log_int(rng, 40, hi) for _ in range(3))
    if rng.rand() < 0.25:
        p = log_int(rng, 1, 64)             # the low-rank V sweep (p <= 64 < k)
    c = {"solver": solver, "m": m, "d": d, "p": p, "k": k, "csr": bool(rng.rand() < 0.4) and focus != "pair", "l1": 0.0, "l2": 0.0}
    if rng.rand() < 0.7:
        c["l1"] = float(rng.choice([0.0, rng.rand() * 0.3, 2.0]))
        c["l2"] = float(rng.choice([0.0, rng.rand() * 0.5, 5.0]))
    if solver == "newton":
        c.update(x_link=str(rng.choice(["linear", "logit"], p=[.65, .35])), y_link=str(rng.choice(["linear", "logit"], p=[.6, .4])),
                 ratio=float(rng.choice([1.0, 0.3, 0.5, 0.8], p=[.5, .15, .2, .15])), nn=int(rng.randint(0, 8)),
                 alpha=float(rng.choice([0.5, 0.2 + 0.6 * rng.rand(), 0.0, 1.0], p=[.3, .6, .05, .05])),
                 pert=float(rng.choice([0.2, 0.01, 1.0], p=[.7, .15, .15])), mask=int(rng.choice([7, 7, 7, rng.randint(1, 8)])))
        if c["ratio"] < 1 and k > 130:
            c["m"], c["d"], c["p"] = min(m, 300), min(d, 300), min(p, 300)     # the oracle solves every row with eigh
    else:
        c["mask"] = int(rng.choice([7, 7, rng.randint(1, 8)]))
    c["options"] = {n: int(rng.choice(v)) for n, v in OPTIONS.items() if rng.rand() < 0.25}
    return c


def run_case(c, seed, info=None):
    rng = np.random.RandomState(seed)
    m, d, p, k = c["m"], c["d"], c["p"], c["k"]
    newton = c["solver"] == "newton"
    xl, yl = (c["x_link"], c["y_link"]) if newton else ("linear", "linear")
    X = rng.rand(m, d) if xl == "logit" else np.abs(rng.randn(m, d))
    Y = (rng.rand(d, p) < 0.3).astype(float) if (yl == "logit" and rng.rand() < 0.5) else (rng.rand(d, p) if yl == "logit" else np.abs(rng.randn(d, p)))
    if c["csr"]:
        X[rng.rand(m, d) < 0.9] = 0.0
    sc = 0.4 / np.sqrt(max(1.0, k / 8.0))
    U0, V0, Z0 = sc * rng.randn(m, k), sc * rng.randn(d, k), sc * rng.randn(p, k)
    if not newton:
        U0, V0, Z0 = np.abs(U0) + 0.02, np.abs(V0) + 0.02, np.abs(Z0) + 0.02
    else:
        if c["nn"] & 1: U0 = np.abs(U0)
        if c["nn"] & 2: V0 = np.abs(V0)
        if c["nn"] & 4: Z0 = np.abs(Z0)
    mask = c["mask"]
    ctx = _lib.Context(0)
    try:
        ctx.set_problem(m, d, p, k)
        for n, v in c["options"].items():
            ctx.set_option(n, v)
        ctx.set_data(0, sp.csr_matrix(X) if c["csr"] else X)
        ctx.set_data(1, Y)
        for w, F in enumerate((U0, V0, Z0)):
            ctx.set_factor(w, F)
        Ur, Vr, Zr = U0.copy(), V0.copy(), Z0.copy()
        if not newton:
            for _ in range(3):
                ctx.mu_step(c["l1"], c["l2"], mask)
                O.mu_update_step(X, Y, Ur, Vr, Zr, c["l1"], c["l2"], update_U=bool(mask & 1), update_V=bool(mask & 2), update_Z=bool(mask & 4))
        else:
            np.random.seed(seed % (2 ** 31))
            masks = {"U": [], "Z": [], "V": []}
            nn, ratio = c["nn"], c["ratio"]
            O.newton_update_step(X, Y, Ur, Vr, Zr, c["alpha"], c["l1"], c["l2"], xl, yl, bool(nn & 1), bool(nn & 2), bool(nn & 4),
                                 ratio, c["pert"], update_U=bool(mask & 1), update_V=bool(mask & 2), update_Z=bool(mask & 4), masks=masks)
            lists = [None] * 4
            if ratio < 1:
                su, sm, sp_ = int(d * ratio), int(m * ratio), int(p * ratio)
                z = lambda rows, per: np.zeros((rows, per), dtype=np.int32)
                lists = [np.array(masks["U"], dtype=np.int32).reshape(m, su) if mask & 1 else z(m, su),
                         np.array(masks["Z"], dtype=np.int32).reshape(p, su) if mask & 4 else z(p, su),
                         np.array([a for a, _ in masks["V"]], dtype=np.int32).reshape(d, sm) if mask & 2 else z(d, sm),
                         np.array([b for _, b in masks["V"]], dtype=np.int32).reshape(d, sp_) if mask & 2 else z(d, sp_)]
                if not (mask & 1): lists[0][:] = np.arange(su)[None, :]
                if not (mask & 4): lists[1][:] = np.arange(su)[None, :]
                if not (mask & 2): lists[2][:] = np.arange(sm)[None, :]; lists[3][:] = np.arange(sp_)[None, :]
            ctx.newton_step(c["alpha"], c["l1"], c["l2"], xl, yl, nn, mask, c["pert"], ratio, *lists)
        errs, got = [], []
        for w, ref in enumerate((Ur, Vr, Zr)):
            got.append(ctx.get_factor(w))
            if not np.isfinite(got[w]).all():
                errs.append(float("inf"))
            else:
                errs.append(float(np.abs(got[w] - ref).max() / max(1e-3, np.abs(ref).max())))
        if info is not None and info.get("want_rows"):       # debugging aid: the V rows furthest from the oracle
            dv = np.abs(got[1] - Vr).max(axis=1) / max(1e-3, np.abs(Vr).max())
            order = np.argsort(-dv)[:6]
            info["worst_v_rows"] = [(int(i), float("%.2e" % dv[i])) for i in order]
            info["v_rows_above_1e-3"] = int((dv > 1e-3).sum())
        if info is not None and newton:
            info["clamp_rows"], info["clamp_ratio"], info["refined_rows"] = ctx.newton_clamp_stats()
        if info is not None and np.isfinite(errs).all():
            a = c.get("alpha", 0.5)
            e_ref = O.weighted_error(X, Y, Ur, Vr, Zr, a, xl, yl)
            e_dev = O.weighted_error(X, Y, got[0], got[1], got[2], a, xl, yl)
            e_data = O.weighted_error(X, Y, 0 * Ur, 0 * Vr, 0 * Zr, a, "linear", "linear")   # a ||X|| + (1 - a) ||Y||: the floor of "relative"
            info["residual_rel"] = float(abs(e_dev - e_ref) / max(e_ref, 1e-6 * e_data))  # north_star's measure (an exact fit has e_ref ~ 1e-13)
            if newton:      # upper bound of ||H|| / lambda_min-after-clamp over the V sweep's Hessians (logit weights <= 1/4)
                su2, sz2 = np.linalg.norm(Ur, 2) ** 2, np.linalg.norm(Zr, 2) ** 2
                info["cond_v"] = float((a * su2 * (0.25 if xl == "logit" else 1.0) + (1 - a) * sz2 * (0.25 if yl == "logit" else 1.0) + c["l2"])
                                       / max(c["pert"], c["l2"]))
        return errs
    finally:
        ctx.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=5.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--threshold", type=float, default=3e-3)
    ap.add_argument("--replay", type=str, default=None, help="JSON of one case (a line of a previous run) to run again")
    ap.add_argument("--focus", type=str, default=None, choices=[None, "pair"], help="pair: dense MU cases at k_pad = 128 only")
    args = ap.parse_args()
    if args.replay:
        c = json.loads(args.replay)
        info = {}
        errs = run_case(c, c["seed"], info)
        print(json.dumps(dict({"case": c, "err": errs}, **info)))
        return
    rng = np.random.RandomState(args.seed)
    t_end = time.time() + 60 * args.minutes
    n = bad = 0
    while time.time() < t_end:
        c = draw_case(rng, "mu", "pair") if args.focus == "pair" else draw_case(rng, "newton" if rng.rand() < 0.75 else "mu")
        c["seed"] = int(rng.randint(1, 2 ** 31 - 1))
        t0 = time.time()
        try:
            info = {}
            errs = run_case(c, c["seed"], info)
            out = {"case": c, "err": errs, "s": round(time.time() - t0, 2), "bad": bool(max(errs) > args.threshold)}
            out.update(info)
        except np.linalg.LinAlgError as e:     # the CHECKER's own eigh gave up (LAPACK "Internal Error" on a finite matrix: the
            out = {"case": c, "error": repr(e)[:300], "bad": False, "oracle_error": True}   # reference would stop here too): no verdict
        except Exception as e:                 # a refusal (CMF_EUNSUPPORTED ...) is a finding too: keep going
            out = {"case": c, "error": repr(e)[:300], "bad": True}
        n += 1
        bad += out["bad"]
        print(json.dumps(out), flush=True)
    print(json.dumps({"cases": n, "bad": bad}), flush=True)


if __name__ == "__main__":
    main()
