#!/usr/bin/env python3
"""Per-iteration residuals and factor magnitudes of the C5L workload (native CSR X 1e6 x 1e5, Y in {0,1}, y logit Newton with the
notebook's l1 = 2, l2 = 5) for a few variants of the synthetic problem: which of them the reference's undamped iteration is stable on.

    python tools/r05_c5l_probe.py [--iters 14] [--small]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(lib, name, m, d, p, k, npr, l1, l2, y_param, zscale, vscale, iters, nn_mask=3):
    import scipy.sparse as sp
    ctx = lib.Context(0)
    ctx.set_problem(m, d, p, k)
    rng = np.random.default_rng(42)
    X = sp.csr_matrix((np.ones(m * npr), rng.integers(0, d, size=m * npr, dtype=np.int32), np.arange(0, m * npr + 1, npr, dtype=np.int64)), shape=(m, d))
    ctx.set_option("sparse_mode", 2)
    ctx.set_data(0, X)
    del X
    ctx.fill_data_synthetic(1, 43, 0, 0, 2, y_param)
    scale = (npr / d / k) ** 0.5
    ctx.fill_factor_synthetic(lib.CMF_U, 101, 0, scale)
    ctx.fill_factor_synthetic(lib.CMF_V, 102, 0, scale * vscale)
    ctx.fill_factor_synthetic(lib.CMF_Z, 103, 0, scale * zscale)
    x2, y2 = ctx.data_sq()
    rows = []
    for it in range(iters):
        ctx.newton_step(0.5, l1, l2, "linear", "logit", nn_mask, 7, 0.2, 1.0)
        ex, ey = ctx.residual_sq("linear", "logit")
        mx = [float(np.abs(ctx.get_factor(w)).max()) for w in range(3)]
        rows.append(dict(it=it + 1, rx=(ex / x2) ** 0.5, ry=(ey / y2) ** 0.5, U=mx[0], V=mx[1], Z=mx[2]))
        print(name, json.dumps(rows[-1]), flush=True)
        if not np.isfinite(ex + ey) or max(mx) > 1e6:
            break
    ctx.close()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=14)
    ap.add_argument("--small", action="store_true")
    args = ap.parse_args()
    from pycmf_amd import _lib
    m, d, p, k, npr = (1000000, 100000, 64, 256, 100) if not args.small else (20000, 3000, 64, 64, 30)
    out = {}
    for name, kw in (("ref_l1_2_l2_5", dict(l1=2.0, l2=5.0, y_param=0.1, zscale=1.0, vscale=1.0)),
                     ("l2_50", dict(l1=2.0, l2=50.0, y_param=0.1, zscale=1.0, vscale=1.0)),
                     ("l2_500", dict(l1=2.0, l2=500.0, y_param=0.1, zscale=1.0, vscale=1.0)),
                     ("y_half", dict(l1=2.0, l2=5.0, y_param=0.5, zscale=1.0, vscale=1.0)),
                     ("free_sign", dict(l1=2.0, l2=5.0, y_param=0.1, zscale=1.0, vscale=1.0, nn_mask=0))):
        out[name] = run(_lib, name, m, d, p, k, npr, iters=args.iters, **kw)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r05_c5l_probe.json"), "w"))


if __name__ == "__main__":
    main()
