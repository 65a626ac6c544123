"""r06: the 640 x 576 / 576 x 320, k = 256 sub-problem at l2 = 0 (x linear, y logit), per iteration: relative residuals of the
device (option sets given on the command line, name=value,... each) against the float64 oracle fed the same lists."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pycmf_amd import _lib as lib  # noqa: E402
from oracle import cmf_oracle as O  # noqa: E402
from threadpoolctl import threadpool_limits  # noqa: E402

m, d, p, k = 640, 576, 320, 256
alpha, l1, l2, pert, ratio, iters = 0.5, 0.0, 0.0, 0.2, 0.5, int(os.environ.get("ITERS", "12"))
sets = sys.argv[1:] or [""]


def make(opts):
    ctx = lib.Context(0)
    for kv in opts.split(","):
        if kv:
            ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    ctx.set_problem(m, d, p, k)
    ctx.fill_data_synthetic(0, 42, 0, 0)
    ctx.fill_data_synthetic(1, 43, 0, 0, 1)
    scale = (0.7979 / k) ** 0.5
    for w, seed in ((lib.CMF_U, 101), (lib.CMF_V, 102), (lib.CMF_Z, 103)):
        ctx.fill_factor_synthetic(w, seed, 0, scale)
    return ctx


ctxs = [make(o) for o in sets]
X = ctxs[0].get_data(0).astype(np.float64)
Y = ctxs[0].get_data(1).astype(np.float64)
U, V, Z = (ctxs[0].get_factor(w) for w in range(3))
sig = lambda t: 1.0 / (1.0 + np.exp(-t))
for it in range(1, iters + 1):
    seed = 700 + it
    for c in ctxs:
        c.newton_step_device_sampled(alpha, l1, l2, "linear", "logit", 0, 7, pert, ratio, seed)
    c = ctxs[0]
    lists = [row for row in c.sample_lists(0, seed, ratio, 0, m)] + [row for row in c.sample_lists(1, seed, ratio, 0, p)]
    lx, ly = c.sample_lists(2, seed, ratio, 0, d), c.sample_lists(3, seed, ratio, 0, d)
    for q in range(d):
        lists += [lx[q], ly[q]]
    itl = iter(lists)
    O.draw_sample = lambda n, ratio: next(itl)
    with threadpool_limits(limits=1):
        O.newton_update_step(X, Y, U, V, Z, alpha, l1, l2, "linear", "logit", False, False, False, ratio=ratio, pert=pert)
    rx = np.linalg.norm(X - U @ V.T) / np.linalg.norm(X)
    ry = np.linalg.norm(Y - sig(V @ Z.T)) / np.linalg.norm(Y)
    line = "it %2d oracle %.6f %.6f |" % (it, rx, ry)
    for o, c in zip(sets, ctxs):
        Ug, Vg, Zg = (c.get_factor(w) for w in range(3))
        gx = np.linalg.norm(X - Ug @ Vg.T) / np.linalg.norm(X)
        gy = np.linalg.norm(Y - sig(Vg @ Zg.T)) / np.linalg.norm(Y)
        fac = max(np.abs(a - b).max() / np.abs(b).max() for a, b in ((Ug, U), (Vg, V), (Zg, Z)))
        st = c.newton_clamp_stats(full=True)
        line += " [%s] dX %.1e dY %.1e fac %.1e ref %d |" % (o or "default", abs(gx - rx) / rx, abs(gy - ry) / ry, fac, st[2])
    print(line, flush=True)
