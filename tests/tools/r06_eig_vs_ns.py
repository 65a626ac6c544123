"""r06 debug: the midrange sub-problem (x logit, l2 = 0) sweep by sweep with the tridiagonal eigen-solve on and off (round 5's
Newton-Schulz clamp): where do the two float32 clamp paths part, and which is nearer the float64 row update?"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pycmf_amd import _lib as lib  # noqa: E402
from oracle import cmf_oracle as O  # noqa: E402

m, d, p, k = 640, 576, 320, 256
alpha, l1, l2, pert, ratio = 0.5, 0.0, 0.0, 0.2, 0.5


def make(eig):
    ctx = lib.Context(0)
    ctx.set_option("eig_clamp", eig)
    ctx.set_option("refine_rows", 0)
    ctx.set_problem(m, d, p, k)
    ctx.fill_data_synthetic(0, 42, 0, 0, 1)
    ctx.fill_data_synthetic(1, 43, 0, 0, 1)
    scale = (0.7979 / k) ** 0.5
    for w, seed in ((lib.CMF_U, 101), (lib.CMF_V, 102), (lib.CMF_Z, 103)):
        ctx.fill_factor_synthetic(w, seed, 0, scale)
    return ctx


a, b = make(1), make(0)
X = a.get_data(0).astype(np.float64)
Y = a.get_data(1).astype(np.float64)
sig = lambda t: 1.0 / (1.0 + np.exp(-np.clip(t, -700, 700)))
for it in range(1, 5):
    seed = 500 + it
    for name, upd, which in (("U", lib.CMF_UPD_U, 0), ("Z", lib.CMF_UPD_Z, 2), ("V", lib.CMF_UPD_V, 1)):
        F0 = [a.get_factor(w).astype(np.float64) for w in range(3)]
        for w in range(3):
            b.set_factor(w, F0[w])      # both paths start every sweep from the same factors
        a.newton_step_device_sampled(alpha, l1, l2, "logit", "logit", 0, upd, pert, ratio, seed)
        b.newton_step_device_sampled(alpha, l1, l2, "logit", "logit", 0, upd, pert, ratio, seed)
        Fa, Fb = a.get_factor(which).astype(np.float64), b.get_factor(which).astype(np.float64)
        diff = np.abs(Fa - Fb).max(axis=1)
        i = int(diff.argmax())
        print("it %d sweep %s: max |eig - ns| = %.3e at row %d (max |F| %.3e); rows above 1e-3 of max: %d"
              % (it, name, diff.max(), i, np.abs(Fb).max(), int((diff > 1e-3 * np.abs(Fb).max()).sum())), flush=True)
        if diff.max() > 1e-4 * np.abs(Fb).max():
            U0, V0, Z0 = F0
            if name == "U":
                lst = a.sample_lists(0, seed, ratio, i, 1)[0]
                Vs = V0[lst]; s_ = sig(Vs @ U0[i]); r = s_ - X[i, lst]
                g = alpha * r @ Vs; H = alpha * (Vs * (s_ * (1 - s_))[:, None]).T @ Vs
                cur = U0[i]
            elif name == "Z":
                lst = a.sample_lists(1, seed, ratio, i, 1)[0]
                Vs = V0[lst]; s_ = sig(Vs @ Z0[i]); r = s_ - Y[lst, i]
                g = (1 - alpha) * r @ Vs; H = (1 - alpha) * (Vs * (s_ * (1 - s_))[:, None]).T @ Vs
                cur = Z0[i]
            else:
                lx = a.sample_lists(2, seed, ratio, i, 1)[0]; ly = a.sample_lists(3, seed, ratio, i, 1)[0]
                Us, Zs = U0[lx], Z0[ly]
                sx = sig(Us @ V0[i]); sy = sig(Zs @ V0[i])
                g = alpha * (sx - X[lx, i]) @ Us + (1 - alpha) * (sy - Y[i, ly]) @ Zs
                H = alpha * (Us * (sx * (1 - sx))[:, None]).T @ Us + (1 - alpha) * (Zs * (sy * (1 - sy))[:, None]).T @ Zs
                cur = V0[i]
            ref = cur - g @ O.safe_invert(H, pert)
            ev = np.linalg.eigvalsh(H)
            print("    float64 row: |eig - ref| %.3e  |ns - ref| %.3e   ||g|| %.3e  H: min %.3e max %.3e, below pert %d, within 1e-3 of pert %d"
                  % (np.abs(Fa[i] - ref).max(), np.abs(Fb[i] - ref).max(), np.linalg.norm(g), ev.min(), ev.max(), int((ev < pert).sum()),
                     int((np.abs(ev - pert) < 1e-3 * pert).sum())), flush=True)
            # the solve alone, both device paths, on the float64 Hessian rounded to float32
            for meth in (1, 0):
                c2 = lib.Context(0)
                got = c2.safe_solve_batch(H[None], g[None], pert, method=meth)[0]
                c2.close()
                refstep = g @ O.safe_invert(H.astype(np.float32).astype(np.float64), pert)
                print("    solve alone, method %d: max err %.3e of max |step| %.3e" % (meth, np.abs(got - refstep).max(), np.abs(refstep).max()), flush=True)
