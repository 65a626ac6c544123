"""Per-iteration distance between the device iterates and the float64 oracle on a clamped, ill-conditioned linear Newton
problem, with the float64 shared-Hessian treatment on and off."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from oracle import cmf_oracle as O
from pycmf_amd import _lib
rng = np.random.RandomState(3)
m, d, p, k = 800, 600, 200, 64
nn = int(sys.argv[1]) if len(sys.argv) > 1 else 1
Ut, Vt, Zt = np.abs(rng.randn(m, 5)), np.abs(rng.randn(d, 5)), np.abs(rng.randn(p, 5))
X, Y = Ut @ Vt.T, Vt @ Zt.T
sc = np.sqrt(X.mean() / k)
U0, V0, Z0 = sc * np.abs(rng.randn(m, k)), sc * np.abs(rng.randn(d, k)), sc * np.abs(rng.randn(p, k))
ctxs = {}
for f64 in (1, 0):
    c = _lib.Context(0); c.set_option("shared_hessian_f64", f64); c.set_problem(m, d, p, k)
    c.set_data(0, X); c.set_data(1, Y)
    for w, F in enumerate((U0, V0, Z0)): c.set_factor(w, F)
    ctxs[f64] = c
Uo, Vo, Zo = U0.copy(), V0.copy(), Z0.copy()
for it in range(4):
    # sweep by sweep: U, Z, V
    for mask, name in ((1, "U"), (4, "Z"), (2, "V")):
        O.newton_update_step(X, Y, Uo, Vo, Zo, 0.5, 0.0, 0.0, "linear", "linear", bool(nn), bool(nn), bool(nn), 1.0, 0.2,
                             update_U=mask == 1, update_V=mask == 2, update_Z=mask == 4)
        ref = {1: Uo, 4: Zo, 2: Vo}[mask]
        H = {1: 0.5 * Vo.T @ Vo, 4: 0.5 * Vo.T @ Vo}.get(mask)
        out = []
        for f64 in (1, 0):
            c = ctxs[f64]
            c.newton_step(0.5, 0.0, 0.0, "linear", "linear", 7 if nn else 0, mask, 0.2, 1.0)
            got = c.get_factor({1: 0, 4: 2, 2: 1}[mask])
            out.append(np.abs(got - ref).max() / np.abs(ref).max())
        print("iter %d sweep %s: rel err f64 %.3e  f32 %.3e   |ref|max %.3g" % (it, name, out[0], out[1], np.abs(ref).max()))
    lam = np.linalg.eigvalsh(0.5 * Vo.T @ Vo)
    print("   eig(0.5 V^T V): min %.3g max %.3g, below pert: %d" % (lam.min(), lam.max(), (lam < 0.2).sum()))
