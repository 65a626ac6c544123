"""Shared-Hessian safe inverse when eigenvalues dip under the perturbation (Jacobi path) at several k."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import numpy as np
from pycmf_amd import _lib
from oracle import cmf_oracle as O
rng = np.random.RandomState(0)
for k in (64, 128, 256):
    m, d, p = 4096, k // 2, 64       # d < k: V^T V is rank deficient -> eigenvalues l2 < pert -> Jacobi
    X, Y = np.abs(rng.randn(m, d)), np.abs(rng.randn(d, p))
    U, V, Z = 0.1 * rng.randn(m, k), 0.1 * rng.randn(d, k), 0.1 * rng.randn(p, k)
    ctx = _lib.Context(0)
    ctx.set_problem(m, d, p, k)
    ctx.set_data(0, X); ctx.set_data(1, Y)
    for w, F in enumerate((U, V, Z)): ctx.set_factor(w, F)
    ctx.newton_step(0.5, 0.0, 0.05, "linear", "linear", 0, 1, 0.2, 1.0)   # U sweep only
    ctx.sync(); t0 = time.time()
    for w, F in enumerate((U, V, Z)): ctx.set_factor(w, F)
    ctx.newton_step(0.5, 0.0, 0.05, "linear", "linear", 0, 1, 0.2, 1.0)
    ctx.sync(); dt = time.time() - t0
    got = ctx.get_factor(0)
    Ur = U.copy(); O.newton_sweep_U(Ur, V, X, 0.5, 0.0, 0.05, "linear", False, 1.0, 0.2)
    print("k=%d: U sweep with clamped shared Hessian %.1f ms, max rel err vs oracle %.2e" % (k, dt * 1e3, np.abs(got - Ur).max() / np.abs(Ur).max()), flush=True)
    ctx.close()
