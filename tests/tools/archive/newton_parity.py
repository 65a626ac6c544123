import sys, time, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
from oracle import cmf_oracle as O
from pycmf_amd.solver_shell import HipNewtonSolver
rng = np.random.RandomState(0)
for (m, d, p, k, xl, yl, l2, nn) in [(1200, 900, 300, 64, "linear", "linear", 0.0, True), (1200, 900, 300, 64, "linear", "linear", 0.0, False),
                                      (600, 500, 200, 48, "linear", "logit", 0.0, False), (600, 500, 200, 48, "logit", "logit", 0.01, False)]:
    Ut, Vt, Zt = np.abs(rng.randn(m, 8)), np.abs(rng.randn(d, 8)), np.abs(rng.randn(p, 8))
    X = Ut @ Vt.T + 0.1 * np.abs(rng.randn(m, d)); Y = Vt @ Zt.T + 0.1 * np.abs(rng.randn(d, p))
    if xl == "logit": X = 1 / (1 + np.exp(-(X - X.mean()) / X.std()))
    if yl == "logit": Y = 1 / (1 + np.exp(-(Y - Y.mean()) / Y.std()))
    sc = np.sqrt(np.abs(X).mean() / k)
    U0, V0, Z0 = sc * np.abs(rng.randn(m, k)), sc * np.abs(rng.randn(d, k)), sc * np.abs(rng.randn(p, k))
    kw = dict(alpha=0.5, l2_reg=l2, x_link=xl, y_link=yl, U_non_negative=nn, V_non_negative=nn, Z_non_negative=nn, max_iter=8, tol=0)
    g = HipNewtonSolver(**kw); Ug, Vg, Zg = U0.copy(), V0.copy(), Z0.copy()
    t0 = time.time(); g.fit_iterative_update(X, Y, Ug, Vg, Zg); tg = time.time() - t0
    eg = (O.factorization_error(X, Ug, Vg.T, xl), O.factorization_error(Y, Vg, Zg.T, yl)); g.release()
    o = O.OracleSolver("newton", **kw); Uo, Vo, Zo = U0.copy(), V0.copy(), Z0.copy()
    t0 = time.time(); o.fit_iterative_update(X, Y, Uo, Vo, Zo); to = time.time() - t0
    eo = (O.factorization_error(X, Uo, Vo.T, xl), O.factorization_error(Y, Vo, Zo.T, yl))
    print((m, d, p, k, xl, yl, l2, nn), "gpu err %.6f %.6f (%.2fs) | oracle err %.6f %.6f (%.1fs) | rel diff %.2e %.2e" % (eg[0], eg[1], tg, eo[0], eo[1], to, abs(eg[0]-eo[0])/eo[0], abs(eg[1]-eo[1])/eo[1]), flush=True)
