"""Where the wall time of a small end-to-end `CMF.fit_transform` goes (the reference's own benchmark shape, 2000 x 150 / 150 x 10,
k = 10, 10 iterations): context creation, upload, the solver loop, download.  usage: python tools/fit_phase_times.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycmf_amd import CMF, _lib

rng = np.random.mtrand.RandomState(42)
X, Y = np.abs(rng.randn(2000, 150)), np.abs(rng.randn(150, 10))
for solver in ("mu", "newton"):
    for rep in range(4):
        t0 = time.perf_counter()
        model = CMF(n_components=10, random_state=42, max_iter=10, solver=solver)
        model.fit_transform(X, Y)
        t1 = time.perf_counter()
        print(solver, "fit_transform %.2f ms" % ((t1 - t0) * 1e3))
    U, V, Z = np.abs(rng.randn(2000, 10)), np.abs(rng.randn(150, 10)), np.abs(rng.randn(10, 10))
    for rep in range(3):
        t = [time.perf_counter()]
        ctx = _lib.Context(0); t.append(time.perf_counter())
        ctx.set_problem(2000, 150, 10, 10); t.append(time.perf_counter())
        ctx.set_data(0, X); ctx.set_data(1, Y); t.append(time.perf_counter())
        for w, F in enumerate((U, V, Z)):
            ctx.set_factor(w, F)
        t.append(time.perf_counter())
        for _ in range(10):
            if solver == "mu":
                ctx.mu_step(0.0, 0.0, 7)
            else:
                ctx.newton_step(0.5, 0.0, 0.0, "linear", "linear", 7, 7, 0.2, 1.0)
        ctx.sync(); t.append(time.perf_counter())
        ex = ctx.residual_sq(); t.append(time.perf_counter())
        got = [ctx.get_factor(w) for w in range(3)]; t.append(time.perf_counter())
        ctx.close(); t.append(time.perf_counter())
        names = ("ctx_create", "set_problem", "upload X,Y", "upload factors", "10 steps", "residual", "download", "close")
        print("  " + solver + "  " + "  ".join("%s %.2f" % (n, (b - a) * 1e3) for n, a, b in zip(names, t, t[1:])) + "  (ms)")
