"""Launch-bound regime: time a few hundred iterations on tiny / small problems with and without graph replay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycmf_amd import _lib
for (m, d, p, k) in [(6, 5, 6, 5), (2000, 150, 10, 10), (4096, 2048, 1024, 128), (16384, 8192, 4096, 128)]:
    for solver in ("mu", "newton"):
        res = {}
        for graph in (0, 1):
            ctx = _lib.Context(0)
            ctx.set_option("graph", graph)
            ctx.set_problem(m, d, p, k)
            ctx.fill_data_synthetic(0, 42); ctx.fill_data_synthetic(1, 43)
            for w in range(3):
                ctx.fill_factor_synthetic(w, 100 + w, 0, (0.8 / k) ** 0.5)
            step = (lambda: ctx.mu_step(0.0, 0.0, 7)) if solver == "mu" else (lambda: ctx.newton_step(0.5, 0.0, 0.1, "linear", "linear", 7, 7, 0.2, 1.0))
            for _ in range(5): step()
            ctx.sync()
            n = 200 if m < 10000 else 50
            t0 = time.time()
            for _ in range(n): step()
            ctx.sync()
            res[graph] = (time.time() - t0) / n * 1e6
            r = ctx.residual_sq()
            ctx.close()
        print("%-7s m=%d d=%d p=%d k=%d: eager %.1f us/iter, graph %.1f us/iter (x%.2f)  res %s" % (solver, m, d, p, k, res[0], res[1], res[0] / res[1], r), flush=True)
