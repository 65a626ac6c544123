"""A/B of the staging schedule of the NT (error metric) GEMM: residual_sq at a given shape.
usage: python tools/ab_nt.py m,d,p,k"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycmf_amd import _lib
m, d, p, k = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "65536,65536,65536,256").split(","))
ctx = _lib.Context(0)
ctx.set_problem(m, d, p, k)
ctx.fill_data_synthetic(0, 42); ctx.fill_data_synthetic(1, 43)
sc = (0.8 / k) ** 0.5
for w in range(3):
    ctx.fill_factor_synthetic(w, 100 + w, 0, sc)
for link in ("linear", "logit"):
    for v in (0, 4, 0, 4):
        ctx.set_option("gemm_pipe_nt", v)
        r = ctx.residual_sq(link, link); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            r = ctx.residual_sq(link, link)
        ctx.sync(); dt = (time.perf_counter() - t0) / 3
        print("%s pipe_nt=%d: residual_sq %.2f ms (%.1f TF/s)  values %.6e %.6e" % (link, v, dt * 1e3, 2.0 * k * d * (m + p) / dt / 1e12, r[0], r[1]))
ctx.close()
