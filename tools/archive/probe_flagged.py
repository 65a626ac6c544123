"""Cost of the flagged-row treatment at k_pad = 256: Newton-Schulz spectral clamp vs Jacobi.
usage: python tools/probe_flagged.py [rows] [samples] [k]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycmf_amd import _lib
m = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
d = int(sys.argv[2]) if len(sys.argv) > 2 else 128      # samples per U row < k: every U Hessian is rank-deficient
p = 64
k = int(sys.argv[3]) if len(sys.argv) > 3 else 256
rng = np.random.RandomState(0)
X, Y = rng.rand(m, d), rng.rand(d, p)
U0, V0, Z0 = 0.1 * rng.randn(m, k), 0.1 * rng.randn(d, k), 0.1 * rng.randn(p, k)
res = {}
for ns in (1, 0):
    ctx = _lib.Context(0)
    ctx.set_option("newton_schulz", ns)
    ctx.set_problem(m, d, p, k)
    ctx.set_data(0, X); ctx.set_data(1, Y)
    for w, F in enumerate((U0, V0, Z0)):
        ctx.set_factor(w, F)
    ctx.newton_step(0.5, 0.0, 0.05, "logit", "logit", 0, 1, 0.2, 1.0)   # warm-up (U sweep only)
    for w, F in enumerate((U0, V0, Z0)):
        ctx.set_factor(w, F)
    ctx.sync(); t0 = time.perf_counter()
    ctx.newton_step(0.5, 0.0, 0.05, "logit", "logit", 0, 1, 0.2, 1.0)
    ctx.sync(); dt = time.perf_counter() - t0
    res[ns] = ctx.get_factor(0)
    print("newton_schulz=%d: U sweep of %d flagged rows in %.1f ms (%.1f us per row)" % (ns, m, dt * 1e3, dt * 1e6 / m))
    ctx.close()
dd = np.abs(res[1] - res[0]).max() / np.abs(res[0]).max()
print("max |U_ns - U_jacobi| / max|U| = %.2e" % dd)
