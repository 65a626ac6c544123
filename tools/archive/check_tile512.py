import sys, numpy as np
sys.path.insert(0, "/root/repo")
from pycmf_amd import _lib
m, d, p, k = 2048, 1536, 1024, 100
rng = np.random.RandomState(0)
X, Y = np.abs(rng.randn(m, d)), np.abs(rng.randn(d, p))
F = [np.abs(rng.randn(n, k)) * 0.3 for n in (m, d, p)]
out = {}
for t in (0, 1):
    ctx = _lib.Context(0); ctx.set_option("gemm_tile512", t)
    ctx.set_problem(m, d, p, k); ctx.set_data(0, X); ctx.set_data(1, Y)
    for w, f in enumerate(F): ctx.set_factor(w, f)
    for _ in range(3): ctx.mu_step(0.0, 0.0, 7)
    out[t] = [ctx.get_factor(w) for w in range(3)]; ctx.close()
for a, b in zip(out[0], out[1]):
    print("max rel diff", np.abs(a - b).max() / np.abs(a).max())
