"""C5-shaped probe: CSR X (m x d, nnz_per_row nonzeros per row, values 1.0), dense Y (d x p), k components.
usage: python tools/probe_sparse.py m d p k nnz_per_row"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import scipy.sparse as sp
from pycmf_amd import _lib

m, d, p, k, npr = (int(x) for x in sys.argv[1:6]) if len(sys.argv) > 5 else (200000, 100000, 64, 256, 100)
rng = np.random.default_rng(0)
t0 = time.time()
indices = rng.integers(0, d, size=m * npr, dtype=np.int32)
indptr = np.arange(0, m * npr + 1, npr, dtype=np.int64)
X = sp.csr_matrix((np.ones(m * npr), indices, indptr), shape=(m, d))
Y = (rng.random((d, p)) < 0.1).astype(np.float64)
print("host build %.1fs nnz=%d" % (time.time() - t0, X.nnz), flush=True)
ctx = _lib.Context(0)
ctx.set_option("sparse_mode", 2)
ctx.set_problem(m, d, p, k)
t0 = time.time(); ctx.set_data(0, X); ctx.set_data(1, Y); print("upload %.1fs" % (time.time() - t0), flush=True)
for w in range(3):
    ctx.fill_factor_synthetic(w, 100 + w, 0, (0.05 / k) ** 0.5)
nnz = X.nnz
kp = ctx.geometry()[3]
def timed(label, fn, n=5):
    fn(); ctx.sync()
    ctx.kernel_timing(True); ctx.kernel_timing_reset()
    t0 = time.time()
    for _ in range(n): fn()
    ctx.sync(); dt = (time.time() - t0) / n
    ms, cnt, fl = ctx.kernel_time("spmm")
    ctx.kernel_timing(False)
    gather = nnz * (kp * 4 + 8.0)
    print("%-28s %8.2f ms/iter | spmm class %.2f ms/iter over %d launches/iter -> %.2f TB/s gathered (%.1f GB per SpMM)" %
          (label, dt * 1e3, ms / n, cnt // n, (cnt / n) * gather / (ms / n * 1e-3) / 1e12 if ms > 0 else 0, gather / 1e9), flush=True)
timed("mu_step", lambda: ctx.mu_step(0.0, 0.0, 7))
timed("newton_step linear", lambda: ctx.newton_step(0.5, 0.0, 0.1, "linear", "linear", 0, 7, 0.2, 1.0))
timed("residual_sq linear", lambda: ctx.residual_sq("linear", "linear"))
print("residuals", ctx.residual_sq("linear", "linear"), ctx.data_sq())
ctx.close()
