"""Newton probe at a given shape (default: BASELINE configs[2] = C3): sigmoid y-link, stochastic Hessian.
usage: python tools/probe_newton.py m d p k ratio x_link y_link [iters]
Samples are drawn on the host in the reference's order (parity mode)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pycmf_amd import _lib

a = sys.argv[1:]
m, d, p, k = (int(x) for x in a[:4]) if len(a) >= 4 else (32768, 16384, 8192, 256)
ratio = float(a[4]) if len(a) > 4 else 0.5
xl = a[5] if len(a) > 5 else "linear"
yl = a[6] if len(a) > 6 else "logit"
iters = int(a[7]) if len(a) > 7 else 1
sampler = a[8] if len(a) > 8 else "numpy"
row_kernel = int(a[9]) if len(a) > 9 else 1
ctx = _lib.Context(0)
ctx.set_option("row_kernel", row_kernel)
ctx.set_option("row_stagger", int(os.environ.get("ROW_STAGGER", "1")))
ctx.set_option("row_diag", int(os.environ.get("ROW_DIAG", "0")))
ctx.set_option("gemm_arith", int(os.environ.get("GEMM_ARITH", "0")))
ctx.set_option("row_classes", int(os.environ.get("ROW_CLASSES", "-1")))
if os.environ.get("CHOL_DIAG"):
    ctx.set_option("chol_diag", int(os.environ["CHOL_DIAG"]))
ctx.set_problem(m, d, p, k)
ctx.fill_data_synthetic(0, 42); ctx.fill_data_synthetic(1, 43)
sc = (0.8 / k) ** 0.5
for w in range(3):
    ctx.fill_factor_synthetic(w, 100 + w, 0, sc)
np.random.seed(0)
def draw(rows, n):
    s = int(n * ratio); out = np.empty((rows, s), np.int32); ar = np.arange(n)
    for i in range(rows): out[i] = np.random.permutation(ar)[:s]
    return out
for it in range(iters):
    t0 = time.time()
    idx = (None,) * 4
    if ratio < 1 and sampler == "numpy":
        idx = (draw(m, d), draw(p, d), draw(d, m), draw(d, p))
    th = time.time() - t0
    ctx.kernel_timing(True); ctx.kernel_timing_reset()
    t0 = time.time()
    if ratio < 1 and sampler == "device":
        ctx.newton_step_device_sampled(0.5, 0.0, 0.1, xl, yl, 0, 7, 0.2, ratio, 1000 + it)
    else:
        ctx.newton_step(0.5, 0.0, 0.1, xl, yl, 0, 7, 0.2, ratio, *idx)
    ctx.sync(); dt = time.time() - t0
    print("iter %d: host sampling %.1fs, device step %.3fs" % (it, th, dt))
    for cls in ("gemm_nn", "gemm_tn", "gemm_nt", "gemm_small", "rowhess", "eigen", "elementwise"):
        ms, n, fl = ctx.kernel_time(cls)
        print("   %-12s %10.2f ms  %4d launches  %.1f TF/s" % (cls, ms, n, fl / max(ms, 1e-9) / 1e9))
    ctx.kernel_timing(False)
    print("   residual^2", ctx.residual_sq(xl, yl), flush=True)
ctx.close()
