"""c5l (CSR X, y logit, l1=2, l2=5, U/V non-negative): residual trace per iteration, full size on the GPU and a mid-size
case beside the float64 oracle."""
import sys, os
import numpy as np, scipy.sparse as sp
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from pycmf_amd import _lib

def make(m, d, p, k, npr, seed=42):
    rng = np.random.default_rng(seed)
    X = sp.csr_matrix((np.ones(m * npr), rng.integers(0, d, size=m * npr, dtype=np.int32), np.arange(0, m * npr + 1, npr, dtype=np.int64)), shape=(m, d))
    X.sum_duplicates()      # the upload merges repeated (row, column) pairs; the oracle's error formula must see the same matrix
    return X

def trace(m, d, p, k, npr, iters, oracle=False, l1=2.0, l2=5.0):
    X = make(m, d, p, k, npr)
    ctx = _lib.Context(0)
    ctx.set_option("sparse_mode", 2)
    ctx.set_problem(m, d, p, k)
    ctx.set_data(0, X)
    ctx.fill_data_synthetic(1, 43, 0, 0, 2, 0.1)
    scale = (npr / d / k) ** 0.5
    for w, sd in ((0, 101), (1, 102), (2, 103)):
        ctx.fill_factor_synthetic(w, sd, 0, scale)
    x2, y2 = ctx.data_sq()
    if oracle:
        from oracle import cmf_oracle as O
        Y = ctx.get_data(1).astype(np.float64)
        U, V, Z = (ctx.get_factor(w) for w in range(3))
    for it in range(iters):
        ctx.newton_step_device_sampled(0.5, l1, l2, "linear", "logit", 3, 7, 0.2, 1.0, 1000 + it)
        ex2, ey2 = ctx.residual_sq("linear", "logit")
        F = [ctx.get_factor(w) for w in range(3)]
        line = "it %d gpu rel res x %.6f y %.6f  max|U| %.3g |V| %.3g |Z| %.3g nnzU %.3f nnzV %.3f" % (
            it, (ex2 / x2) ** 0.5, (ey2 / y2) ** 0.5, np.abs(F[0]).max(), np.abs(F[1]).max(), np.abs(F[2]).max(), (F[0] != 0).mean(), (F[1] != 0).mean())
        if oracle:
            O.newton_update_step(X, Y, U, V, Z, 0.5, l1, l2, "linear", "logit", True, True, False, 1.0, 0.2)
            ox = O.factorization_error(X, U, V.T, "linear") / np.sqrt(x2); oy = O.factorization_error(Y, V, Z.T, "logit") / np.sqrt(y2)
            line += "   oracle x %.6f y %.6f  dU %.2e dV %.2e dZ %.2e" % (ox, oy, np.abs(F[0] - U).max(), np.abs(F[1] - V).max(), np.abs(F[2] - Z).max())
        print(line, flush=True)
    ctx.close()

if __name__ == "__main__":
    if "--full" in sys.argv:
        trace(1000000, 100000, 64, 256, 100, 12)
    else:
        trace(6000, 1500, 64, 32, 30, 10, oracle=True)
