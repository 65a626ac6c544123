"""C5 anatomy: native-CSR linear Newton at the BASELINE shape, plain row kernel vs column-blocked SpMM."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.sparse as sp
from pycmf_amd import _lib
m, d, p, k, npr = 1000000, 100000, 64, 256, 100
if len(sys.argv) > 1: m = int(sys.argv[1])
rng = np.random.default_rng(42)
X = sp.csr_matrix((np.ones(m * npr), rng.integers(0, d, size=m * npr, dtype=np.int32), np.arange(0, m * npr + 1, npr, dtype=np.int64)), shape=(m, d))
variants = [(1, 0, 0), (1, 1024, 0), (1, 4096, 0), (1, 0, 8192), (1, 0, 32768)] if os.environ.get("SWEEP") else [(1, 0, 0)]
for blocked, bcols, stretch in variants:
    ctx = _lib.Context(0)
    ctx.set_option("sparse_mode", 2); ctx.set_option("spmm_blocked", blocked)
    ctx.set_option("spmm_block_cols", bcols); ctx.set_option("spmm_stretch", stretch)
    print("variant block_cols=%d stretch=%d" % (bcols, stretch))
    ctx.set_problem(m, d, p, k)
    t0 = time.time(); ctx.set_data(0, X); t_up = time.time() - t0
    ctx.fill_data_synthetic(1, 43, 0, 0)
    sc = (npr / d / k) ** 0.5
    for w in range(3): ctx.fill_factor_synthetic(w, 101 + w, 0, sc)
    for _ in range(2): ctx.newton_step(0.5, 0.0, 0.1, "linear", "linear", 0, 7, 0.2, 1.0)
    ctx.sync(); t0 = time.time()
    for _ in range(5): ctx.newton_step(0.5, 0.0, 0.1, "linear", "linear", 0, 7, 0.2, 1.0)
    ctx.sync(); wall = (time.time() - t0) / 5 * 1e3
    ctx.kernel_timing(True); ctx.kernel_timing_reset()
    for _ in range(5): ctx.newton_step(0.5, 0.0, 0.1, "linear", "linear", 0, 7, 0.2, 1.0)
    parts = []
    for cls in ("spmm", "gemm_nn", "gemm_tn", "gemm_small", "eigen", "elementwise"):
        ms, n, fl = ctx.kernel_time(cls)
        parts.append("%s %.2f ms (%d)" % (cls, ms / 5, n // 5))
    for mask, name in ((1, "U sweep: X V"), (2, "V sweep: X^T U")):
        ctx.kernel_timing_reset()
        for _ in range(5): ctx.newton_step(0.5, 0.0, 0.1, "linear", "linear", 0, mask, 0.2, 1.0)
        ms, n, fl = ctx.kernel_time("spmm")
        print("   blocked=%d %s: spmm %.2f ms per launch (%d launches)" % (blocked, name, ms / max(n, 1), n), flush=True)
    print("blocked=%d: upload %.1f s, wall %.2f ms/iter = %.1f it/s | %s | resid %s" % (blocked, t_up, wall, 1e3 / wall, " | ".join(parts), ctx.residual_sq()), flush=True)
    ctx.close()
