"""Time the batched safe inverse: Cholesky fast path vs Jacobi, k x k, n matrices."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pycmf_amd import _lib
ctx = _lib.Context(0)
rng = np.random.RandomState(0)
for k, n in ((32, 2048), (64, 1024), (128, 512), (256, 256)):
    A = rng.randn(n, k, 2 * k).astype(np.float64)
    H = A @ A.transpose(0, 2, 1) / (2 * k) + 0.3 * np.eye(k)
    for chol in (1, 0):
        ctx.set_option("safe_inverse_cholesky", chol)
        ctx.safe_invert_batch(H[:2], 0.2)
        t0 = time.time(); out = ctx.safe_invert_batch(H, 0.2); dt = time.time() - t0
        ctx.kernel_timing(True); ctx.kernel_timing_reset()
        out = ctx.safe_invert_batch(H, 0.2)
        ms, cnt, _ = ctx.kernel_time("eigen"); ctx.kernel_timing(False)
        ref = np.linalg.inv(H[0])
        err = np.abs(out[0] - ref).max() / np.abs(ref).max()
        waves = (n + 255) // 256
        print("k=%3d n=%4d chol=%d: kernel %.2f ms total -> %.3f ms per matrix-per-CU-slot, err %.1e" % (k, n, chol, ms, ms / waves, err), flush=True)
ctx.close()
