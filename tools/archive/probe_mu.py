"""Quick perf probe: time cmf_mu_step at a given shape with per-kernel-class timing."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycmf_amd import _lib

def run(m, d, p, k, steps=5, warm=2):
    ctx = _lib.Context(0)
    ctx.set_problem(m, d, p, k)
    ctx.fill_data_synthetic(0, 42); ctx.fill_data_synthetic(1, 43)
    sc = (0.8 / k) ** 0.5
    for w in range(3):
        ctx.fill_factor_synthetic(w, 100 + w, 0, sc)
    for _ in range(warm):
        ctx.mu_step(0.0, 0.0, 7)
    ctx.sync()
    t0 = time.time()
    for _ in range(steps):
        ctx.mu_step(0.0, 0.0, 7)
    ctx.sync()
    dt = (time.time() - t0) / steps
    flops = 4.0 * k * d * (m + p) + 4.0 * k * k * (m + d + p)
    print("m=%d d=%d p=%d k=%d: %.3f ms/iter  %.1f TF/s (algorithmic)  %.2f it/s" % (m, d, p, k, dt * 1e3, flops / dt / 1e12, 1 / dt))
    ctx.kernel_timing(True)
    ctx.kernel_timing_reset()
    for _ in range(steps):
        ctx.mu_step(0.0, 0.0, 7)
    for cls in ("gemm_nn", "gemm_tn", "gemm_small", "gemm_nt", "elementwise"):
        ms, n, fl = ctx.kernel_time(cls)
        print("   %-12s %9.3f ms/iter  (%d launches/iter)  %.1f TF/s" % (cls, ms / steps, n // steps, fl / max(ms, 1e-9) / 1e9))
    ex2, ey2 = ctx.residual_sq()
    print("   residual^2:", ex2, ey2)
    ctx.close()

if __name__ == "__main__":
    shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [(16384, 8192, 4096, 128)]
    for s in shapes:
        run(*s)
