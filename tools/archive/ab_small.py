"""A/B the staging schedule of the factor-side products (gemm_pipe_small 0 vs 4) on MU / Newton steps."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycmf_amd import _lib
for (m, d, p, k) in [(16384, 8192, 4096, 128), (1000000, 2048, 64, 256), (65536, 65536, 4096, 256)]:
    ctx = _lib.Context(0)
    ctx.set_problem(m, d, p, k)
    ctx.fill_data_synthetic(0, 42); ctx.fill_data_synthetic(1, 43)
    for w in range(3): ctx.fill_factor_synthetic(w, 100 + w, 0, (0.8 / k) ** 0.5)
    for v in (0, 4): 
        ctx.set_option("gemm_pipe_small", v); ctx.mu_step(0, 0, 7); ctx.newton_step(0.5, 0, 0.1, "linear", "linear", 0, 7, 0.2, 1.0)
    for rnd in range(3):
        for v in (0, 4):
            ctx.set_option("gemm_pipe_small", v)
            ctx.kernel_timing(True); ctx.kernel_timing_reset()
            for _ in range(5): ctx.mu_step(0, 0, 7); ctx.newton_step(0.5, 0, 0.1, "linear", "linear", 0, 7, 0.2, 1.0)
            ms, n, fl = ctx.kernel_time("gemm_small"); ctx.kernel_timing(False)
            if rnd == 2: print("m=%d d=%d p=%d k=%d pipe_small=%d: gemm_small %.3f ms per (mu+newton) step, %.1f TF/s" % (m, d, p, k, v, ms / 5, fl / ms / 1e9))
    ctx.close()
