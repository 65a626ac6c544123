"""C2 anatomy: data-pass time against the split-K factor, and the BN = 128 kernel on a shape with no split."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycmf_amd import _lib

def run(m, d, p, k, split, steps=6, **opts):
    ctx = _lib.Context(0)
    ctx.set_option("gemm_split", split)
    for name, val in opts.items():
        ctx.set_option(name, val)
    ctx.set_problem(m, d, p, k)
    ctx.fill_data_synthetic(0, 42); ctx.fill_data_synthetic(1, 43)
    for w in range(3): ctx.fill_factor_synthetic(w, 100 + w, 0, (0.8 / k) ** 0.5)
    for _ in range(2): ctx.mu_step(0, 0, 7)
    ctx.sync()
    import time
    t0 = time.time()
    for _ in range(steps): ctx.mu_step(0, 0, 7)
    ctx.sync(); wall = (time.time() - t0) / steps * 1e3
    ctx.kernel_timing(True); ctx.kernel_timing_reset()
    for _ in range(steps): ctx.mu_step(0, 0, 7)
    out = []
    for cls in ("gemm_nn", "gemm_tn", "gemm_small", "elementwise"):
        ms, n, fl = ctx.kernel_time(cls)
        out.append("%s %.3f ms (%d) %.0f TF" % (cls, ms / steps, n // steps, fl / max(ms, 1e-9) / 1e9))
    print("%s m=%d d=%d p=%d k=%d split=%d: wall %.3f ms/iter | %s" % (opts, m, d, p, k, split, wall, " | ".join(out)), flush=True)
    ctx.close()

for rep in range(2):
    run(16384, 8192, 4096, 128, -1)
    run(16384, 8192, 4096, 128, -1, small_tile_update=0)
run(4096, 2048, 1024, 64, -1)
run(4096, 2048, 1024, 64, -1, small_tile_update=0)
