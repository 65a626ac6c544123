import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycmf_amd import _lib
m, d, p, k = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "16384,8192,4096,128").split(","))
ctx = _lib.Context(0)
ctx.set_problem(m, d, p, k)
ctx.fill_data_synthetic(0, 42); ctx.fill_data_synthetic(1, 43)
for w in range(3):
    ctx.fill_factor_synthetic(w, 100 + w, 0, (0.8 / k) ** 0.5)
for rnd in range(3):
    for split in (-1, 1, 2, 4, 8, 16, 32):
        ctx.set_option("gemm_split", split)
        for _ in range(3): ctx.mu_step(0.0, 0.0, 7)
        ctx.sync(); t0 = time.time()
        for _ in range(30): ctx.mu_step(0.0, 0.0, 7)
        ctx.sync(); dt = (time.time() - t0) / 30
        if rnd == 2: print("split %3d: %.1f us/iter" % (split, dt * 1e6), flush=True)
ctx.close()
