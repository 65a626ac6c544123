"""Cost of the float64 refinement per row: every row of a per-row Newton sweep is forced through it (refine_rows_ratio = refine_rows_cond = 0)
and the sweep is timed with the batched form (cmf_refine64.hip.h) and with the one-row-at-a-time form of round 3.

    python tools/refine_timing.py [m,d,p,k] [ratio]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycmf_amd import _lib

m, d, p, k = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2048,4096,512,256").split(","))
ratio = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
res = {}
for batched in (1, 0):
    for pert, tag in ((0.2, "pert 0.2: threshold test passes, plain solves"), (30.0, "pert 30: the spectral clamp acts on every row")):
        ctx = _lib.Context(0)
        ctx.set_option("refine_rows_batched", batched)
        ctx.set_problem(m, d, p, k)
        ctx.fill_data_synthetic(0, 42, 0, 0)
        ctx.fill_data_synthetic(1, 43, 0, 0, 1)
        sc = (0.7979 / k) ** 0.5
        for w, s in ((0, 101), (1, 102), (2, 103)):
            ctx.fill_factor_synthetic(w, s, 0, sc)
        l2 = 0.1
        times = {}
        ctx.newton_step_device_sampled(0.5, 0.0, l2, "linear", "logit", 0, 1, pert, ratio, 6)   # warm-up: workspaces, code objects
        ctx.set_option("refine_rows_ratio", 0); ctx.set_option("refine_rows_cond", 0)
        ctx.newton_step_device_sampled(0.5, 0.0, l2, "linear", "logit", 0, 1, pert, ratio, 6)
        ctx.set_option("refine_rows_ratio", 1 << 40); ctx.set_option("refine_rows_cond", 1 << 40)
        for force in (0, 1):
            if force:
                ctx.set_option("refine_rows_ratio", 0)
                ctx.set_option("refine_rows_cond", 0)
            ctx.newton_clamp_stats(reset=True)
            ctx.sync()
            t0 = time.perf_counter()
            ctx.newton_step_device_sampled(0.5, 0.0, l2, "linear", "logit", 0, 1, pert, ratio, 7)   # the U sweep only: m rows, int(d * ratio) samples each
            ctx.sync()
            times[force] = time.perf_counter() - t0
            st = ctx.newton_clamp_stats(full=True)
        rows = st[2]
        per = (times[1] - times[0]) / max(rows, 1) * 1e6
        print("batched %d, %s: %d rows of %d refined, sweep %.1f ms -> %.1f ms: %.1f us per refined row (k = %d, %d samples per row); clamped rows float32: %d"
              % (batched, tag, rows, m, times[0] * 1e3, times[1] * 1e3, per, k, int(d * ratio), st[0]))
        ctx.close()
