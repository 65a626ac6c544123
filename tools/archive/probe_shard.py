"""Per-rank compute time of the strong-scaled C4 problem at world sizes 1,2,4,8, measured on ONE GPU
(the rank-0 shard only; the all-reduce itself is not included)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pycmf_amd import _lib
from pycmf_amd.sharded import shard_bounds
m = d = p = 65536; k = 256
base = None
for world in (1, 2, 4, 8):
    r0, r1 = shard_bounds(m, world, 0); c0, c1 = shard_bounds(p, world, 0)
    ctx = _lib.Context(0)
    ctx.set_problem(r1 - r0, d, c1 - c0, k)
    ctx.fill_data_synthetic(0, 42, r0, 0); ctx.fill_data_synthetic(1, 43, 0, c0)
    for w in range(3):
        ctx.fill_factor_synthetic(w, 100 + w, 0, (0.8 / k) ** 0.5)
    buf = torch.zeros(ctx.v_buf_elems(), dtype=torch.float32, device="cuda:0")
    def step():
        ctx.mu_v_partials(buf.data_ptr()); ctx.mu_v_apply(buf.data_ptr(), 0.0, 0.0); ctx.mu_uz_update(0.0, 0.0, 7)
    for _ in range(3): step()
    ctx.sync(); t0 = time.time()
    n = 10
    for _ in range(n): step()
    ctx.sync(); dt = (time.time() - t0) / n
    ctx.kernel_timing(True); ctx.kernel_timing_reset()
    for _ in range(3): step()
    cls = {c: ctx.kernel_time(c)[0] / 3 for c in ("gemm_nn", "gemm_tn", "gemm_small", "elementwise")}
    ctx.kernel_timing(False)
    base = base or dt
    print("world %d: rank-0 compute %.2f ms/iter (x%.2f vs world 1)  %s" % (world, dt * 1e3, base / dt, {a: round(b, 2) for a, b in cls.items()}), flush=True)
    ctx.close(); del buf
