"""Debug: unsharded MU step vs two-shard partial form at a given size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pycmf_amd import _lib as lib
from pycmf_amd.sharded import shard_bounds
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
k = 256
m = d = p = n
def synth(r0=0, c0=0, rows=None, cols=None, opts=()):
    rows = m if rows is None else rows; cols = p if cols is None else cols
    ctx = lib.Context(0)
    for o, v in opts: ctx.set_option(o, v)
    ctx.set_problem(rows, d, cols, k)
    ctx.fill_data_synthetic(0, 42, r0, 0); ctx.fill_data_synthetic(1, 43, 0, c0)
    sc = (0.7979 / k) ** 0.5
    ctx.fill_factor_synthetic(0, 101, r0, sc); ctx.fill_factor_synthetic(1, 102, 0, sc); ctx.fill_factor_synthetic(2, 103, c0, sc)
    return ctx
res = {}
for name, opts in (("fused", ()), ("unfused", (("fused_mu_update", 0),)), ("fused_again", ())):
    c = synth(opts=opts); c.mu_step(0.0, 0.0, 7); res[name] = [c.get_factor(w) for w in range(3)]; c.close()
    print(name, [float(np.abs(f).max()) for f in res[name]], flush=True)
for a in ("unfused", "fused_again"):
    print("fused vs", a, [float(np.abs(x - y).max() / np.abs(y).max()) for x, y in zip(res["fused"], res[a])], flush=True)
shards = []
for r in range(2):
    r0, r1 = shard_bounds(m, 2, r); c0, c1 = shard_bounds(p, 2, r)
    sc = synth(r0, c0, r1 - r0, c1 - c0)
    buf = torch.zeros(sc.v_buf_elems(), dtype=torch.float32, device="cuda:0")
    sc.mu_v_partials(buf.data_ptr()); shards.append((sc, buf, r0, r1, c0, c1))
for sc, *_ in shards: sc.sync()
total = shards[0][1] + shards[1][1]
torch.cuda.synchronize()
print("partials finite", bool(torch.isfinite(total).all()), float(total.abs().max()))
for sc, buf, r0, r1, c0, c1 in shards:
    buf.copy_(total); torch.cuda.synchronize()
    sc.mu_v_apply(buf.data_ptr(), 0.0, 0.0); sc.mu_uz_update(0.0, 0.0, 7); sc.sync()
    for nm in ("fused", "unfused"):
        f = res[nm]
        print("shard vs", nm, float(np.abs(sc.get_factor(1) - f[1]).max() / np.abs(f[1]).max()),
              float(np.abs(sc.get_factor(0) - f[0][r0:r1]).max() / np.abs(f[0]).max()),
              float(np.abs(sc.get_factor(2) - f[2][c0:c1]).max() / np.abs(f[2]).max()), flush=True)
    sc.close()
