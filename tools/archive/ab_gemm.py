"""A/B the GEMM staging schedules (gemm_pipe 0/1/2) interleaved in ONE process on one device.
usage: python tools/ab_gemm.py m,d,p,k [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycmf_amd import _lib

m, d, p, k = (int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "32768,32768,32768,256").split(","))
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
variants = [int(v) for v in (sys.argv[3].split(",") if len(sys.argv) > 3 else "0,1,2".split(","))]
ctx = _lib.Context(0)
ctx.set_problem(m, d, p, k)
ctx.fill_data_synthetic(0, 42); ctx.fill_data_synthetic(1, 43)
sc = (0.8 / k) ** 0.5
for w in range(3):
    ctx.fill_factor_synthetic(w, 100 + w, 0, sc)
for v in variants:
    ctx.set_option("gemm_pipe", v)
    ctx.mu_step(0.0, 0.0, 7)
ctx.sync()
res = {v: [] for v in variants}
ctx.kernel_timing(True)
for r in range(rounds):
    for v in variants:
        ctx.set_option("gemm_pipe", v)
        ctx.kernel_timing_reset()
        for _ in range(3):
            ctx.mu_step(0.0, 0.0, 7)
        nn = ctx.kernel_time("gemm_nn"); tn = ctx.kernel_time("gemm_tn")
        res[v].append((nn[0] / nn[1], nn[2] / nn[0] / 1e9, tn[0] / tn[1], tn[2] / tn[0] / 1e9))
for v in variants:
    a = res[v]
    med = lambda i: sorted(x[i] for x in a)[len(a) // 2]
    best = lambda i: max(x[i] for x in a)
    print("pipe %d: NN %.3f ms/launch %.1f TF (best %.1f) | TN %.3f ms/launch %.1f TF (best %.1f)" %
          (v, med(0), med(1), best(1), med(2), med(3), best(3)))
ctx.close()
