#!/usr/bin/env python3
"""bench.py -- factor-update iterations/s of the CMF MU solver on MI355X.

Metric (BASELINE.json): factor-update iterations/s (and cells/s) at
n_components=256, 65536^2 dense.  One "step" = one full ``update_step``
(V, U, Z multiplicative updates; pycmf/cmf_solvers.py:248-263) over synthetic
non-negative X (m x d), Y (d x p) already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c2|tiny]

N > 1 is launched by torch.distributed.run (one rank per GPU).  The problem is
FIXED (strong scaling): rank g owns rows [g*m/N, (g+1)*m/N) of X/U and the same
fraction of Y's columns / Z's rows; V is replicated and reassembled by one RCCL
all-reduce per iteration (pycmf_amd/sharded.py).

Prints ONE JSON line on rank 0 (contract in the task description) with
``roofline`` (dominant GEMM kernel, HIP events on the launch stream over the
timed region) and ``cpu_baseline`` (CPU oracle, bounded sample, rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    # name: (m, d, p, k, description)
    "c4": (65536, 65536, 65536, 256,
           "BASELINE configs[3]: CMF(n_components=256, solver='mu'), dense 65536x65536 X, "
           "65536x65536 Y, non-negative synthetic"),
    "c2": (16384, 8192, 4096, 128,
           "BASELINE configs[1]: CMF(n_components=128, solver='mu', linear link), dense "
           "16384x8192 X, 8192x4096 Y, non-negative synthetic"),
    "tiny": (2048, 1024, 512, 64, "debug shape"),
}
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense


def mu_algorithmic_flops(m, d, p, k):
    """F_MU = 4 k d (m+p) + 4 k^2 (m+d+p)."""
    return 4.0 * k * d * (m + p) + 4.0 * k * k * (m + d + p)


def cpu_baseline(k, budget_s=20.0):
    """Time the CPU oracle (reference operation order, float64, all BLAS threads) on a
    bounded sample of the same workload and scale to the bench shape by the work ratio."""
    import numpy as np
    from oracle import cmf_oracle as O
    ms = ds = ps = 4096
    rng = np.random.RandomState(42)
    X, Y = np.abs(rng.randn(ms, ds)), np.abs(rng.randn(ds, ps))
    sc = np.sqrt(X.mean() / k)
    U, V, Z = (sc * np.abs(rng.randn(n, k)) for n in (ms, ds, ps))
    O.mu_update_step(X, Y, U, V, Z)  # warm-up
    t0 = time.perf_counter()
    iters = 0
    while True:
        O.mu_update_step(X, Y, U, V, Z)
        iters += 1
        el = time.perf_counter() - t0
        if el > budget_s or (iters >= 3 and el > budget_s / 2):
            break
    try:
        from threadpoolctl import threadpool_info
        threads = max([i.get("num_threads", 1) for i in threadpool_info()] or [os.cpu_count() or 1])
    except Exception:
        threads = os.cpu_count() or 1
    return iters / el, (ms, ds, ps), iters, el, threads


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus %d needs torch.distributed.run with %d ranks" % (args.gpus, args.gpus))
        args.gpus = world

    import torch
    import torch.distributed as dist
    from pycmf_amd import _lib
    from pycmf_amd.sharded import make_torch_sharded_mu, shard_bounds

    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    m, d, p, k, desc = WORKLOADS[args.workload]
    r0, r1 = shard_bounds(m, world, rank)
    c0, c1 = shard_bounds(p, world, rank)

    stream = torch.cuda.Stream(device=device)
    with torch.cuda.stream(stream):
        ctx = _lib.Context(local_rank, stream.cuda_stream)
        ctx.set_problem(r1 - r0, d, c1 - c0, k)
        # X rows [r0,r1), Y columns [c0,c1): values depend only on global coordinates
        ctx.fill_data_synthetic(0, 42, r0, 0)
        ctx.fill_data_synthetic(1, 43, 0, c0)
        scale = (0.7979 / k) ** 0.5  # 'random' init rule: sqrt(mean(|N(0,1)|) / k), pycmf/cmf.py:111
        ctx.fill_factor_synthetic(_lib.CMF_U, 101, r0, scale)
        ctx.fill_factor_synthetic(_lib.CMF_V, 102, 0, scale)
        ctx.fill_factor_synthetic(_lib.CMF_Z, 103, c0, scale)
        drv = make_torch_sharded_mu(ctx, world, device)

        def sync_all():
            torch.cuda.synchronize(device)
            if world > 1:
                dist.barrier()
                torch.cuda.synchronize(device)

        for _ in range(args.warmup):
            drv.step(0.0, 0.0, 7)
        ctx.kernel_timing(True)
        ctx.kernel_timing_reset()
        sync_all()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            drv.step(0.0, 0.0, 7)
        sync_all()
        elapsed = time.perf_counter() - t0
        if world > 1:
            te = torch.tensor([elapsed], dtype=torch.float64, device=device)
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            elapsed = float(te.item())

        classes = {c: ctx.kernel_time(c) for c in ("gemm_nn", "gemm_tn", "gemm_small", "gemm_nt", "elementwise")}
        ctx.kernel_timing(False)
        ex2, ey2 = ctx.residual_sq()
        x2, y2 = ctx.data_sq()

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    its = args.steps / elapsed
    # dominant kernel = GEMM class with the most device time on rank 0
    dom = max(("gemm_nn", "gemm_tn"), key=lambda c: classes[c][0])
    dms, dn, dfl = classes[dom]
    achieved = dfl / (dms * 1e-3) / 1e12 if dms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
    if world == 1 and os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(dom)
        except Exception:
            traffic = None
    out = {
        "metric": "factor-update iterations/s (MU solver: V,U,Z update per iteration)",
        "value": its,
        "unit": "it/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "cells_per_s": (float(m) * d + float(d) * p) * its,
        "algorithmic_tflops": mu_algorithmic_flops(m, d, p, k) * its / 1e12,
        "config": {"workload": desc, "m": m, "d": d, "p": p, "n_components": k, "solver": "mu",
                   "parallelism": "X/U row-sharded, Y/Z column-sharded x%d, V replicated, "
                                  "1 RCCL all-reduce of (d+k)*k f32 per iteration" % world},
        "roofline": {
            "bound": "mfma",
            "kernel": "cmfk::gemm_kernel<%d, %d, 0, 4>  (%s data pass)" % (0 if dom == "gemm_nn" else 1, 256 if k >= 256 else k,
                                                                     "NN: X V / Y Z" if dom == "gemm_nn" else "TN: X^T U / Y^T V"),
            "achieved": achieved,
            "peak": FP32_MFMA_PEAK_TFLOPS,
            "unit": "TFLOP/s",
            "frac": achieved / FP32_MFMA_PEAK_TFLOPS,
            "traffic": traffic,
            "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/traffic_%s.json)" % args.workload,
            "algorithmic_flops_per_launch": dfl / max(dn, 1),
            "avg_launch_ms": dms / max(dn, 1),
            "launches": dn,
            "per_class_ms_per_step": {c: v[0] / args.steps for c, v in classes.items()},
        },
        "rel_residual": {"x": (ex2 / x2) ** 0.5 if x2 > 0 else None, "y": (ey2 / y2) ** 0.5 if y2 > 0 else None,
                         "note": "rank-0 shard, after warmup+steps iterations"},
    }
    if world == 1 and not args.no_cpu_baseline:
        cits, shp, n_it, el, threads = cpu_baseline(k)
        work_ratio = (float(shp[1]) * (shp[0] + shp[2])) / (float(d) * (m + p))
        out["cpu_baseline"] = {
            "value": cits * work_ratio,
            "unit": "it/s",
            "cores": threads,
            "kind": "port",
            "sample": "oracle/cmf_oracle.mu_update_step (NumPy float64, reference operation order incl. "
                      "(U V^T) V), m=d=p=%d k=%d, %d iterations in %.1f s = %.3f it/s, scaled to the bench "
                      "shape by d(m+p) ratio %.3g" % (shp[0], k, n_it, el, cits, work_ratio),
        }
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
