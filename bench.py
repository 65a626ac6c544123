#!/usr/bin/env python3
"""bench.py -- factor-update iterations/s of the CMF solvers on MI355X.

Metric (BASELINE.json): factor-update iterations/s (and cells/s) at
n_components=256, 65536^2 dense.  One "step" = one full ``update_step`` of the
reference (MU: V,U,Z, pycmf/cmf_solvers.py:248-263; Newton: U,Z,V, :510-522)
over synthetic X (m x d), Y (d x p) already resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c4|c2|c3|c5|tiny] [--option name=value ...]

Default workload = C4 (the configuration the metric is quoted on; it fits one
GPU: 34.4 GB).  c2 / c3 / c5 are the other BASELINE configs (parity-test
shapes; same JSON contract, for the record in DESIGN.md).

N > 1: either under torch.distributed.run (one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the
environment) or plain ``python bench.py --gpus N``, which starts the N rank processes itself before any GPU call
in the parent and forwards rank 0's JSON line.  The problem is
FIXED (strong scaling): rank g owns rows [g*m/N, (g+1)*m/N) of X/U and the same
fraction of Y's columns / Z's rows; V is replicated and reassembled by one RCCL
all-reduce per iteration (pycmf_amd/sharded.py).  The per-row Newton workload (c3) shards the rows of all three
factors instead (two contexts per rank, factor rows exchanged; ShardedNewtonRows).

Prints ONE JSON line on rank 0 with ``roofline`` (dominant kernel class, HIP
events on the launch stream over the timed region) and ``cpu_baseline`` (CPU
oracle, bounded sample, rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "c4": dict(m=65536, d=65536, p=65536, k=256, solver="mu",
               desc="BASELINE configs[3]: CMF(n_components=256, solver='mu'), dense 65536x65536 X, "
                    "65536x65536 Y, non-negative synthetic"),
    "c4q": dict(m=16384, d=16384, p=16384, k=256, solver="mu",
                desc="C4 at a quarter of its edge (the N = 8 dress rehearsal of the test suite): CMF(n_components=256, solver='mu'), dense "
                     "16384x16384 X and Y, non-negative synthetic"),
    "c2": dict(m=16384, d=8192, p=4096, k=128, solver="mu",
               desc="BASELINE configs[1]: CMF(n_components=128, solver='mu', linear link), dense "
                    "16384x8192 X, 8192x4096 Y, non-negative synthetic"),
    "c3": dict(m=32768, d=16384, p=8192, k=256, solver="newton", x_link="linear", y_link="logit", ratio=0.5, y_kind=1, l2=0.0,
               desc="BASELINE configs[2] at the reference's defaults (l2_reg = 0, pycmf/cmf.py:622): CMF(n_components=256, solver='newton', "
                    "y_link='logit', sg_sample_ratio=0.5, device sampler) on dense 32768x16384 X, 16384x8192 Y"),
    "c3r": dict(m=32768, d=16384, p=8192, k=256, solver="newton", x_link="linear", y_link="logit", ratio=0.5, y_kind=1, l2=0.1,
                desc="BASELINE configs[2] with l2_reg = 0.1 (NOT the reference's default: rounds 1-5 quoted this as 'c3'; with it "
                     "_safe_invert's clamp never acts): CMF(n_components=256, solver='newton', y_link='logit', sg_sample_ratio=0.5, device "
                     "sampler) on dense 32768x16384 X, 16384x8192 Y"),
    "c3x": dict(m=32768, d=16384, p=8192, k=256, solver="newton", x_link="logit", y_link="logit", ratio=0.5, x_kind=1, y_kind=1, l2=0.0,
                desc="BASELINE configs[2] with the sigmoid link on BOTH sides, the reference's default l2_reg = 0: CMF(n_components=256, solver='newton', x_link='logit', "
                     "y_link='logit', sg_sample_ratio=0.5, device sampler) on dense 32768x16384 X, 16384x8192 Y (sigmoid(N(0,1)) targets)"),
    "c5": dict(m=1000000, d=100000, p=64, k=256, solver="newton", x_link="linear", y_link="linear", ratio=1.0,
               nnz_per_row=100,
               desc="BASELINE configs[4]: CSR X 1e6 x 1e5 at 0.1% nnz (100 per row, values 1.0, native CSR), "
                    "dense Y 1e5 x 64, n_components=256, newton solver, linear links"),
    "c5z": dict(m=1000000, d=100000, p=64, k=256, solver="newton", x_link="linear", y_link="linear", ratio=1.0,
                nnz_per_row=100, zipf=1.1,
                desc="BASELINE configs[4] with bag-of-words column statistics: CSR X 1e6 x 1e5, 100 non-zeros per row (values 1.0, native "
                     "CSR) whose columns follow Zipf(1.1) popularity over a randomly ordered vocabulary (samples/toxic_comments.ipynb:"
                     "446-465: word counts are Zipfian; uniformly scattered columns, workload c5, are the worst case for the SpMM's "
                     "gathers), dense Y 1e5 x 64, n_components=256, newton solver, linear links"),
    "c5zs": dict(m=1000000, d=100000, p=64, k=256, solver="newton", x_link="linear", y_link="linear", ratio=1.0,
                 nnz_per_row=100, zipf=1.1, zipf_sorted=True,
                 desc="c5z with the vocabulary ordered by popularity (hot columns first): what a relabelling of V's rows at upload would "
                      "give the SpMM; CSR X 1e6 x 1e5, 100 non-zeros per row, Zipf(1.1) columns, dense Y 1e5 x 64, n_components=256, "
                      "newton solver, linear links"),
    "c5l": dict(m=1000000, d=100000, p=64, k=256, solver="newton", x_link="linear", y_link="logit", ratio=1.0,
                nnz_per_row=100, l1=2.0, l2=5.0, nn_mask=3, y_kind=2, y_param=0.1, may_diverge=True,
                desc="BASELINE configs[4] with the reference's own Newton settings (samples/toxic_comments.ipynb:853-856: "
                     "x_link='linear', y_link='logit', l1_reg=2, l2_reg=5, U and V non-negative): CSR X 1e6 x 1e5 at 0.1% nnz "
                     "(values 1.0, native CSR), dense Y 1e5 x 64 in {0,1} (10% ones), n_components=256.  On this unstructured synthetic Y "
                     "the reference's undamped iteration itself diverges by iteration 3 (profiles/r05_c5l_probe.txt): the line is a "
                     "throughput figure, rel_residual says where the iterates are; c5l_l2x10 is the same on bounded iterates"),
    "c5l_l2x10": dict(m=1000000, d=100000, p=64, k=256, solver="newton", x_link="linear", y_link="logit", ratio=1.0,
                      nnz_per_row=100, l1=2.0, l2=50.0, nn_mask=3, y_kind=2, y_param=0.1,
                      desc="c5l with l2_reg = 50 INSTEAD OF the notebook's 5 (5 scaled by d / 1e4, the number of rows each Z gradient sums "
                           "over -- a deviation from the reference's settings, made because every iterate then stays finite): CSR X 1e6 x "
                           "1e5 at 0.1% nnz (values 1.0, native CSR), dense Y 1e5 x 64 in {0,1} (10% ones), n_components=256, newton, "
                           "x linear / y logit, l1_reg=2, U and V non-negative"),
    "tiny5l": dict(m=20000, d=3000, p=64, k=64, solver="newton", x_link="linear", y_link="logit", ratio=1.0,
                   nnz_per_row=30, l1=2.0, l2=5.0, nn_mask=3, y_kind=2, y_param=0.1,
                   desc="debug shape (native CSR X, y logit newton; the notebook's l1 = 2, l2 = 5, on which the iteration stays bounded at this size)"),
    "tiny": dict(m=2048, d=1024, p=512, k=64, solver="mu", desc="debug shape (mu)"),
    "tiny3": dict(m=1536, d=1024, p=512, k=64, solver="newton", x_link="linear", y_link="logit", ratio=0.5, y_kind=1,
                  desc="debug shape (per-row newton, y logit, sg_sample_ratio 0.5, device sampler)"),
    "tiny5": dict(m=20000, d=3000, p=64, k=64, solver="newton", x_link="linear", y_link="linear", ratio=1.0,
                  nnz_per_row=30, desc="debug shape (native CSR X, linear newton)"),
}
WORKLOADS["c3z"] = WORKLOADS["c3"]   # rounds 4-5 named the l2 = 0 configuration c3z: kept as an alias
FP32_MFMA_PEAK_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
BF16_MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 (v_mfma_f32_32x32x16_bf16)
HBM_PEAK_GBPS = 8000.0         # MI355X_MICROARCH.md: HBM3E spec


def algorithmic_flops(w):
    """SURVEY.md 8(d).  MU: 4kd(m+p) + 4k^2(m+d+p).  Newton: residual/gradient contractions
    + per-row Hessians k(k+1)(m s_d + p s_d + d(s_m+s_p)) on the sweeps that need them -- every H_i is
    symmetric and only its upper half is credited (8(d): "symmetric half may be credited as half")."""
    m, d, p, k = w["m"], w["d"], w["p"], w["k"]
    if "nnz_per_row" in w:  # sparse X: the X contractions cost 2 k nnz each
        nnz = float(m) * w["nnz_per_row"]
        base = 4.0 * k * nnz + 4.0 * k * d * p + 4.0 * k * k * (m + d + p)
    else:
        base = 4.0 * k * d * (m + p) + 4.0 * k * k * (m + d + p)
    if w["solver"] == "mu":
        return base
    r = w.get("ratio", 1.0)
    xl, yl = w["x_link"], w["y_link"]
    s_d, s_m, s_p = int(d * r), int(m * r), int(p * r)
    f = base
    if xl == "logit" or r < 1:  # U rows
        f += k * (k + 1.0) * m * s_d + 2.0 * k * m * s_d
    if yl == "logit" or r < 1:  # Z rows
        f += k * (k + 1.0) * p * s_d + 2.0 * k * p * s_d
    if xl == "logit" or yl == "logit" or r < 1:  # V rows
        hx = s_m if (xl == "logit" or r < 1) else 0
        hy = s_p if (yl == "logit" or r < 1) else 0
        f += k * (k + 1.0) * d * (hx + hy) + 2.0 * k * d * (s_m + s_p)
    return f


def cpu_baseline(w, budget_s=20.0):
    """CPU oracle (reference operation order incl. the wasteful (U V^T) V, float64) per BASELINE.md section 3.  MU workloads time
    BASELINE config C2 IN FULL (16384 x 8192 / 8192 x 4096, k = 128: 3 warm-up + 5 timed update_step, all BLAS threads) and scale
    to the bench shape by the reference-order flop ratio 8 k d (m + p) (C4 itself needs 140 GB of float64 temporaries and minutes
    of data generation: 'scaled from C2', never a full C4 iteration here).  Newton workloads, whose per-row Python loop with one
    eigh(k x k) per row makes full shapes impractical, time ONE update_step at a reduced shape with at least 1024 rows per
    data-side factor ("reduced shape (<= 2048 rows per factor) and scaled") at the workload's own k, links,
    ratio and regularisation, and scale by algorithmic work (scale factor >= 1e-3); the per-row sweeps run on ONE BLAS thread --
    LAPACK's 256 x 256 eigh is an order of magnitude slower on a many-threaded OpenBLAS -- and say so (cores = 1)."""
    import numpy as np
    from oracle import cmf_oracle as O
    k = w["k"]
    per_row = False
    if w["solver"] == "mu":
        ms, ds, ps, ks = 16384, 8192, 4096, 128
        if w["m"] * w["d"] < ms * ds:     # debug shapes: the shape itself
            ms, ds, ps, ks = w["m"], w["d"], w["p"], k
    else:
        per_row = w["x_link"] == "logit" or w["y_link"] == "logit" or w.get("ratio", 1.0) < 1.0
        if "nnz_per_row" in w:   # C5 / C5L: keep p, d >= 1024 rows of V, many rows of U (shared Hessian on the X side: no eigh per row)
            ms, ds, ps, ks = min(w["m"], 16384), min(w["d"], 1024 if per_row else 2048), w["p"], k
        else:                    # C3: 1024 rows of U and V, 512 of Z (2560 eigh(k x k) per iteration)
            ms, ds, ps, ks = min(w["m"], 1024), min(w["d"], 1024), min(w["p"], 512), k
    rng = np.random.RandomState(42)
    if "nnz_per_row" in w and w["solver"] != "mu":
        X = (rng.rand(ms, ds) < float(w["nnz_per_row"]) / w["d"]).astype(np.float64)   # binary bag of words at the workload's density
    else:
        X = np.abs(rng.randn(ms, ds))
    Y = np.abs(rng.randn(ds, ps))
    if w.get("x_kind") == 1:
        X = 1.0 / (1.0 + np.exp(-rng.randn(ms, ds)))
    if w.get("y_kind") == 2:
        Y = (rng.rand(ds, ps) < w.get("y_param", 0.1)).astype(np.float64)
    elif w.get("y_link") == "logit":
        Y = 1.0 / (1.0 + np.exp(-rng.randn(ds, ps)))
    sc = np.sqrt(max(X.mean(), 1e-12) / ks)
    U, V, Z = (sc * np.abs(rng.randn(n, ks)) for n in (ms, ds, ps))
    if w["solver"] == "mu":
        def step():
            O.mu_update_step(X, Y, U, V, Z)
        warm, timed = 3, 5
    else:
        np.random.seed(0)   # (oracle.safe_invert pins ITS eigh calls to one BLAS thread; the GEMMs of the step keep every thread -- ADVICE r5)

        def step():
            nnm = w.get("nn_mask", 0)
            O.newton_update_step(X, Y, U, V, Z, 0.5, w.get("l1", 0.0), w.get("l2", 0.1), w["x_link"], w["y_link"],
                                 bool(nnm & 1), bool(nnm & 2), bool(nnm & 4), ratio=w.get("ratio", 1.0), pert=0.2)
        warm, timed = (0, 1) if per_row else (1, None)
    for _ in range(warm):
        step()
    t0 = time.perf_counter()
    iters = 0
    while True:
        step()
        iters += 1
        el = time.perf_counter() - t0
        if (timed is not None and iters >= timed) or el > budget_s or (timed is None and iters >= 3 and el > budget_s / 2):
            break
    blas = "unknown"
    threads = os.cpu_count() or 1
    try:
        from threadpoolctl import threadpool_info
        info = [i for i in threadpool_info() if i.get("user_api") == "blas"]
        if info:
            threads = max(i.get("num_threads", 1) for i in info)
            blas = ", ".join(sorted({"%s %s" % (i.get("internal_api", "?"), i.get("version", "?")) for i in info}))
    except Exception:
        pass
    if w["solver"] == "mu":   # reference-order flops: 8 k d (m + p) (+ Grams / applies, the same order in both shapes)
        ref = lambda m_, d_, p_, k_: 8.0 * k_ * d_ * (m_ + p_) + 4.0 * k_ * k_ * (m_ + d_ + p_)
        ratio = ref(ms, ds, ps, ks) / ref(w["m"], w["d"], w["p"], k)
    else:
        sample = dict(w)
        if "nnz_per_row" in w:   # the sample keeps the density: nnz per row scales with d
            sample["nnz_per_row"] = w["nnz_per_row"] * float(ds) / w["d"]
        sample.update(m=ms, d=ds, p=ps)
        ratio = algorithmic_flops(sample) / algorithmic_flops(w)
    return dict(its=iters / el, shape=(ms, ds, ps, ks), iters=iters, seconds=el, threads=threads, blas=blas,
                cpu_count=os.cpu_count() or 1, ratio=ratio, per_row=per_row)


def launch_ranks(n, argv):
    """``python bench.py --gpus N`` without a launcher: start N fresh rank processes (one per GPU) BEFORE anything in
    this process touches the GPU, wait for them, forward rank 0's JSON line.  Never exec: the parent only supervises."""
    import shutil
    import socket
    import subprocess
    from pycmf_amd.comm import fresh_job_env
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    base_env, job_dir = fresh_job_env()     # private, empty rendezvous directory + random key: nothing stale, nobody else's
    procs = []
    for r in range(n):
        env = dict(base_env, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    failed = None
    while failed is None and any(q.poll() is None for q in procs):
        for r, q in enumerate(procs):
            if q.poll() not in (None, 0):
                failed = r
        time.sleep(0.2)
    if failed is None:
        failed = next((r for r, q in enumerate(procs) if q.returncode != 0), None)
    if failed is not None:  # a dead rank leaves the others waiting in a collective: end exactly the processes started here
        for q in procs:
            if q.poll() is None:
                q.kill()
        for q in procs:
            q.wait()
        shutil.rmtree(job_dir, ignore_errors=True)
        raise SystemExit("bench.py: rank %d exited with code %s" % (failed, procs[failed].returncode))
    out = procs[0].stdout.read().decode()
    shutil.rmtree(job_dir, ignore_errors=True)
    line = next((ln for ln in reversed(out.splitlines()) if ln.startswith("{")), None)
    if line is None:
        raise SystemExit("bench.py: rank 0 printed no JSON line")
    print(line)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c4", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--overlap-chunks", type=int, default=1,
                    help="N > 1 GPUs, MU on dense data: reduce the (d + k) k buffer in this many row blocks on a side stream while the "
                         "next block's partial is computed (default 1: one serial all-reduce)")
    ap.add_argument("--mu-collective", choices=("auto", "rsag", "allreduce"), default=None,
                    help="N > 1 GPUs, MU: 'allreduce' = ONE all-reduce of the (d + k) k buffer and the replicated epilogue (north_star's "
                         "protocol); 'rsag' = reduce-scatter of the partial, V epilogue on the rank's row block, all-gather of V, the two "
                         "k^2 Grams in the same two RCCL groups; 'auto' (default) = both timed on the live ranks before the warm-up, the "
                         "faster kept (rsag only when it wins by > 2 %%), decision and timings in collective.protocol_trial")
    ap.add_argument("--max-warmup", type=int, default=40,
                    help="per-row Newton workloads: upper limit of the extra warm-up iterations run until the clamp / refinement counts "
                         "of two consecutive iterations agree (steady state)")
    ap.add_argument("--tol", type=float, default=0.0,
                    help="> 0: the reference's convergence check (pycmf/cmf_solvers.py:175-187: the error metric every 10th iteration, "
                         "default tol of pycmf.CMF 1e-4) runs INSIDE the timed region -- the device error pass, its 16-byte read-back and, "
                         "N > 1, the all-reduce of the two squared residuals; the stop itself is never taken (exactly --steps iterations)")
    ap.add_argument("--dump-rows", default=None, metavar="PREFIX",
                    help="after the timed iterations every rank writes the rows it owns out of 16 fixed global rows of U, V, Z to "
                         "PREFIX.rank<r>.npz (tools/compare_rows.py compares two such sets: the N = 8 dress rehearsal against N = 1)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="cmf_set_option knob for A/B runs (e.g. row_symmetric=0); recorded in config")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args.gpus, sys.argv[1:])
    # (the host driver of this pool only supports dmabuf IPC: without this RCCL cannot open peer memory; it must be in the
    # environment before the first HIP call of the process, i.e. before libcmfhip is loaded)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    import numpy as np
    from pycmf_amd import _lib
    from pycmf_amd.sharded import (block_bounds, make_sharded_mu, make_sharded_newton, make_sharded_newton_rows,
                                   nnz_balanced_bounds, shard_bounds)

    # No PyTorch anywhere: the collectives are RCCL calls inside libcmfhip on the context's stream (pycmf_amd/comm.py).
    # Test hooks (not used by the driver): CMF_BENCH_SAME_DEVICE=1 puts every rank on GPU 0 and CMF_COMM_BACKEND=host swaps
    # RCCL (which refuses two ranks on one device) for the host-staged test double, so the N>1 code path runs on a 1-GPU box;
    # CMF_BENCH_FORCE_DIST=1 takes the sharded drivers at N = 1 too (RCCL with a single rank).
    if os.environ.get("CMF_BENCH_SAME_DEVICE") == "1":
        local_rank = 0
    use_dist = world > 1 or os.environ.get("CMF_BENCH_FORCE_DIST") == "1"

    w = WORKLOADS[args.workload]
    m, d, p, k = w["m"], w["d"], w["p"], w["k"]
    bf16x6 = "gemm_arith=1" in args.option
    newton = w["solver"] == "newton"
    sharded_ok = (not newton) or (w["x_link"] == "linear" and w["y_link"] == "linear" and w["ratio"] == 1.0)
    rows_mode = newton and not sharded_ok and use_dist
    bounds_fn = block_bounds if rows_mode else shard_bounds   # the row-sharded Newton all-gathers equal blocks
    r0, r1 = bounds_fn(m, world, rank)
    c0, c1 = bounds_fn(p, world, rank)
    if "nnz_per_row" in w and not rows_mode and world > 1:
        # CSR X in nnz-balanced row blocks (the synthetic rows all hold nnz_per_row entries, so this coincides
        # with the balanced row count; real inputs take the same call in pycmf_amd/multi_gpu.py)
        off = nnz_balanced_bounds(np.arange(0, m + 1, dtype=np.int64) * w["nnz_per_row"], world)
        r0, r1 = int(off[rank]), int(off[rank + 1])

    # every context of a rank launches on ONE stream (the first context's own); the collectives are enqueued on it
    ctx = _lib.Context(local_rank)
    for kv in args.option:
        name, _, val = kv.partition("=")
        ctx.set_option(name, int(val))
    ctx.set_problem(r1 - r0, d, c1 - c0, k)
    if "nnz_per_row" in w:
        # CSR row block generated on the host (values 1.0: binary bag-of-words like the reference's
        # notebook) and kept native on the device
        import scipy.sparse as sp
        npr = w["nnz_per_row"]
        X_cols_block = None
        if rows_mode:
            # the row-sharded Newton (y logit: BASELINE configs[4] with the reference's own settings is an 8-GPU configuration) holds
            # X by rows AND by columns: every rank draws the WHOLE synthetic matrix from the same stream and cuts its two blocks
            rng = np.random.default_rng(42)
            Xall = sp.csr_matrix((np.ones(m * npr), rng.integers(0, d, size=m * npr, dtype=np.int32),
                                  np.arange(0, m * npr + 1, npr, dtype=np.int64)), shape=(m, d))
            X = Xall[r0:r1]
            q0_, q1_ = block_bounds(d, world, rank)
            X_cols_block = Xall[:, q0_:q1_].tocsr()
            del Xall
        else:
            rng = np.random.default_rng(42 + rank)
            rows = r1 - r0
            if "zipf" in w:
                # columns by popularity rank ~ Zipf(s) (inverse CDF), duplicates inside a row re-drawn uniformly (binary bag of words:
                # a word counts once per document), the vocabulary in random order unless zipf_sorted
                wgt = 1.0 / np.arange(1, d + 1, dtype=np.float64) ** w["zipf"]
                cdf = np.cumsum(wgt)
                cdf /= cdf[-1]
                cols = np.empty(rows * npr, dtype=np.int32)
                step = 1 << 24
                for a in range(0, rows * npr, step):
                    cols[a:a + step] = np.searchsorted(cdf, rng.random(min(step, rows * npr - a)), side="right").astype(np.int32)
                np.minimum(cols, d - 1, out=cols)
                cols = cols.reshape(rows, npr)
                cols.sort(axis=1)
                dup = np.zeros(cols.shape, dtype=bool)
                dup[:, 1:] = cols[:, 1:] == cols[:, :-1]
                cols[dup] = rng.integers(0, d, size=int(dup.sum()), dtype=np.int32)
                if not w.get("zipf_sorted"):
                    cols = np.random.default_rng(7).permutation(d).astype(np.int32)[cols]
                X = sp.csr_matrix((np.ones(rows * npr), cols.reshape(-1), np.arange(0, rows * npr + 1, npr, dtype=np.int64)), shape=(rows, d))
                del cols, dup
            else:
                X = sp.csr_matrix((np.ones(rows * npr), rng.integers(0, d, size=rows * npr, dtype=np.int32),
                                   np.arange(0, rows * npr + 1, npr, dtype=np.int64)), shape=(rows, d))
        ctx.set_option("sparse_mode", 2)
        ctx.set_data(0, X)
        del X
        scale = (npr / d / k) ** 0.5
    else:
        ctx.fill_data_synthetic(0, 42, r0, 0, w.get("x_kind", 0))   # X rows [r0,r1): values depend only on global coordinates
        scale = (0.7979 / k) ** 0.5             # 'random' init rule sqrt(mean / k), pycmf/cmf.py:111
    # Y columns [c0,c1): |N(0,1)|, or the targets of a logit side: sigmoid(N(0,1)) as in
    # benchmarks/benchmark_cmf.py:78, {0,1} labels as in samples/toxic_comments.ipynb
    ctx.fill_data_synthetic(1, 43, 0, c0, w.get("y_kind", 0), w.get("y_param", 0.0))
    l1_reg, l2_reg, nn_mask = w.get("l1", 0.0), w.get("l2", 0.1), w.get("nn_mask", 0)
    ctx.fill_factor_synthetic(_lib.CMF_U, 101, r0, scale)
    ctx.fill_factor_synthetic(_lib.CMF_V, 102, 0, scale)
    ctx.fill_factor_synthetic(_lib.CMF_Z, 103, c0, scale)
    ctxs = [ctx]
    drv = None
    coll = None
    if use_dist:
        from pycmf_amd.comm import init_collectives
        coll = init_collectives(ctx, rank, world, timed=True)

    if not newton:
        drv = make_sharded_mu(ctx, coll, chunks=args.overlap_chunks if "nnz_per_row" not in w and not bf16x6 else 1,
                              mode=args.mu_collective if args.overlap_chunks <= 1 else "allreduce")

        def do_step(it):
            drv.step(0.0, 0.0, 7)
    elif sharded_ok:
        drv = make_sharded_newton(ctx, coll, alpha=0.5, nn_mask=nn_mask, pert=0.2,
                                  single_collective="newton_single_collective=1" in args.option)

        def do_step(it):
            drv.step(l1_reg, l2_reg, 7)
    elif use_dist:
        # per-row sweeps (logit link and / or sampling): a second context holds the rank's COLUMNS of X and rows
        # of Y with U and Z whole, and sweeps the rank's rows of V; factor rows are all-gathered in between
        q0, q1 = block_bounds(d, world, rank)
        ctx_v = _lib.Context(local_rank, ctx.stream_handle())
        for kv in args.option:
            name, _, val = kv.partition("=")
            ctx_v.set_option(name, int(val))
        ctx_v.set_problem(m, q1 - q0, p, k)
        if "nnz_per_row" in w:
            ctx_v.set_option("sparse_mode", 2)
            ctx_v.set_data(0, X_cols_block)
            del X_cols_block
        else:
            ctx_v.fill_data_synthetic(0, 42, 0, q0, w.get("x_kind", 0))
        ctx_v.fill_data_synthetic(1, 43, q0, 0, w.get("y_kind", 0), w.get("y_param", 0.0))
        ctx_v.fill_factor_synthetic(_lib.CMF_U, 101, 0, scale)
        ctx_v.fill_factor_synthetic(_lib.CMF_V, 102, q0, scale)
        ctx_v.fill_factor_synthetic(_lib.CMF_Z, 103, 0, scale)
        ctxs.append(ctx_v)
        drv = make_sharded_newton_rows(ctx, ctx_v, (r0, r1, q0, q1, c0, c1), (m, d, p), coll, 0.5,
                                       w["x_link"], w["y_link"], nn_mask=nn_mask, pert=0.2, ratio=w["ratio"])

        def do_step(it):
            drv.step(l1_reg, l2_reg, 7, 1000 + it)
    else:
        def do_step(it):
            ctx.newton_step_device_sampled(0.5, l1_reg, l2_reg, w["x_link"], w["y_link"], nn_mask, 7, 0.2,
                                           w["ratio"], 1000 + it)

    def sync_all():
        for c_ in ctxs:
            c_.sync()              # hipStreamSynchronize of the launch stream
        if coll:
            coll.barrier()         # every rank's stream has drained

    checks = []

    def maybe_check(it):
        # iteration numbers count from the first timed step, as a fit's would from its first iteration
        if args.tol > 0 and (it + 1) % 10 == 0:
            sq = np.array(ctx.residual_sq(w.get("x_link", "linear"), w.get("y_link", "linear")))
            if coll:
                sq = coll.all_reduce_host(sq)
            checks.append(float(0.5 * sq[0] ** 0.5 + 0.5 * sq[1] ** 0.5))

    for it in range(args.warmup):
        do_step(it)
    # Per-row Newton: the work of an iteration depends on the iterate -- how many rows _safe_invert's clamp acts on, how many the
    # float64 refinement redoes (C3 at l2 = 0: none for six iterations, every row of U from the seventh on; round 4 and round 5 each
    # quoted a line from BEFORE such a change).  Warm up until two consecutive iterations record the same clamp / refinement
    # counts (at most --max-warmup more), and say how long that took; the line is checked again after the timed region.
    warm_extra = 0
    warm_ms = []
    if newton and not sharded_ok and not use_dist:
        def clamp_counts():
            ctx.sync()
            st = ctx.newton_clamp_stats(full=True)
            return (st[0], st[2])
        prev = clamp_counts()
        last, same = None, 0
        while warm_extra < args.max_warmup:
            t_it = time.perf_counter()
            do_step(args.warmup + warm_extra)
            warm_extra += 1
            cur = clamp_counts()
            warm_ms.append((time.perf_counter() - t_it) * 1e3)
            delta = (cur[0] - prev[0], cur[1] - prev[1])
            prev = cur
            same = same + 1 if (last is not None and delta == last) else 0
            last = delta
            # (C3 idles at "8192 rows of Z clamped" for iterations 3-6 before every row of U joins at the seventh: two equal
            # iterations in a row are no steady state -- at least 12 iterations in all, then three alike)
            # (C3X: iterations 6-12 take 495 ms each, the 13th 625 ms once -- the eigen-solve of one sweep in a single iteration --
            # then 493 ms for good: the counts do not show it.  At least 16 iterations in all, and the last three alike in TIME too.)
            t3 = warm_ms[-3:]
            steady_time = len(t3) == 3 and max(t3) <= 1.03 * min(t3)
            if same >= 2 and steady_time and args.warmup + warm_extra >= min(16, args.max_warmup):
                break
        args.warmup += warm_extra
    for c_ in ctxs:
        c_.kernel_timing(2)        # HIP events around the data-pass launches only (the class the roofline prices): an event pair per
        c_.kernel_timing_reset()   # launch serialises the stream for a few microseconds, 7 % of a C2 iteration when every launch has one
    if coll:
        ctx.sync()
        coll.reset()
    sync_all()
    t0 = time.perf_counter()
    ctx.marker()
    fused_check = args.tol > 0 and not newton and not use_dist   # cmf_mu_step_error: the step and the error of its result, no pass over X / Y
    for it in range(args.steps):
        if fused_check and (it + 1) % 10 == 0:
            ex2_, ey2_ = ctx.mu_step_error(0.0, 0.0, 7)
            checks.append(float(0.5 * ex2_ ** 0.5 + 0.5 * ey2_ ** 0.5))
        else:
            do_step(args.warmup + it)
            maybe_check(it)
        ctx.marker()               # one event per iteration on the launch stream: the auditable time series
    sync_all()
    elapsed = time.perf_counter() - t0
    if coll:
        elapsed = float(coll.all_reduce_host([elapsed], "max")[0])   # the slowest rank's clock
    marks = ctx.marker_times()
    series_ms = [b_ - a_ for a_, b_ in zip(marks, marks[1:])]
    launch_points = coll.launch_points_seen() if coll and hasattr(coll, "launch_points_seen") else None
    coll_stats = coll.stats() if coll else None
    coll_exposed = coll.exposed_ms() if coll and hasattr(coll, "exposed_ms") else None
    coll_kinds = coll.stats_by_kind() if coll and hasattr(coll, "stats_by_kind") else {}
    if args.dump_rows:
        pick = np.random.RandomState(7)
        dump = {}
        for name, which, n, lo, hi in (("U", _lib.CMF_U, m, r0, r1), ("V", _lib.CMF_V, d, 0, d if rank == 0 else 0), ("Z", _lib.CMF_Z, p, c0, c1)):
            rows = np.sort(pick.choice(n, size=min(16, n), replace=False))
            mine = rows[(rows >= lo) & (rows < hi)]
            F = ctx.get_factor(which)
            dump[name + "_rows"] = mine
            dump[name] = F[mine - lo]
            dump[name + "_absmax"] = np.array([np.abs(F).max() if F.size else 0.0])
        np.savez(args.dump_rows + ".rank%d.npz" % rank, **dump)
    replicas = None
    if coll and not rows_mode:
        # V is replicated: after the timed iterations every rank must hold the same V, bit for bit (a collective that summed
        # the wrong chunks, or ranks that never met, shows up here and not only as a strange residual)
        vsum = float(np.abs(ctx.get_factor(_lib.CMF_V)).sum(dtype=np.float64))
        hi = float(coll.all_reduce_host([vsum], "max")[0])
        lo = -float(coll.all_reduce_host([-vsum], "max")[0])
        replicas = {"sum_abs_V_max_over_ranks": hi, "sum_abs_V_min_over_ranks": lo, "identical": hi == lo}

    names = ("gemm_nn", "gemm_tn", "gemm_pair", "gemm_small", "gemm_nt", "spmm", "rowhess", "eigen", "elementwise")
    classes = {c: tuple(sum(v) for v in zip(*(c_.kernel_time(c) for c_ in ctxs))) for c in names}
    rh_samples = tuple(sum(v) for v in zip(*(c_.rowhess_samples() for c_ in ctxs)))
    # the other kernel classes: a few more iterations OUTSIDE the timed region with events around every launch
    extra = min(args.steps, 3)
    for c_ in ctxs:
        c_.kernel_timing(1)
        c_.kernel_timing_reset()
    t_extra = time.perf_counter()
    for it in range(extra):
        do_step(args.warmup + args.steps + it)
    sync_all()
    t_extra = (time.perf_counter() - t_extra) / max(extra, 1) * 1e3
    other = {c: tuple(sum(v) for v in zip(*(c_.kernel_time(c) for c_ in ctxs))) for c in names}
    for c_ in ctxs:
        c_.kernel_timing(False)
    ex2, ey2 = ctx.residual_sq(w.get("x_link", "linear"), w.get("y_link", "linear"))
    x2, y2 = ctx.data_sq()
    kp = ctx.geometry()[3]

    if coll:
        coll.barrier()
        coll.close()
    if rank != 0:
        return

    ms_per_step = elapsed / args.steps * 1e3
    its = args.steps / elapsed
    # dominant kernel class = most device time on rank 0
    dom = max(("gemm_nn", "gemm_tn", "gemm_pair", "spmm", "rowhess"), key=lambda c: classes[c][0])
    dms, dn, dfl = classes[dom]
    if dom == "spmm":
        # HBM-bound gather kernel (column-blocked, output-stationary SpMM).  Algorithmic (compulsory) bytes of one A*F
        # product: the regrouped entry list once (16 B per non-zero) + the gathered operand once + the output once;
        # the k_pad*4-byte factor-row gather per non-zero is served by the XCD L2s (rows of a 2 MB column block are
        # shared by the workgroups of an XCD) and reported separately.
        nnz = float(r1 - r0) * w["nnz_per_row"]
        comp = nnz * 16.0 + ((r1 - r0) + d) * kp * 4.0
        achieved = comp / (dms / max(dn, 1) * 1e-3) / 1e9
        spmm_traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if world == 1 and os.path.exists(tpath):
            try:
                spmm_traffic = json.load(open(tpath)).get("spmm")
            except Exception:
                spmm_traffic = None
        roof = {"bound": "hbm", "kernel": "cmfk::spmm_blocked_kernel<%d>  (A F and A^T F of the native CSR input)" % (kp // 64),
                "achieved": achieved, "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": spmm_traffic,
                "traffic_unit": "bytes per launch past L2 (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/traffic_c5.json): "
                                "Infinity-Cache hits of the factor-row gathers included",
                "gathered_GBps": nnz * (kp * 4.0 + 16.0) / (dms / max(dn, 1) * 1e-3) / 1e9,
                "gather_peak_GBps": 17800.0,
                "gather_frac": nnz * (kp * 4.0 + 16.0) / (dms / max(dn, 1) * 1e-3) / 1e9 / 17800.0,
                "note": "achieved = compulsory HBM bytes (entry list + gathered operand + output) / launch time; the bound that "
                        "matters is the per-nonzero factor-row gather (gathered_GBps) against the measured L2 row-gather rate "
                        "of the chip, 16.8-18.8 TB/s (MI355X_MICROARCH.md, Indexed rows: gather into LDS)",
                "avg_launch_ms": dms / max(dn, 1), "launches": dn}
    else:
        achieved = dfl / (dms * 1e-3) / 1e12 if dms > 0 else 0.0
        reference_order = achieved
        if dom == "rowhess" and rh_samples[0] > 0:
            # linear sampled sides share partial outer-product sums between rows (option row_classes): the kernel is priced
            # on the sample rows it actually gathers, the rate in the reference's own count is reported beside it
            achieved *= rh_samples[1] / rh_samples[0]
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if world == 1 and os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(dom)
            except Exception:
                traffic = None
        if dom == "rowhess":
            kname = ("cmfk::row_hess_kernel<%d>  (fused per-row gradient + Hessian over the sampled rows; flops "
                     "credited for the symmetric half of each H_i, k(k+1) per sample row gathered)" % kp)
        elif dom == "gemm_pair":
            kname = ("cmfk::gemm_pair_kernel  (k_pad = 128: X^T U with Y Z, X V with Y^T V -- the two data passes of an MU "
                     "half-iteration as one launch, their K-steps cut into equal quotas per CU)")
        else:
            kname = "cmfk::gemm_kernel<%d, %d, 0, 4>  (%s data pass)" % (
                0 if dom == "gemm_nn" else 1, 256 if k >= 256 else kp,
                "NN: X V / Y Z / W KR" if dom == "gemm_nn" else "TN: X^T U / Y^T V / W^T KR")
        peak = FP32_MFMA_PEAK_TFLOPS
        if bf16x6 and dom in ("gemm_nn", "gemm_tn", "gemm_pair"):
            # optional arithmetic: 6 bf16 MFMA products per fp32-equivalent product -> the bound is the bf16 matrix peak / 6
            peak = BF16_MFMA_PEAK_TFLOPS / 6.0
            kname = ("cmfk::bf16x6_gemm_kernel  (data pass on v_mfma_f32_32x32x16_bf16, three bf16 planes per fp32 operand, "
                     "six cross products, f32 accumulation; peak = bf16 dense peak / 6)")
            traffic = None
        roof = {"bound": "mfma",
                "kernel": kname,
                "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                "frac": achieved / peak, "traffic": traffic,
                "traffic_unit": "HBM bytes per launch (rocprofv3 PMC, profiles/traffic_%s.json)" % args.workload,
                "algorithmic_flops_per_launch": dfl / max(dn, 1) * (achieved / reference_order if reference_order > 0 else 1.0),
                "avg_launch_ms": dms / max(dn, 1), "launches": dn}
        if dom == "rowhess":
            roof["sample_rows_reference"] = rh_samples[0] / args.steps
            roof["sample_rows_gathered"] = rh_samples[1] / args.steps
            roof["reference_order_tflops"] = reference_order
            roof["note"] = ("sample_rows_reference = outer products the reference runs per iteration (one per row and sampled "
                            "index); sample_rows_gathered = what the kernel runs: with a linear link the rows of a group of 4 "
                            "share the sums of the samples they have in common (shared partial sums, DESIGN.md section 4); "
                            "achieved / frac price the kernel on the gathered rows, reference_order_tflops on the reference's count")
    if dom == "rowhess" and bf16x6 and kp == 256:
        roof["kernel"] = ("cmfk::row_hess6_kernel  (fused per-row gradient + Hessian; Hessian on v_mfma_f32_32x32x16_bf16 from three "
                          "bf16 planes of sqrt(w) o, six products per block; flops credited for the symmetric half, k(k+1) per sample)")
        roof["peak"] = BF16_MFMA_PEAK_TFLOPS / 6.0
        roof["frac"] = roof["achieved"] / roof["peak"]
    elif dom == "rowhess" and kp == 256 and "row_symmetric=0" not in args.option:
        # the MFMAs cover 28 whole 32x32 blocks above the diagonal and three 16x16 sub-blocks of each of the 8 diagonal ones
        # (34 block-equivalents; 36 with row_symmetric=1 / 3, which compute the diagonal blocks in full)
        whole = 36 if any(o in args.option for o in ("row_symmetric=1", "row_symmetric=3")) else 34
        roof["mfma_executed_tflops"] = achieved * (whole * 2048.0 + 4 * kp) / (kp * (kp + 1.0) + 4 * kp)
    roof["traffic_provenance"] = (None if roof.get("traffic") is None else
                                  "stored: profiles/traffic_%s.json, from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                  "command (tools/refresh_r05.sh, gfx950 corrections of MI355X_MICROARCH.md); PMC collection cannot run "
                                  "inside a timed bench run, so this figure is NOT measured by the run that printed this line" % args.workload)
    roof["per_class_ms_per_step"] = {c: v[0] / extra for c, v in other.items() if v[1]}
    roof["per_class_launches_per_step"] = {c: v[1] / extra for c, v in other.items() if v[1]}
    roof["instrumented_ms_per_step"] = t_extra
    roof["per_class_note"] = ("%d extra iterations after the timed region with HIP events around every launch (nested scopes record once, "
                              "under the outermost class): the classes add up to instrumented_ms_per_step, the wall clock of THOSE iterations, "
                              "which exceeds ms_per_step where the events keep host round trips from overlapping with device work; inside the "
                              "timed region only the data-pass classes carry events (what achieved / avg_launch_ms are computed from)" % extra)
    out = {
        "metric": "factor-update iterations/s (%s solver: one full update_step per iteration)" % w["solver"],
        "value": its,
        "unit": "it/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "bf16x6 (fp32 operands split exactly into three bf16 planes, f32 accumulation)" if bf16x6 else "f32",
        "data": "synthetic",
        "cells_per_s": (float(m) * d + float(d) * p) * its,
        "algorithmic_tflops": algorithmic_flops(w) * its / 1e12,
        "config": {"workload": w["desc"], "m": m, "d": d, "p": p, "n_components": k, "solver": w["solver"],
                   "parallelism": "single GPU: X, Y and all three factors resident on one device, no collective" if not use_dist else ("rows of U, V, Z sharded x%d (every sweep row-parallel; X and Y held by rows and by "
                                   "columns), factor rows reassembled by 3 in-place RCCL all-gathers, (m+p+d)*k f32 in all "
                                   "per iteration" % world) if rows_mode else
                                  ("X/U row-sharded, Y/Z column-sharded x%d, V replicated; the one sum over the ranks per iteration, (d+k)*k f32, "
                                   "as %s" % (world, "reduce-scatter + row-blocked V epilogue + all-gather, the two k^2 Gram all-reduces inside the same two RCCL groups"
                                              if getattr(drv, "mode", None) == "rsag" else "1 RCCL all-reduce"))},
        "roofline": roof,
        "rel_residual": {"x": (ex2 / x2) ** 0.5 if x2 > 0 else None, "y": (ey2 / y2) ** 0.5 if y2 > 0 else None,
                         "note": "rank-0 shard, after warmup+steps iterations"},
    }
    out["series_ms"] = {"per_iteration": [round(v, 4) for v in series_ms],
                        "note": "HIP events on the launch stream of rank 0, one per iteration inside the timed region"}
    if use_dist:
        calls, nbytes, cms = coll_stats if coll_stats else (0, 0, 0.0)
        exposed = (cms if args.overlap_chunks <= 1 else coll_exposed)
        out["collective"] = {"backend": coll.backend, "ranks": world,
                             "ranks_seen": getattr(coll, "ranks_seen", None), "rank_seen": getattr(coll, "rank_seen", None),
                             "protocol": getattr(drv, "mode", None) or ("all_gather x3" if rows_mode else "all_reduce"),
                             "protocol_chosen": getattr(drv, "mode", None), "protocol_trial": getattr(drv, "protocol_trial", None),
                             "calls_per_iteration": calls / args.steps, "payload_bytes_per_iteration": nbytes / args.steps,
                             "launch_points_per_iteration": (launch_points / args.steps) if launch_points is not None else None,
                             "ms_per_iteration": cms / args.steps,
                             "per_kind": {kname: {"calls_per_iteration": kv[0] / args.steps, "payload_bytes_per_iteration": kv[1] / args.steps,
                                                  "us_per_call": (kv[2] / kv[0] * 1e3) if kv[0] and kv[2] > 0 else None}
                                          for kname, kv in coll_kinds.items() if kv[0]},
                             "overlap_chunks": args.overlap_chunks,
                             "exposed_ms_per_iteration": exposed / args.steps,
                             "hidden_ms_per_iteration": (0.0 if args.overlap_chunks <= 1 else max(0.0, cms - coll_exposed)) / args.steps,
                             "compute_ms_per_iteration": ms_per_step - exposed / args.steps,
                             "replicas": replicas,
                             "note": "rank 0; ms = events on the launch stream around every collective (waiting for the "
                                     "slowest rank included); ranks_seen = ncclCommCount of the communicator; replicas = every "
                                     "rank's copy of V compared after the run"}
    if args.tol > 0:
        out["convergence_check"] = {"tol": args.tol, "every": 10, "checks_in_timed_region": len(checks), "errors": checks,
                                    "form": ("cmf_mu_step_error: ||X||^2 - 2 <U, X V> + <U^T U, V^T V> from the step's own products (NT error pass only "
                                             "where the expansion would cancel)" if fused_check else "cmf_residual_sq: NT error pass over X and Y"),
                                    "note": "the reference's stopping test (pycmf/cmf_solvers.py:175-187) evaluated inside the timed region; "
                                            "the stop is not taken, so that exactly `steps` iterations are timed"}
    for key in ("x_link", "y_link", "ratio", "l1", "l2", "nn_mask"):
        if key in w:
            out["config"][key] = w[key]
    if newton:
        out["config"]["l1"], out["config"]["l2"] = l1_reg, l2_reg
        if args.workload == "c3r" or "l2" not in w:
            out["config"]["l2_note"] = ("l2_reg = 0.1 is this bench's choice, not BASELINE.json's (which names no regularisation: the "
                                        "reference's default is 0, pycmf/cmf.py:622)" + ("; --workload c3 is the configuration at the default" if args.workload == "c3r" else ""))
        out["config"]["workload"] += "; l1_reg = %g, l2_reg = %g" % (l1_reg, l2_reg)
    if args.option:
        out["config"]["options"] = list(args.option)
    if args.workload == "c4" and not bf16x6:
        out["note"] = ("fp32 MFMA arithmetic (the default).  Opt-in --option gemm_arith=1 (fp32 operands split exactly into three "
                       "bf16 planes, six products on the bf16 matrix pipe, fp32-equivalent to 4e-7): 23.3-24.0 it/s, "
                       "profiles/r01_c4_n1_bench_bf16x6.json, DESIGN.md section 4")
    if newton:
        st = ctx.newton_clamp_stats(full=True)
        out["conditioning"] = {"clamped_rows_float32": st[0], "max_ratio_left_in_float32": st[1], "rows_refined_in_float64": st[2],
                               "max_condition_estimate_plain_solves": st[3],
                               "clamp_routes": dict(zip(("tridiagonal_eigen_solve_rows", "rank_one_rows"), ctx.newton_clamp_routes())),
                               "note": "per-row sweeps since the context was created (cmf_newton_clamp_stats): rows whose float32 Hessian "
                                       "went through the spectral clamp, largest ||H||_F / pert left in float32, rows redone in float64, "
                                       "largest max H_ii / min L_ii^2 over all plain Cholesky solves"}
    if world == 1 and not args.no_cpu_baseline:
        cb = cpu_baseline(w)
        full = w["solver"] == "mu"
        per_row = bool(cb.get("per_row"))
        out["cpu_baseline"] = {
            "value": cb["its"] * cb["ratio"],
            "unit": "it/s",
            "cores": cb["threads"],
            "host_cpu_count": cb["cpu_count"],
            "blas": cb["blas"],
            # "port" = the oracle timed on the configuration itself (C2 in full) or scaled by a flop ratio of order 1e-2 (C4 from
            # C2); "extrapolated" = ONE reduced-shape step scaled by 1e-3 .. 1e-2 of the algorithmic work (per-row Newton)
            "kind": "extrapolated" if (newton and per_row) else "port",
            "cores_note": ("eigh: 1 BLAS thread (oracle.safe_invert pins LAPACK's k x k eigh, an order of magnitude slower on a many-threaded "
                           "OpenBLAS); GEMMs of the step: %d" % cb["threads"] if per_row else
                           "all host cores" if cb["threads"] >= cb["cpu_count"] else
                           "the BLAS under NumPy (%s) is built for at most %d threads: that is every thread this build can use on the "
                           "%d-CPU host" % (cb["blas"], cb["threads"], cb["cpu_count"])),
            "sample": ("oracle/cmf_oracle %s update_step (NumPy float64, reference operation order) at m,d,p,k=%s%s: %d timed "
                       "iterations in %.1f s = %.3f it/s on %d BLAS threads%s; scaled to the bench shape by the %s ratio %.3g"
                       % (w["solver"], cb["shape"], " (BASELINE config C2 in full)" if full and cb["shape"][0] == 16384 else "",
                          cb["iters"], cb["seconds"], cb["its"], cb["threads"], " (eigh on 1)" if per_row else "",
                          "reference-order flop" if full else "algorithmic-work", cb["ratio"])),
        }
    if newton:
        out["steady_state"] = {"extra_warmup_iterations": warm_extra, "instrumented_over_timed": t_extra / ms_per_step if ms_per_step > 0 else None,
                               "warmup_series_ms": [round(x, 1) for x in warm_ms] if warm_extra else [],
                               "note": "warm-up continued until three consecutive iterations recorded the same clamp / refinement counts and took "
                                       "the same time to 3 % (at least 16 iterations in all; warmup_series_ms: their wall clock, host-synchronised); the "
                                       "iterations after the timed region (instrumented_ms_per_step) must not cost more than 1.25 x the timed "
                                       "ones, else the line is refused (exit code 3)"}
    sys.stdout.flush()
    print(json.dumps(out), flush=True)
    if newton and ms_per_step > 0 and t_extra > 1.25 * ms_per_step and extra > 0 and not w.get("may_diverge"):
        sys.stderr.write("bench.py: NOT a steady-state line: the %d iterations after the timed region took %.1f ms each against %.1f ms "
                         "inside it (a regime change of the solver behind the timed window); raise --warmup / --max-warmup\n"
                         % (extra, t_extra, ms_per_step))
        sys.exit(3)


if __name__ == "__main__":
    main()
