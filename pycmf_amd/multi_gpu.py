"""``CMF(n_gpus=N)``: the sharded solvers behind the scikit-learn front end.

The parent process never touches a GPU: it validates, initialises the factors, writes the job (X, Y, factors, solver
keywords) to a scratch directory and starts N worker processes -- one rank per GPU, ``torch.distributed`` with backend
'nccl' (RCCL over xGMI), rendezvous on 127.0.0.1.  Rank g memory-maps the job, takes its row block of X / U and column
block of Y / Z (CSR X is cut into nnz-balanced row blocks for the MU solver) and runs
``fit_mu_sharded`` / ``fit_newton_sharded`` (pycmf_amd/sharded.py: the reference's outer loop, pycmf/cmf_solvers.py:132-195,
with one all-reduce per iteration for V).  For large inputs with a non-custom init, rank 0 first computes the initial
factors with the device-side initialisers on its GPU (whole X, Y resident once, released before the fit).  The parent
reassembles U, V, Z in the caller's arrays.

Test hooks (a 1-GPU box): PYCMF_AMD_SAME_DEVICE=1 puts every rank on GPU 0 and PYCMF_AMD_DIST_BACKEND=gloo replaces RCCL,
which refuses two ranks on one device.
"""
import json
import os
import shutil
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np
import scipy.sparse as sp

from .sharded import block_bounds, nnz_balanced_bounds, shard_bounds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _save(path, A):
    if sp.issparse(A):
        sp.save_npz(path + ".npz", A.tocsr(), compressed=False)
    else:
        np.save(path + ".npy", np.ascontiguousarray(A))


def _load(path):
    if os.path.exists(path + ".npz"):
        return sp.load_npz(path + ".npz")
    return np.load(path + ".npy", mmap_mode="r")


def partition(X, Y, solver, world):
    """(row offsets of X / U, row offsets of V, column offsets of Y = row offsets of Z), world + 1 entries each."""
    m, d = X.shape
    p = Y.shape[1]
    if solver == "newton":   # the row-sharded Newton all-gathers equal blocks
        cut = lambda n: np.array([block_bounds(n, world, r)[0] for r in range(world)] + [n], dtype=np.int64)
        return cut(m), cut(d), cut(p)
    rows = nnz_balanced_bounds(X.indptr, world) if sp.issparse(X) and X.format == "csr" else \
        np.array([shard_bounds(m, world, r)[0] for r in range(world)] + [m], dtype=np.int64)
    cols = np.array([shard_bounds(p, world, r)[0] for r in range(world)] + [p], dtype=np.int64)
    return rows, None, cols


class MultiGpuResult:
    """What the front end needs from a solver object after the fit."""

    def __init__(self, ex, ey):
        self._err = ex + ey

    def reconstruction_error(self):
        return self._err

    def release(self):
        pass


def fit_multi_gpu(X, Y, U, V, Z, solver, n_gpus, params, timeout=None):
    """Run the sharded fit on ``n_gpus`` worker processes; U, V, Z are updated in place.
    Returns (n_iter, MultiGpuResult)."""
    if sp.issparse(X):
        X = X.tocsr()
    if sp.issparse(Y):
        Y = Y.toarray() if solver == "newton" else Y.tocsc()   # column blocks of Y
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    job = tempfile.mkdtemp(prefix="pycmf_amd_job_", dir=base)
    try:
        _save(os.path.join(job, "X"), X)
        _save(os.path.join(job, "Y"), Y)
        if not params.get("init"):   # else rank 0 computes the start on its GPU and writes this file
            np.savez(os.path.join(job, "factors.npz"), U=U, V=V, Z=Z)
        rows, vrows, cols = partition(X, Y, solver, n_gpus)
        meta = dict(solver=solver, params=params, rows=[int(v) for v in rows], cols=[int(v) for v in cols],
                    vrows=None if vrows is None else [int(v) for v in vrows])
        with open(os.path.join(job, "job.json"), "w") as f:
            json.dump(meta, f)
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        procs = []
        for r in range(n_gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), MASTER_ADDR="127.0.0.1",
                       MASTER_PORT=str(port))
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
            log = open(os.path.join(job, "rank%d.log" % r), "wb")
            procs.append((subprocess.Popen([sys.executable, "-m", "pycmf_amd._worker", job], env=env, stdout=log,
                                           stderr=subprocess.STDOUT), log))
        failed = None
        t0 = time.time()
        while failed is None and any(q.poll() is None for q, _ in procs):
            for r, (q, _) in enumerate(procs):
                if q.poll() not in (None, 0):
                    failed = r
            if timeout and time.time() - t0 > timeout:
                failed = -1
            time.sleep(0.05)
        if failed is None:
            failed = next((r for r, (q, _) in enumerate(procs) if q.returncode != 0), None)
        for q, log in procs:
            if q.poll() is None:
                q.kill()      # exactly the processes started here
            q.wait()
            log.close()
        if failed is not None:
            r = max(failed, 0)
            tail = open(os.path.join(job, "rank%d.log" % r), "rb").read().decode("utf-8", "replace")[-3000:]
            raise RuntimeError("multi-GPU fit: rank %d failed (exit code %s)\n%s" % (r, procs[r][0].returncode, tail))
        n_iter, ex2, ey2 = None, 0.0, 0.0
        for r in range(n_gpus):
            o = np.load(os.path.join(job, "out%d.npz" % r))
            U[rows[r]:rows[r + 1]] = o["U"]
            Z[cols[r]:cols[r + 1]] = o["Z"]
            if r == 0:
                V[...] = o["V"]
                n_iter, ex2, ey2 = int(o["n_iter"]), float(o["ex2"]), float(o["ey2"])
        return n_iter, MultiGpuResult(np.sqrt(ex2), np.sqrt(ey2))
    finally:
        shutil.rmtree(job, ignore_errors=True)
