"""``CMF(n_gpus=N)``: the sharded solvers behind the scikit-learn front end.

The parent process never touches a GPU: it validates, initialises the factors, writes the job (X, Y, factors, solver
keywords) to a scratch directory and starts N worker processes -- one rank per GPU; the collectives are RCCL calls inside
libcmfhip (pycmf_amd/comm.py), no PyTorch anywhere.  Rank g memory-maps the job, takes its block and runs

* ``fit_mu_sharded`` (MU) or ``fit_newton_linear_sharded`` (Newton, linear links, no sampling) on north_star's partition: row
  block of X / U (CSR X: nnz-balanced), column block of Y / rows of Z, V replicated, ONE large all-reduce per iteration;
* ``fit_newton_sharded`` (any other Newton configuration): rows of all three factors in equal blocks, X and Y held by rows and
  by columns, three in-place all-gathers of factor rows per iteration

(pycmf_amd/sharded.py: the reference's outer loop, pycmf/cmf_solvers.py:132-195).  For large inputs with a non-custom init,
rank 0 first computes the initial factors with the device-side initialisers on its GPU (whole X, Y resident once, released
before the fit).  The parent reassembles U, V, Z in the caller's arrays.

Test hooks (a 1-GPU box): PYCMF_AMD_SAME_DEVICE=1 puts every rank on GPU 0 and CMF_COMM_BACKEND=host replaces RCCL -- which
refuses two ranks on one device -- with the host-staged test double of pycmf_amd/comm.py.
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
    """Dense inputs are written ONCE, as float32 -- the arithmetic type of the device (the upload shim rounds float64 the same
    way) -- in row slabs straight from the caller's array (any strides): half the bytes of a float64 copy and no contiguous
    temporary.  (r04 wrote np.save of the float64 arrays: a second 68.7 GB host copy at C4.)"""
    if sp.issparse(A):
        sp.save_npz(path + ".npz", A.tocsr(), compressed=False)
        return
    A = np.asarray(A)
    out = np.lib.format.open_memmap(path + ".npy", mode="w+", dtype=np.float32, shape=A.shape)
    step = max(1, (64 << 20) // max(1, A.shape[1] * 4))
    for r in range(0, A.shape[0], step):
        out[r:r + step] = A[r:r + step]
    out.flush()
    del out


def _load(path):
    if os.path.exists(path + ".npz"):
        return sp.load_npz(path + ".npz")
    return np.load(path + ".npy", mmap_mode="r")


def partition(X, Y, solver, world, params=None):
    """(row offsets of X / U, row offsets of V, column offsets of Y = row offsets of Z), world + 1 entries each."""
    from ._worker import linear_newton
    m, d = X.shape
    p = Y.shape[1]
    if solver == "newton" and not (params is not None and linear_newton(params)):
        # the row-sharded Newton all-gathers equal blocks
        cut = lambda n: np.array([block_bounds(n, world, r)[0] for r in range(world)] + [n], dtype=np.int64)
        return cut(m), cut(d), cut(p)
    # north_star's partition (MU, and Newton with linear links and no sampling): CSR X in nnz-balanced row blocks
    rows = nnz_balanced_bounds(X.indptr, world) if sp.issparse(X) and X.format == "csr" else \
        np.array([shard_bounds(m, world, r)[0] for r in range(world)] + [m], dtype=np.int64)
    cols = np.array([shard_bounds(p, world, r)[0] for r in range(world)] + [p], dtype=np.int64)
    return rows, None, cols


def _plain(v):
    """JSON-serialisable copy of a parameter value (NumPy scalars from user code included)."""
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.floating,)):
        return float(v)
    if isinstance(v, (np.bool_,)):
        return bool(v)
    return v


def _job_base(nbytes):
    """/dev/shm when it is writable AND has room for the job (it is RAM), else the ordinary temp directory."""
    shm = "/dev/shm"
    if os.path.isdir(shm) and os.access(shm, os.W_OK):
        try:
            if shutil.disk_usage(shm).free > 1.25 * nbytes + (64 << 20):
                return shm
        except OSError:
            pass
    return None


def _nbytes(A):
    if sp.issparse(A):
        return A.data.nbytes + A.indices.nbytes + A.indptr.nbytes
    return np.asarray(A).nbytes


def _job_nbytes(A):
    """Bytes `_save` writes for one matrix: dense inputs travel as float32 whatever the caller's dtype, sparse ones as they are."""
    if sp.issparse(A):
        return _nbytes(A)
    return int(A.shape[0]) * int(A.shape[1]) * 4


def _gpu_runtime_present():
    """Has ANYTHING in this process opened the GPU -- not only pycmf_amd?  An open descriptor on /dev/kfd or a DRM render node
    means some HIP / HSA runtime is initialised (torch, cupy, hip-python, another ctypes library, rocprofv3's preloaded tool
    library with --pmc): a forked child would inherit state it must not.  (A runtime library that is merely mapped has opened
    neither: HIP initialises on the first call.)"""
    try:
        for fd in os.listdir("/proc/self/fd"):
            try:
                target = os.readlink("/proc/self/fd/" + fd)
            except OSError:
                continue
            if target == "/dev/kfd" or target.startswith("/dev/dri/renderD"):
                return True
    except OSError:
        return True      # cannot tell: assume the worst
    return False


def can_fork_ranks():
    """May the ranks be forked off this process (they then read the caller's X, Y, U, V, Z in place -- no copy of the job at all)?
    OPT-IN: PYCMF_AMD_FORK_RANKS=1 (the default is the fresh-child path with float32 job files).  Even then only while
    this process (i) has never initialised or even loaded a GPU runtime -- pycmf_amd's own flag, torch's, AND what the process
    table shows: no descriptor on /dev/kfd or a render node (a runtime brought up by any other library or by a profiler's
    preloaded tool is seen there) -- and (ii) runs a single Python thread.  (Native pools Python cannot see are the
    opting-in caller's statement: OpenBLAS registers fork handlers for its own; a parent with a live OpenMP team or RCCL proxy
    threads must not opt in.)"""
    if os.environ.get("PYCMF_AMD_FORK_RANKS", "0") != "1" or not hasattr(os, "fork"):
        return False
    from . import _lib
    if _lib.gpu_touched():
        return False
    torch = sys.modules.get("torch")
    try:
        if torch is not None and torch.cuda.is_initialized():
            return False
    except Exception:
        return False
    if _gpu_runtime_present():
        return False
    import threading
    if threading.active_count() > 1:
        return False
    return True


def _forked_rank(job, meta, X, Y, factors, env, rank):
    """Body of a forked rank: the parent's arrays are this process's arrays (copy-on-write pages, never written)."""
    try:
        os.environ.update(env)
        log = os.open(os.path.join(job, "rank%d.log" % rank), os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o600)
        os.dup2(log, 1)
        os.dup2(log, 2)
        from . import _worker
        _worker.run_rank(job, meta, X, Y, factors)
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)
    except BaseException:
        import traceback
        traceback.print_exc()
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(1)


def _linear(params):
    from ._worker import linear_newton
    return linear_newton(params)


class _ForkedProc:
    """subprocess.Popen's poll / kill / wait / returncode over a multiprocessing.Process."""

    def __init__(self, p):
        self.p = p

    def poll(self):
        return self.p.exitcode

    @property
    def returncode(self):
        return self.p.exitcode

    def kill(self):
        self.p.kill()

    def wait(self):
        self.p.join()


last_fit_info = {}               # how the most recent fit handed its data to the ranks: forked (no copy) or through float32 files
last_collective_calls = [None]   # rank 0's collective count of the most recent fit (tests: one large all-reduce per iteration)


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
    Returns (n_iter, MultiGpuResult).  ``timeout`` (seconds; default: the environment's PYCMF_AMD_FIT_TIMEOUT, else none) bounds
    the whole fit; a rank that waits longer than CMF_COMM_TIMEOUT (default 600 s) for the job's RCCL id fails on its own."""
    if timeout is None and os.environ.get("PYCMF_AMD_FIT_TIMEOUT"):
        timeout = float(os.environ["PYCMF_AMD_FIT_TIMEOUT"])
    if sp.issparse(X):
        X = X.tocsr()
    if sp.issparse(Y):
        Y = Y.tocsc()   # column blocks of Y
    params = _plain(params)
    sparse_y = sp.issparse(Y)
    if solver == "newton" and sparse_y and not _linear(params):
        Y = Y.toarray()
    fork = can_fork_ranks()
    job = tempfile.mkdtemp(prefix="pycmf_amd_job_", dir=None if fork else _job_base(_job_nbytes(X) + _job_nbytes(Y) + U.nbytes + V.nbytes + Z.nbytes))
    last_fit_info.clear()
    last_fit_info.update(forked=fork, job_bytes=0)
    try:
        if not fork:
            _save(os.path.join(job, "X"), X)
            _save(os.path.join(job, "Y"), Y)
            if not params.get("init"):   # else rank 0 computes the start on its GPU and writes this file
                np.savez(os.path.join(job, "factors.npz"), U=U, V=V, Z=Z)
            last_fit_info["job_bytes"] = sum(os.path.getsize(os.path.join(job, f)) for f in os.listdir(job))
        rows, vrows, cols = partition(X, Y, solver, n_gpus, params)
        meta = dict(solver=solver, params=params, rows=[int(v) for v in rows], cols=[int(v) for v in cols],
                    vrows=None if vrows is None else [int(v) for v in vrows])
        with open(os.path.join(job, "job.json"), "w") as f:
            json.dump(meta, f)
        s = socket.socket()          # MASTER_PORT only names the job (pycmf_amd/comm.py); nothing listens on it
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        procs = []
        for r in range(n_gpus):
            env = dict(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env["HSA_ENABLE_IPC_MODE_LEGACY"] = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            env["CMF_COMM_DIR"], env["CMF_COMM_KEY"] = job, "job"
            if fork:
                # the rank IS this process at this point in time: X, Y, U, V, Z are reachable without a copy; it has never touched a
                # GPU, so the child initialises its own runtime
                import multiprocessing as mp
                q = mp.get_context("fork").Process(target=_forked_rank, args=(job, meta, X, Y, (U, V, Z), env, r), daemon=False)
                q.start()
                procs.append((_ForkedProc(q), None))
                continue
            env = dict(os.environ, **env)
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
            if log is not None:
                log.close()
        if failed is not None:
            r = max(failed, 0)
            tail = open(os.path.join(job, "rank%d.log" % r), "rb").read().decode("utf-8", "replace")[-3000:]
            raise RuntimeError("multi-GPU fit: rank %d failed (exit code %s)\n%s" % (r, procs[r][0].returncode, tail))
        n_iter, ex2, ey2 = None, 0.0, 0.0
        last_collective_calls[0] = None
        for r in range(n_gpus):
            o = np.load(os.path.join(job, "out%d.npz" % r))
            U[rows[r]:rows[r + 1]] = o["U"]
            Z[cols[r]:cols[r + 1]] = o["Z"]
            if r == 0:
                V[...] = o["V"]
                n_iter, ex2, ey2 = int(o["n_iter"]), float(o["ex2"]), float(o["ey2"])
                last_collective_calls[0] = int(o["collective_calls"]) if "collective_calls" in o else None
        return n_iter, MultiGpuResult(np.sqrt(ex2), np.sqrt(ey2))
    finally:
        shutil.rmtree(job, ignore_errors=True)
