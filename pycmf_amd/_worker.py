"""One rank of ``CMF(n_gpus=N)`` (started by pycmf_amd/multi_gpu.py, never imported by user code).  No PyTorch: the collectives
are RCCL calls inside libcmfhip (pycmf_amd/comm.py); the unique id travels through the job directory."""
import json
import os
import sys
import time

import numpy as np
import scipy.sparse as sp


def linear_newton(P):
    """north_star's partition with ONE large all-reduce serves this Newton configuration."""
    return P["x_link"] == "linear" and P["y_link"] == "linear" and P["sg_sample_ratio"] >= 1.0


def main(job):
    """A rank started as a fresh process: the job's data are the float32 files the parent wrote (memory-mapped: a rank touches the
    pages of its own blocks only)."""
    from pycmf_amd.multi_gpu import _load
    meta = json.load(open(os.path.join(job, "job.json")))
    run_rank(job, meta, _load(os.path.join(job, "X")), _load(os.path.join(job, "Y")))


def run_rank(job, meta, X, Y, factors=None):
    """One rank's fit.  X, Y: the WHOLE inputs as this process sees them -- memory-mapped job files (main) or, in a rank forked
    from the caller's process before it ever touched a GPU, the caller's own arrays (no copy at all: multi_gpu.fit_multi_gpu);
    ``factors``: the initial (U, V, Z) in the same way, else read from the job directory."""
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if os.environ.get("PYCMF_AMD_SAME_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", rank))
    os.environ.setdefault("CMF_COMM_DIR", job)
    os.environ.setdefault("CMF_COMM_KEY", "job")
    from pycmf_amd.sharded import fit_mu_sharded, fit_newton_linear_sharded, fit_newton_sharded
    P, rows, cols = meta["params"], meta["rows"], meta["cols"]
    fpath = os.path.join(job, "factors.npz")
    if P.get("init"):
        # initial factors on rank 0's GPU: the device-side initialisers need the whole matrices resident once; the other
        # ranks wait for the file (no collective is open yet: nothing can time out underneath them)
        if rank == 0:
            import warnings
            from pycmf_amd import _lib
            from pycmf_amd.estimator import initial_factors
            from pycmf_amd.factor_init import DeviceOperand
            I = P["init"]
            Xw = X if sp.issparse(X) else np.asarray(X)
            Yw = Y if sp.issparse(Y) else np.asarray(Y)
            ctx = _lib.Context(local)
            ctx.set_problem(Xw.shape[0], Xw.shape[1], Yw.shape[1], I["n_components"])
            ctx.set_data(0, Xw); ctx.set_data(1, Yw)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                U0, V0, Z0 = initial_factors(Xw, Yw, None, None, None, op_x=DeviceOperand(ctx, 0, Xw.shape),
                                             op_y=DeviceOperand(ctx, 1, Yw.shape), **I)
            ctx.close()
            np.savez(fpath + ".tmp.npz", U=U0, V=V0, Z=Z0)
            os.replace(fpath + ".tmp.npz", fpath)
        else:
            while not os.path.exists(fpath):
                time.sleep(0.05)
    F = dict(zip("UVZ", factors)) if factors is not None and not P.get("init") else np.load(fpath)
    r0, r1, c0, c1 = rows[rank], rows[rank + 1], cols[rank], cols[rank + 1]
    # dense blocks go to the upload shim as VIEWS (any strides, float32 or float64: Context.set_data): no host copy of a block
    dense = lambda A: A
    Xr = dense(X[r0:r1])
    Yc = dense(Y[:, c0:c1]) if not sp.issparse(Y) else Y[:, c0:c1].tocsr()
    U, V, Z = np.array(F["U"][r0:r1]), np.array(F["V"]), np.array(F["Z"][c0:c1])
    stats = {}
    common = dict(max_iter=P["max_iter"], tol=P["tol"], device=local, verbose=P["verbose"] if rank == 0 else 0, stats=stats,
                  rank=rank, world=world, update_mask=int(P.get("update_mask", 7)))
    if meta["solver"] == "mu":
        # a seeded fit is run-to-run reproducible: no wall-clock choice between protocols that differ in summation order
        U, V, Z, n_iter = fit_mu_sharded(Xr, Yc, U, V, Z, l1_reg=P["l1_reg"], l2_reg=P["l2_reg"],
                                         collective="allreduce" if P.get("random_state") is not None else None, **common)
    elif linear_newton(P):
        U, V, Z, n_iter = fit_newton_linear_sharded(
            Xr, Yc, U, V, Z, alpha=P["alpha"], l1_reg=P["l1_reg"], l2_reg=P["l2_reg"], U_non_negative=P["U_non_negative"],
            V_non_negative=P["V_non_negative"], Z_non_negative=P["Z_non_negative"],
            hessian_pertubation=P["hessian_pertubation"], **common)
    else:
        q0, q1 = meta["vrows"][rank], meta["vrows"][rank + 1]
        Xc = dense(X[:, q0:q1]) if not sp.issparse(X) else X[:, q0:q1].tocsr()
        Yr = dense(Y[q0:q1])
        U, V, Z, n_iter = fit_newton_sharded(
            Xr, Xc, Yc, Yr, U, V, Z, alpha=P["alpha"], l1_reg=P["l1_reg"], l2_reg=P["l2_reg"], x_link=P["x_link"],
            y_link=P["y_link"], U_non_negative=P["U_non_negative"], V_non_negative=P["V_non_negative"],
            Z_non_negative=P["Z_non_negative"], hessian_pertubation=P["hessian_pertubation"],
            sg_sample_ratio=P["sg_sample_ratio"], random_state=P["random_state"], **common)
    np.savez(os.path.join(job, "out%d.npz" % rank), U=U, V=V, Z=Z, n_iter=n_iter, ex2=stats["ex2"], ey2=stats["ey2"],
             collective_calls=stats.get("collective_calls", -1))


if __name__ == "__main__":
    main(sys.argv[1])
