"""One rank of ``CMF(n_gpus=N)`` (started by pycmf_amd/multi_gpu.py, never imported by user code)."""
import json
import os
import sys

import numpy as np
import scipy.sparse as sp


def main(job):
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    local = 0 if os.environ.get("PYCMF_AMD_SAME_DEVICE") == "1" else int(os.environ.get("LOCAL_RANK", rank))
    backend = os.environ.get("PYCMF_AMD_DIST_BACKEND", "nccl")
    import torch
    import torch.distributed as dist
    from pycmf_amd.multi_gpu import _load
    from pycmf_amd.sharded import fit_mu_sharded, fit_newton_sharded
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    meta = json.load(open(os.path.join(job, "job.json")))
    P, rows, cols = meta["params"], meta["rows"], meta["cols"]
    X, Y = _load(os.path.join(job, "X")), _load(os.path.join(job, "Y"))
    if P.get("init"):
        # initial factors on rank 0's GPU: the device-side initialisers need the whole matrices resident once
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
            np.savez(os.path.join(job, "factors.npz"), U=U0, V=V0, Z=Z0)
        dist.barrier()
    F = np.load(os.path.join(job, "factors.npz"))
    r0, r1, c0, c1 = rows[rank], rows[rank + 1], cols[rank], cols[rank + 1]
    dense = lambda A: np.ascontiguousarray(A) if not sp.issparse(A) else A
    Xr = dense(X[r0:r1])
    Yc = dense(Y[:, c0:c1]) if not sp.issparse(Y) else Y[:, c0:c1].tocsr()
    U, V, Z = np.array(F["U"][r0:r1]), np.array(F["V"]), np.array(F["Z"][c0:c1])
    stats = {}
    if meta["solver"] == "mu":
        U, V, Z, n_iter = fit_mu_sharded(Xr, Yc, U, V, Z, l1_reg=P["l1_reg"], l2_reg=P["l2_reg"], max_iter=P["max_iter"],
                                         tol=P["tol"], device=local, verbose=P["verbose"] if rank == 0 else 0, stats=stats)
    else:
        q0, q1 = meta["vrows"][rank], meta["vrows"][rank + 1]
        Xc = dense(X[:, q0:q1]) if not sp.issparse(X) else X[:, q0:q1].tocsr()
        Yr = dense(Y[q0:q1])
        U, V, Z, n_iter = fit_newton_sharded(
            Xr, Xc, Yc, Yr, U, V, Z, alpha=P["alpha"], l1_reg=P["l1_reg"], l2_reg=P["l2_reg"], x_link=P["x_link"],
            y_link=P["y_link"], U_non_negative=P["U_non_negative"], V_non_negative=P["V_non_negative"],
            Z_non_negative=P["Z_non_negative"], hessian_pertubation=P["hessian_pertubation"],
            sg_sample_ratio=P["sg_sample_ratio"], random_state=P["random_state"], max_iter=P["max_iter"], tol=P["tol"],
            device=local, verbose=P["verbose"] if rank == 0 else 0, stats=stats)
    np.savez(os.path.join(job, "out%d.npz" % rank), U=U, V=V, Z=Z, n_iter=n_iter, ex2=stats["ex2"], ey2=stats["ey2"])
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
