"""Where the wall time of ``CMF(...).fit(X, Y)`` goes at a BASELINE shape (default C2: 16384 x 8192 / 8192 x 4096, k = 128,
mu, default 'nndsvdar' init, 100 iterations): host validation, upload, initialisers, solver loop, download.
usage: python tools/fit_breakdown.py [m d p k iters]"""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sklearn.utils import check_array
from pycmf_amd import CMF
from pycmf_amd.factor_init import initialize_mf, DeviceOperand
from pycmf_amd.solver_shell import HipMUSolver

a = [int(v) for v in sys.argv[1:]]
m, d, p, k, iters = (a + [16384, 8192, 4096, 128, 100][len(a):])[:5]
rng = np.random.RandomState(42)
t = time.time(); X, Y = np.abs(rng.randn(m, d)), np.abs(rng.randn(d, p)); print("generate data            %.3f s" % (time.time() - t))
warnings.simplefilter("ignore")

t0 = time.time()
model = CMF(n_components=k, solver="mu", max_iter=iters, tol=0, random_state=0)
model.fit(X, Y)
total = time.time() - t0
print("CMF.fit end to end (cold: includes library load, first-launch code upload)   %.3f s" % total)
t0 = time.time()
model = CMF(n_components=k, solver="mu", max_iter=iters, tol=0, random_state=0)
model.fit(X, Y)
total = time.time() - t0
print("CMF.fit end to end (warm) %.3f s   n_iter %d  err %.4f" % (total, model.n_iter_, model.reconstruction_err_))

# the same steps one by one
t = time.time(); Xc = check_array(X, accept_sparse=("csr", "csc"), dtype=float); Yc = check_array(Y, accept_sparse=("csr", "csc"), dtype=float)
print("  check_array (finite scan, no copy)        %.3f s" % (time.time() - t))
s = HipMUSolver(max_iter=iters, tol=0, random_state=0)
t = time.time(); ctx = s.bind_data(Xc, Yc, k); ctx.sync(); print("  upload X, Y (float64 host -> float32 HBM) %.3f s  = %.1f GB/s of float64 source" % (time.time() - t, (X.nbytes + Y.nbytes) / 1e9 / (time.time() - t)))
opx, opy = DeviceOperand(ctx, 0, X.shape), DeviceOperand(ctx, 1, Y.shape)
t = time.time(); U, V = initialize_mf(Xc, k, init="nndsvdar", random_state=0, non_negative=True, operand=opx); tx = time.time() - t
t = time.time(); V2, Z = initialize_mf(Yc, k, init="nndsvdar", random_state=0, non_negative=True, operand=opy); ty = time.time() - t
print("  initialisers nndsvdar: X %.3f s, Y %.3f s" % (tx, ty))
V = (V + V2) / 2
U, V, Z = (np.array(F, dtype=np.float64, order="C") for F in (U, V, Z))
t = time.time(); s.fit_iterative_update(Xc, Yc, U, V, Z); print("  solver loop (%d iterations) + factor round trip   %.3f s" % (iters, time.time() - t))
s.release()
