"""End-to-end fit time at C2 size and where it goes: host-only initialiser vs device-assisted."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pycmf_amd import CMF, _lib
from pycmf_amd import factor_init as FI
rng = np.random.RandomState(0)
m, d, p, k = 16384, 8192, 4096, 128
X = np.abs(rng.randn(m, d)).astype(np.float64); Y = np.abs(rng.randn(d, p)).astype(np.float64)
warnings.simplefilter("ignore")
for thr, label in ((10**18, "host sklearn randomized_svd"), (4_000_000, "device-assisted randomized_svd")):
    FI.DEVICE_SVD_MIN_CELLS = thr
    import pycmf_amd.estimator as E
    E.DEVICE_SVD_MIN_CELLS = thr
    t0 = time.time()
    mdl = CMF(n_components=k, solver="mu", random_state=0, max_iter=100, tol=0)
    mdl.fit(X, Y)
    print("%-34s total fit %.2f s (100 MU iterations), err %.3f" % (label, time.time() - t0, mdl.reconstruction_err_), flush=True)
ctx = _lib.Context(0); ctx.set_problem(m, d, p, k)
t0 = time.time(); ctx.set_data(0, X); ctx.set_data(1, Y); print("upload X,Y (fp64->fp32 + PCIe): %.2f s" % (time.time() - t0))
op = FI.DeviceOperand(ctx, 0, X.shape)
t0 = time.time(); FI.randomized_svd_device(op, k, random_state=0); print("device-assisted rsvd(X): %.2f s" % (time.time() - t0))
from sklearn.utils.extmath import randomized_svd
t0 = time.time(); randomized_svd(X, k, random_state=0); print("sklearn rsvd(X) on host: %.2f s" % (time.time() - t0))
