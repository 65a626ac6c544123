"""End-to-end `fit_transform` wall time on the six workloads the reference's own benchmark prints
(benchmarks/benchmark_cmf.py:42-82 upstream: 2000 x 150 / 150 x 10, n_components=10, max_iter=10), here through
pycmf_amd.CMF.  BASELINE.md section 2 holds the reference's times on 8 vCPU (0.23 .. 0.50 s; 28.1 s at ratio 0.2).
usage: python tools/reference_cases.py [repeats]"""
import os, sys, time
import numpy as np
from scipy.sparse import csr_matrix
from scipy.special import expit
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pycmf_amd import CMF

repeats = int(sys.argv[1]) if len(sys.argv) > 1 else 3


def data(sparse, logits):
    rng = np.random.mtrand.RandomState(42)
    X = np.abs(rng.randn(2000, 150))
    if sparse:
        X[:1000, 2 * np.arange(10) + 100] = 0
        X[1000:, 2 * np.arange(10)] = 0
        X = csr_matrix(X)
    Y = rng.randn(150, 10)
    return X, (expit(Y) if logits else np.abs(Y))


cases = [("dense, mu", False, False, dict(solver="mu")),
         ("dense, newton", False, False, dict(solver="newton")),
         ("sparse, mu", True, False, dict(solver="mu")),
         ("sparse, newton", True, False, dict(solver="newton")),
         ("sparse, Y logits, newton, ratio 1.0", True, True, dict(solver="newton", sg_sample_ratio=1.0)),
         ("sparse, Y logits, newton, ratio 0.2", True, True, dict(solver="newton", sg_sample_ratio=0.2)),
         ("sparse, Y logits, newton, ratio 0.2, device sampler", True, True,
          dict(solver="newton", sg_sample_ratio=0.2, sg_sampler="device"))]
for name, sparse, logits, kw in cases:
    X, Y = data(sparse, logits)
    best = 1e9
    for _ in range(repeats):
        t0 = time.perf_counter()
        model = CMF(n_components=10, random_state=42, max_iter=10, **kw)
        U, V, Z = model.fit_transform(X, Y)
        best = min(best, time.perf_counter() - t0)
    print("%-55s best of %d: %.3f s   (n_iter_ %d, err %.4f)" % (name, repeats, best, model.n_iter_, model.reconstruction_err_))


# ---- the reference's published workloads (samples/toxic_comments.ipynb): X = binary bag-of-words CSR
# 9927 x 10000 (words x comments), Y = 10000 x 6 labels, n_components = 20.  The corpus itself is not available offline: same
# shape, binary values, ~30 distinct words per comment, 10 % positive labels.  Published: mu 50 iterations in 36.267 s
# (1.38 it/s); newton (x linear / y logit, l1 = 2, l2 = 5, U and V non-negative) 10 iterations in 233.713 s (0.043 it/s).
rng = np.random.RandomState(0)
m, d, p, k = 9927, 10000, 6, 20
cols = np.concatenate([np.sort(rng.choice(m, 30, replace=False)) for _ in range(d)])
Xt = csr_matrix((np.ones(30 * d), cols, np.arange(0, 30 * d + 1, 30)), shape=(d, m))
X = csr_matrix(Xt.T)
Y = (rng.rand(d, p) < 0.1).astype(np.float64)
for name, kw, iters, published in (("toxic-comments shape, mu, 50 iterations", dict(solver="mu", max_iter=50, tol=0), 50, 36.267),
                                   ("toxic-comments shape, newton, y logit, l1=2, l2=5, 10 iterations",
                                    dict(solver="newton", x_link="linear", y_link="logit", l1_reg=2., l2_reg=5., alpha=0.14, max_iter=10, tol=0,
                                         U_non_negative=True, V_non_negative=True, Z_non_negative=False), 10, 233.713)):
    best = 1e9
    for _ in range(repeats):
        t0 = time.perf_counter()
        model = CMF(n_components=k, random_state=0, **kw)
        model.fit_transform(X, Y)
        best = min(best, time.perf_counter() - t0)
    print("%-75s best of %d: %.3f s = %.1f it/s end to end (init + upload + %d iterations); published %.3f s = %.3f it/s; n_iter_ %d err %.4f"
          % (name, repeats, best, iters / best, iters, published, iters / published, model.n_iter_, model.reconstruction_err_))
