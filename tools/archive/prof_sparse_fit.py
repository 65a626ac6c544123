import cProfile, pstats, sys, os, io
import numpy as np
from scipy.sparse import csr_matrix
sys.path.insert(0, "/root/repo")
from pycmf_amd import CMF
rng = np.random.mtrand.RandomState(42)
X = np.abs(rng.randn(2000, 150)); X[:1000, 2 * np.arange(10) + 100] = 0; X[1000:, 2 * np.arange(10)] = 0
Xs = csr_matrix(X); Y = np.abs(rng.randn(150, 10))
for _ in range(2):
    CMF(n_components=10, random_state=42, max_iter=10, solver="mu").fit_transform(Xs, Y)
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    CMF(n_components=10, random_state=42, max_iter=10, solver="mu").fit_transform(Xs, Y)
pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:5000])
