"""Randomised small-shape sweep through the C ABI against the fp64 oracle: ragged extents (1 .. 70, nothing a
multiple of a tile), k from 1, every link pair, sampled and unsampled, partial update masks, dense and CSR."""
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from pycmf_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("no GPU visible: the gpu-marked tests need an MI355X")
    return _lib


@pytest.mark.parametrize("seed", range(12))
def test_mu_random_shapes(lib, seed):
    from oracle import cmf_oracle as O
    rng = np.random.RandomState(100 + seed)
    m, d, p = (int(v) for v in rng.randint(1, 70, size=3))
    k = int(rng.randint(1, 10))
    l1, l2 = (0.0, 0.0) if seed % 2 else (float(rng.rand() * 0.3), float(rng.rand() * 0.3))
    mask = int(rng.randint(1, 8))
    X, Y = np.abs(rng.randn(m, d)), np.abs(rng.randn(d, p))
    if seed % 3 == 0:
        X[rng.rand(m, d) < 0.6] = 0.0
    U, V, Z = np.abs(rng.randn(m, k)) + 0.05, np.abs(rng.randn(d, k)) + 0.05, np.abs(rng.randn(p, k)) + 0.05
    ctx = lib.Context(0)
    ctx.set_problem(m, d, p, k)
    ctx.set_data(0, sp.csr_matrix(X) if seed % 3 == 0 else X)
    ctx.set_data(1, Y)
    for w, F in enumerate((U, V, Z)):
        ctx.set_factor(w, F)
    for _ in range(3):
        ctx.mu_step(l1, l2, mask)
        O.mu_update_step(X, Y, U, V, Z, l1, l2, update_U=bool(mask & 1), update_V=bool(mask & 2), update_Z=bool(mask & 4))
    for w, ref in enumerate((U, V, Z)):
        np.testing.assert_allclose(ctx.get_factor(w), ref, rtol=3e-4, atol=1e-6 * max(1.0, np.abs(ref).max()))
    ex, ey = ctx.residual_sq()
    np.testing.assert_allclose([ex, ey], [np.linalg.norm(X - U @ V.T) ** 2, np.linalg.norm(Y - V @ Z.T) ** 2], rtol=1e-3,
                               atol=1e-6)
    ctx.close()


@pytest.mark.parametrize("seed", range(12))
def test_newton_random_shapes(lib, seed):
    from oracle import cmf_oracle as O
    rng = np.random.RandomState(200 + seed)
    m, d, p = (int(v) for v in rng.randint(2, 60, size=3))
    k = int(rng.randint(1, 9))
    xl, yl = [("linear", "linear"), ("logit", "linear"), ("linear", "logit"), ("logit", "logit")][seed % 4]
    ratio = [1.0, 0.5, 0.8][seed % 3]
    nn = int(rng.randint(0, 8))
    mask = 7 if seed < 8 else int(rng.randint(1, 8))
    alpha, l1, l2, pert = float(0.2 + 0.6 * rng.rand()), float(rng.rand() * 0.1), float(rng.rand() * 0.3), 0.2
    X = rng.rand(m, d) if xl == "logit" else np.abs(rng.randn(m, d))
    Y = rng.rand(d, p) if yl == "logit" else np.abs(rng.randn(d, p))
    U0, V0, Z0 = 0.4 * rng.randn(m, k), 0.4 * rng.randn(d, k), 0.4 * rng.randn(p, k)
    if nn & 1: U0 = np.abs(U0)
    if nn & 2: V0 = np.abs(V0)
    if nn & 4: Z0 = np.abs(Z0)
    np.random.seed(seed)
    masks = {"U": [], "Z": [], "V": []}
    Ur, Vr, Zr = U0.copy(), V0.copy(), Z0.copy()
    O.newton_update_step(X, Y, Ur, Vr, Zr, alpha, l1, l2, xl, yl, bool(nn & 1), bool(nn & 2), bool(nn & 4), ratio, pert,
                         update_U=bool(mask & 1), update_V=bool(mask & 2), update_Z=bool(mask & 4), masks=masks)
    ctx = lib.Context(0)
    ctx.set_problem(m, d, p, k)
    ctx.set_data(0, X); ctx.set_data(1, Y)
    for w, F in enumerate((U0, V0, Z0)):
        ctx.set_factor(w, F)
    lists = [None] * 4
    if ratio < 1:
        def arr(rows, per):
            return np.zeros((rows, per), dtype=np.int32)
        su, sm, sp_ = int(d * ratio), int(m * ratio), int(p * ratio)
        lists = [np.array(masks["U"]).reshape(-1, su) if mask & 1 else arr(m, su),
                 np.array(masks["Z"]).reshape(-1, su) if mask & 4 else arr(p, su),
                 np.array([a for a, _ in masks["V"]]).reshape(-1, sm) if mask & 2 else arr(d, sm),
                 np.array([b for _, b in masks["V"]]).reshape(-1, sp_) if mask & 2 else arr(d, sp_)]
    ctx.newton_step(alpha, l1, l2, xl, yl, nn, mask, pert, ratio, *lists)
    for w, ref in enumerate((Ur, Vr, Zr)):
        np.testing.assert_allclose(ctx.get_factor(w), ref, rtol=3e-3, atol=3e-3 * max(1e-3, np.abs(ref).max()))
    ctx.close()


@pytest.mark.parametrize("alpha", [0.0, 1.0, 0.3])
@pytest.mark.parametrize("l1,l2", [(0.0, 0.0), (0.1, 0.0), (0.0, 0.5), (0.05, 0.1)])
@pytest.mark.parametrize("k", [5, 70])
def test_linear_newton_edge_weights(lib, alpha, l1, l2, k):
    """The shared (re-associated) sweeps at the edges of their parameter range: alpha = 0 / 1 remove one matrix from a sweep
    altogether (H = l2 I, or exactly zero when l2 = 0 too: every eigenvalue clamps to the perturbation), l2 = 0 leaves
    rank-deficient Grams, l1 adds the sign term.  Two iterations against the float64 oracle (pycmf/cmf_solvers.py:394-522)."""
    from oracle import cmf_oracle as O
    rng = np.random.RandomState(int(100 * alpha) + k)
    m, d, p = 90, 75, 40
    X, Y = np.abs(rng.randn(m, d)), np.abs(rng.randn(d, p))
    U0, V0, Z0 = 0.3 * rng.randn(m, k), 0.3 * rng.randn(d, k), 0.3 * rng.randn(p, k)
    U, V, Z = U0.copy(), V0.copy(), Z0.copy()
    ctx = lib.Context(0)
    ctx.set_problem(m, d, p, k)
    ctx.set_data(0, X); ctx.set_data(1, Y)
    for w, F in enumerate((U0, V0, Z0)):
        ctx.set_factor(w, F)
    for _ in range(2):
        O.newton_update_step(X, Y, U, V, Z, alpha, l1, l2, "linear", "linear", False, False, False, 1.0, 0.2)
        ctx.newton_step(alpha, l1, l2, "linear", "linear", 0, 7, 0.2, 1.0)
    for w, ref in enumerate((U, V, Z)):
        np.testing.assert_allclose(ctx.get_factor(w), ref, rtol=0, atol=2e-4 * max(1e-3, np.abs(ref).max()))
    ctx.close()
