"""The float32 limit of the per-row spectral clamp, made visible: `cmf_newton_clamp_stats` counts the row Hessians whose smallest
eigenvalue was below the perturbation and keeps the largest ||H||_F / pert among them; `HipNewtonSolver` warns when that ratio
leaves the range in which the stated tolerances hold (tools/fuzz_campaign.py).  Reference: `_safe_invert`,
pycmf/cmf_solvers.py:346-356, on float64 Hessians."""
import warnings

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from pycmf_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("no GPU visible: the gpu-marked tests need an MI355X")
    return _lib


def _problem(seed, m, d, p, k):
    rng = np.random.RandomState(seed)
    X = np.abs(rng.randn(m, d))
    Y = (rng.rand(d, p) < 0.3).astype(float)
    sc = 0.4 / np.sqrt(max(1.0, k / 8.0))
    return X, Y, sc * rng.randn(m, k), np.abs(sc * rng.randn(d, k)), sc * rng.randn(p, k)


def test_clamp_stats_follow_the_conditioning(lib):
    from oracle import cmf_oracle as O
    # (a) l2 >= pert: every Hessian is certified positive definite above the threshold, the clamp never acts
    X, Y, U, V, Z = _problem(0, 60, 50, 40, 8)
    ctx = lib.Context(0)
    ctx.set_problem(60, 50, 40, 8)
    ctx.set_data(0, X); ctx.set_data(1, Y)
    for w, F in enumerate((U, V, Z)):
        ctx.set_factor(w, F)
    ctx.newton_step(0.5, 0.0, 0.5, "linear", "logit", 0, 7, 0.2, 1.0)
    assert ctx.newton_clamp_stats() == (0, 0.0)
    ctx.close()
    # (b) more components than samples and no l2: rank-deficient Hessians, every V row is clamped; after the U sweep's 1 / pert
    # steps ||H|| / pert is ~1e5 -- beyond what float32 Hessians resolve: the record says so, and the factors of the
    # well-conditioned sweeps (U, Z) still agree with the float64 oracle
    m, d, p, k = 40, 103, 2, 100
    X, Y, U, V, Z = _problem(1, m, d, p, k)
    ctx = lib.Context(0)
    ctx.set_problem(m, d, p, k)
    ctx.set_data(0, X); ctx.set_data(1, Y)
    for w, F in enumerate((U, V, Z)):
        ctx.set_factor(w, F)
    ctx.newton_step(0.75, 2.0, 0.0, "linear", "logit", 2, 7, 0.2, 1.0)
    rows, ratio = ctx.newton_clamp_stats(reset=True)
    assert rows >= d and ratio > 1e4
    assert ctx.newton_clamp_stats() == (0, 0.0)
    Ur, Vr, Zr = U.copy(), V.copy(), Z.copy()
    O.newton_update_step(X, Y, Ur, Vr, Zr, 0.75, 2.0, 0.0, "linear", "logit", False, True, False, 1.0, 0.2)
    for w, ref in ((0, Ur), (2, Zr)):
        np.testing.assert_allclose(ctx.get_factor(w), ref, rtol=0, atol=1e-4 * np.abs(ref).max())
    # the ill-conditioned sweep itself: within a few percent (float32 Hessians; the float64 reference resolves it)
    assert np.abs(ctx.get_factor(1) - Vr).max() < 0.1 * np.abs(Vr).max()
    ctx.close()


def test_solver_warns_when_the_clamp_leaves_the_float32_range():
    from pycmf_amd.solver_shell import HipNewtonSolver
    m, d, p, k = 40, 103, 2, 100
    X, Y, U, V, Z = _problem(1, m, d, p, k)
    kw = dict(max_iter=1, tol=0, alpha=0.75, x_link="linear", y_link="logit", U_non_negative=False, V_non_negative=True,
              Z_non_negative=False, hessian_pertubation=0.2)
    s = HipNewtonSolver(l1_reg=2.0, l2_reg=0.0, **kw)
    with pytest.warns(RuntimeWarning, match="float32"):
        s.fit_iterative_update(X, Y, U.copy(), V.copy(), Z.copy())
    assert s.clamped_rows_ >= d and s.clamp_ratio_ > HipNewtonSolver.CLAMP_RATIO_WARN
    # the same problem with l2 >= pert: silent
    s = HipNewtonSolver(l1_reg=2.0, l2_reg=0.5, **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        s.fit_iterative_update(X, Y, U.copy(), V.copy(), Z.copy())
    assert s.clamped_rows_ == 0
