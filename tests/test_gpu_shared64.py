"""float64 treatment of the shared Hessian of the linear-link Newton sweeps (csrc/cmf_shared64.hip.h) and the
residual-level parity criterion of north_star: relative reconstruction residuals within 1e-4 of the CPU reference after
8 iterations (reference loop: pycmf/cmf_solvers.py:510-522; _safe_invert :346-356)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from pycmf_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("no GPU visible: the gpu-marked tests need an MI355X")
    return _lib


def _safe_invert(M, pert):
    """pycmf/cmf_solvers.py:346-356"""
    lam, Q = np.linalg.eigh(M)
    lam = np.abs(lam)
    lam[lam < pert] = pert
    return (Q / lam) @ Q.T


@pytest.mark.parametrize("k", [3, 20, 64, 65, 128, 200, 256, 384])
@pytest.mark.parametrize("mode", ["pd", "clamped", "near"])
def test_safe_invert_f64_matches_eigh(lib, k, mode):
    """One symmetric positive semi-definite matrix through the float64 route: Cholesky inverse when lambda_min >= pert,
    eigenvalue clamp (Jacobi for k <= 64, spectral clamp by float64 matrix polynomials above) otherwise."""
    rng = np.random.RandomState(k)
    pert = 0.2
    if mode == "pd":            # Gram of a tall factor + l2: every eigenvalue far above the perturbation
        F = rng.randn(5 * k + 50, k)
        H = 0.5 * F.T @ F + 0.3 * np.eye(k)
    elif mode == "clamped":     # rank-deficient Gram (zero columns after non-negative clamping) + small l2
        F = np.abs(rng.randn(max(2, k // 3), k))
        F[:, ::5] = 0.0
        H = 0.7 * F.T @ F + 0.01 * np.eye(k)
    else:                       # a spectrum that straddles the threshold, eigenvalues on both sides close to it
        Q, _ = np.linalg.qr(rng.randn(k, k))
        lam = np.concatenate([np.linspace(0.05, 0.1999, k // 2), np.linspace(0.2001, 40.0, k - k // 2)])
        H = (Q * lam) @ Q.T
        H = 0.5 * (H + H.T)
    want = _safe_invert(H, pert)
    ctx = lib.Context(0)
    ctx.set_problem(4, 4, 4, k)
    got = ctx.safe_invert_f64(H, pert)
    ctx.close()
    # the result is handed back after its rounding to float32 (6e-8 relative per entry)
    np.testing.assert_allclose(got, want, rtol=0, atol=2e-6 * np.abs(want).max())


def _make_problem(case, rng, plain=False):
    m, d, p, k, xl, yl, l2, nn, signed = case
    if plain:   # the shape of the reference's own stochastic tests: |N(0,1)| / U(0,1) data, small signed factors
        X = rng.rand(m, d) if xl == "logit" else np.abs(rng.randn(m, d))
        Y = rng.rand(d, p) if yl == "logit" else np.abs(rng.randn(d, p))
        return X, Y, 0.3 * rng.randn(m, k), 0.3 * rng.randn(d, k), 0.3 * rng.randn(p, k)
    Ut, Vt, Zt = np.abs(rng.randn(m, 8)), np.abs(rng.randn(d, 8)), np.abs(rng.randn(p, 8))
    X = Ut @ Vt.T + 0.1 * np.abs(rng.randn(m, d))
    Y = Vt @ Zt.T + 0.1 * np.abs(rng.randn(d, p))
    if signed:
        X, Y = X - X.mean(), Y - Y.mean()
    if xl == "logit":
        X = 1 / (1 + np.exp(-(X - X.mean()) / X.std()))
    if yl == "logit":
        Y = 1 / (1 + np.exp(-(Y - Y.mean()) / Y.std()))
    sc = np.sqrt(np.abs(X).mean() / k)
    if signed:
        U0, V0, Z0 = sc * rng.randn(m, k), sc * rng.randn(d, k), sc * rng.randn(p, k)
    else:
        U0, V0, Z0 = sc * np.abs(rng.randn(m, k)), sc * np.abs(rng.randn(d, k)), sc * np.abs(rng.randn(p, k))
    return X, Y, U0, V0, Z0


PARITY_TOL = {}   # north_star's 1e-4 for every case
PARITY_CASES = {
    # (m, d, p, k, x_link, y_link, l2, non_negative, signed data), sg_sample_ratio
    "linear_signed": ((700, 500, 300, 48, "linear", "linear", 0.0, False, True), 1.0),
    # non-negative clamping: with l2 = 0 the reference's own iteration is erratic (error-increasing from iteration 3 on in
    # float64), so the contract is stated on a run that the reference itself converges on.  cond(H) = 1e4 here and the
    # hard clamp at 0 turns rounding into different active sets: the gradient form F - grad Hinv measured 1.7e-3 on the
    # residuals (float32 rounding of X V / X^T U times cond(H)); the re-associated form F E + T (O Hinv) -- float64 inverse
    # applied to the other factor BEFORE the float32 data pass -- measures 5e-7 (tests/tools/emul_newton_precision.py)
    "linear_nonneg": ((900, 700, 300, 32, "linear", "linear", 1.0, True, False), 1.0),
    "linear_logit_ratio05": ((260, 200, 120, 24, "linear", "logit", 0.05, False, False), 0.5),
    "logit_logit": ((260, 200, 120, 24, "logit", "logit", 0.01, False, False), 1.0),
}


@pytest.mark.parametrize("name", sorted(PARITY_CASES))
def test_newton_residual_parity_8_iterations(lib, name):
    """north_star's own criterion: after 8 full Newton iterations from the same start (and, for sg_sample_ratio < 1,
    the same NumPy sample stream) both relative residuals agree with the float64 CPU oracle to 1e-4 relative."""
    from oracle import cmf_oracle as O
    from pycmf_amd.solver_shell import HipNewtonSolver
    case, ratio = PARITY_CASES[name]
    m, d, p, k, xl, yl, l2, nn, signed = case
    X, Y, U0, V0, Z0 = _make_problem(case, np.random.RandomState(11), plain=ratio < 1)
    kw = dict(alpha=0.5, l2_reg=l2, x_link=xl, y_link=yl, U_non_negative=nn, V_non_negative=nn, Z_non_negative=nn,
              sg_sample_ratio=ratio, max_iter=8, tol=0)
    g = HipNewtonSolver(random_state=4, **kw)
    Ug, Vg, Zg = U0.copy(), V0.copy(), Z0.copy()
    g.fit_iterative_update(X, Y, Ug, Vg, Zg)
    g.release()
    o = O.OracleSolver("newton", random_state=4, **kw)
    Uo, Vo, Zo = U0.copy(), V0.copy(), Z0.copy()
    o.fit_iterative_update(X, Y, Uo, Vo, Zo)
    for T, L, R, Lo, Ro, link in ((X, Ug, Vg, Uo, Vo, xl), (Y, Vg, Zg, Vo, Zo, yl)):
        eg = O.factorization_error(T, L, R.T, link) / np.linalg.norm(T)
        eo = O.factorization_error(T, Lo, Ro.T, link) / np.linalg.norm(T)
        assert abs(eg - eo) <= PARITY_TOL.get(name, 1e-4) * eo, (name, eg, eo)


def test_clamped_shared_hessian_tracks_the_oracle(lib):
    """The case round 1 could not follow: non-negative factors with l2 = 0 leave the shared Hessians
    alpha V^T V etc. with eigenvalues under the perturbation and a condition number of 1e5, so every step goes through
    the eigenvalue clamp of _safe_invert and the reference's own iterates are erratic.  What remains between the device
    and the oracle with float64 Grams and a float64 clamp is the float32 rounding of the gradient products
    (1e-7 |X^T U| per entry, multiplied by up to 1 / pert in the clamped directions): measured 0.018 after three
    iterations against 0.17 with the float32 treatment of the Hessian (option off)."""
    from oracle import cmf_oracle as O
    rng = np.random.RandomState(3)
    m, d, p, k = 800, 600, 200, 64
    Ut, Vt, Zt = np.abs(rng.randn(m, 5)), np.abs(rng.randn(d, 5)), np.abs(rng.randn(p, 5))
    X, Y = Ut @ Vt.T, Vt @ Zt.T                      # exactly rank 5: Grams of the fitted factors are nearly singular
    sc = np.sqrt(X.mean() / k)
    U0, V0, Z0 = sc * np.abs(rng.randn(m, k)), sc * np.abs(rng.randn(d, k)), sc * np.abs(rng.randn(p, k))
    Uo, Vo, Zo = U0.copy(), V0.copy(), Z0.copy()
    for _ in range(3):
        O.newton_update_step(X, Y, Uo, Vo, Zo, 0.5, 0.0, 0.0, "linear", "linear", True, True, True, 1.0, 0.2)
    errs = {}
    for f64 in (1, 0):
        ctx = lib.Context(0)
        ctx.set_option("shared_hessian_f64", f64)
        ctx.set_problem(m, d, p, k)
        ctx.set_data(0, X); ctx.set_data(1, Y)
        for w, F in enumerate((U0, V0, Z0)):
            ctx.set_factor(w, F)
        for _ in range(3):
            ctx.newton_step(0.5, 0.0, 0.0, "linear", "linear", 7, 7, 0.2, 1.0)
        got = [ctx.get_factor(w) for w in range(3)]
        ctx.close()
        errs[f64] = max(np.abs(a - b).max() / np.abs(b).max() for a, b in zip(got, (Uo, Vo, Zo)))
    assert errs[1] < 0.05, errs
    assert errs[1] < errs[0] / 3, errs


def test_large_n_components_shared_hessian(lib):
    """n_components > 1024 (k_pad = 1280): beyond the float64 route, the shared Hessian goes through the float32
    chip-wide Jacobi; every element of the k_pad x k_pad Hessian image must be written (a capped elementwise
    grid left rows 819.. stale)."""
    from oracle import cmf_oracle as O
    rng = np.random.RandomState(9)
    m, d, p, k = 150, 120, 24, 1280
    X, Y = np.abs(rng.randn(m, d)), np.abs(rng.randn(d, p))
    U0, V0, Z0 = 0.05 * rng.randn(m, k), 0.05 * rng.randn(d, k), 0.05 * rng.randn(p, k)
    ctx = lib.Context(0)
    ctx.set_problem(m, d, p, k)
    ctx.set_data(0, X); ctx.set_data(1, Y)
    for w, F in enumerate((U0, V0, Z0)):
        ctx.set_factor(w, F)
    ctx.newton_step(0.5, 0.0, 0.3, "linear", "linear", 0, 7, 0.2, 1.0)
    got = [ctx.get_factor(w) for w in range(3)]
    ctx.close()
    U, V, Z = U0.copy(), V0.copy(), Z0.copy()
    O.newton_update_step(X, Y, U, V, Z, 0.5, 0.0, 0.3, "linear", "linear", False, False, False, 1.0, 0.2)
    for a, b in zip(got, (U, V, Z)):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-3 * np.abs(b).max())


@pytest.mark.parametrize("k,nn", [(100, False), (200, True)])
def test_direct_step_of_unclamped_linear_sweeps(lib, k, nn):
    """Linear link, l1 = 0, shared Hessian H = s G + l2 I with lambda_min >= pert (the clamp of _safe_invert,
    pycmf/cmf_solvers.py:346-356, is the identity): F - (F H - s T O) H^-1 = s (T O) H^-1, so the sweep is one product
    (option direct_newton_step, default on for k > 64).  Same iterates as the two-product form and as the oracle, with and
    without the non-negativity clamp."""
    from oracle import cmf_oracle as O
    rng = np.random.RandomState(k)
    m, d, p = 600, 500, 260
    X, Y = np.abs(rng.randn(m, d)), np.abs(rng.randn(d, p))
    sc = (1.0 / k) ** 0.5
    U0, V0, Z0 = sc * np.abs(rng.randn(m, k)), sc * np.abs(rng.randn(d, k)), sc * np.abs(rng.randn(p, k))
    U, V, Z = U0.copy(), V0.copy(), Z0.copy()
    for _ in range(2):
        O.newton_update_step(X, Y, U, V, Z, 0.5, 0.0, 0.4, "linear", "linear", nn, nn, nn, 1.0, 0.2)
    got = {}
    for direct in (1, 0):
        ctx = lib.Context(0)
        ctx.set_option("direct_newton_step", direct)
        ctx.set_problem(m, d, p, k)
        ctx.set_data(0, X); ctx.set_data(1, Y)
        for w, F in enumerate((U0, V0, Z0)):
            ctx.set_factor(w, F)
        for _ in range(2):
            ctx.newton_step(0.5, 0.0, 0.4, "linear", "linear", 7 if nn else 0, 7, 0.2, 1.0)
        got[direct] = [ctx.get_factor(w) for w in range(3)]
        ctx.close()
    for a, b, o in zip(got[1], got[0], (U, V, Z)):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-4 * np.abs(b).max())   # the two-product form cancels F H against s T O in float32
        np.testing.assert_allclose(a, o, rtol=0, atol=2e-4 * np.abs(o).max())
