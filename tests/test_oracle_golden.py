"""Pin the CPU oracle (oracle/cmf_oracle.py) against vectors minted from the
genuine reference (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import scipy.sparse as sp

from oracle import cmf_oracle as O
from conftest import load_golden

TOL = dict(rtol=1e-10, atol=1e-12)


@pytest.mark.parametrize("tag,l1,l2", [("plain", 0.0, 0.0), ("reg", 0.3, 0.7)])
@pytest.mark.parametrize("fmt", ["dense", "csr"])
def test_mu_steps_match_reference(tag, l1, l2, fmt):
    g = load_golden("g2_mu_steps")
    X = sp.csr_matrix(g["X"]) if fmt == "csr" else g["X"]
    U, V, Z = g["U0"].copy(), g["V0"].copy(), g["Z0"].copy()
    for it in range(1, 11):
        O.mu_update_step(X, g["Y"], U, V, Z, l1, l2)
        if it in (1, 10):
            for n, a in (("U", U), ("V", V), ("Z", Z)):
                np.testing.assert_allclose(a, g["%s_%s_%s%d" % (tag, fmt, n, it)], **TOL)


def test_mu_signed_zero_den_and_partial():
    g = load_golden("g2_mu_steps")
    U, V, Z = g["sU0"].copy(), g["sV0"].copy(), g["sZ0"].copy()
    O.mu_update_step(g["sX"], g["sY"], U, V, Z)
    for n, a in (("U", U), ("V", V), ("Z", Z)):
        np.testing.assert_allclose(a, g["s%s1" % n], **TOL)
    U, V, Z = g["zU0"].copy(), g["zV0"].copy(), g["zZ0"].copy()
    O.mu_update_step(g["zX"], g["zY"], U, V, Z)
    for n, a in (("U", U), ("V", V), ("Z", Z)):
        np.testing.assert_allclose(a, g["z%s1" % n], **TOL)
    U, V, Z = g["U0"].copy(), g["V0"].copy(), g["Z0"].copy()
    O.mu_update_step(g["X"], g["Y"], U, V, Z, update_V=False, update_Z=False)
    for n, a in (("U", U), ("V", V), ("Z", Z)):
        np.testing.assert_allclose(a, g["p%s1" % n], **TOL)


NEWTON_CASES = {
    "lin_lin_nn": ("linear", "linear", True, 1.0, None, 0.1, 0.2, False),
    "lin_log_nn": ("linear", "logit", True, 1.0, None, 0.1, 0.2, False),
    "log_log_free": ("logit", "logit", False, 1.0, None, 0.1, 0.2, True),
    "log_lin_free": ("logit", "linear", False, 1.0, None, 0.0, 0.0, True),
    "lin_log_free_sg": ("linear", "logit", False, 0.5, 3, 0.1, 0.2, True),
    "log_log_nn_sg": ("logit", "logit", True, 0.5, 5, 0.05, 0.1, False),
    "lin_lin_free_sg": ("linear", "linear", False, 0.5, 9, 0.0, 0.3, True),
}


@pytest.mark.parametrize("name", sorted(NEWTON_CASES))
@pytest.mark.parametrize("fmt", ["dense", "csr"])
def test_newton_steps_match_reference(name, fmt):
    xl, yl, nn, ratio, seed, l1, l2, signed = NEWTON_CASES[name]
    g = load_golden("g3_newton_steps")
    X = g["Xlog"] if xl == "logit" else g["X"]
    Y = g["Ylog"] if yl == "logit" else g["Y"]
    if fmt == "csr":
        X = sp.csr_matrix(X)
    sfx = "s" if signed else "p"
    U, V, Z = g["U0" + sfx].copy(), g["V0" + sfx].copy(), g["Z0" + sfx].copy()
    s = O.OracleSolver("newton", alpha=0.3, l1_reg=l1, l2_reg=l2, x_link=xl, y_link=yl,
                       U_non_negative=nn, V_non_negative=nn, Z_non_negative=nn,
                       hessian_pertubation=0.2, sg_sample_ratio=ratio, random_state=seed)
    for it in range(1, 4):
        s.update_step(X, Y, U, V, Z)
        if it in (1, 3):
            for n, a in (("U", U), ("V", V), ("Z", Z)):
                np.testing.assert_allclose(a, g["%s_%s_%s%d" % (name, fmt, n, it)],
                                           rtol=1e-8, atol=1e-10)


def test_newton_sample_draw_order():
    """The oracle draws its samples in the reference's order (U rows, Z rows,
    then per V row: U-sample, Z-sample)."""
    g = load_golden("g3_newton_steps")
    np.random.seed(3)
    masks = {"U": [], "Z": [], "V": []}
    U, V, Z = g["U0s"].copy(), g["V0s"].copy(), g["Z0s"].copy()
    O.newton_update_step(g["X"], g["Ylog"], U, V, Z, 0.3, 0.1, 0.2, "linear", "logit",
                         False, False, False, ratio=0.5, pert=0.2, masks=masks)
    flat = [*masks["U"], *masks["Z"]]
    for su, sz in masks["V"]:
        flat += [su, sz]
    flat = np.concatenate(flat)
    ref = g["lin_log_free_sg_draws"]
    np.testing.assert_array_equal(flat, ref[: len(flat)])


@pytest.mark.parametrize("solver", ["mu", "newton"])
def test_fit_level_matches_reference(solver):
    g = load_golden("g4_fit_level")
    U, V, Z = g["fc_U0"].copy(), g["fc_V0"].copy(), g["fc_Z0"].copy()
    X, Y = g["fc_X"], g["fc_Y"]
    # driver semantics with custom init: V = (V + V)/2, mu alpha 0.5, newton 'auto'
    alpha = 0.5 if solver == "mu" else Y.shape[1] / (X.shape[0] + Y.shape[1])
    s = O.OracleSolver(solver, max_iter=1000, tol=1e-4, alpha=alpha, random_state=0)
    U, V, Z, n_iter = s.fit_iterative_update(X, Y, U, V, Z)
    assert n_iter == int(g["fc_%s_n_iter" % solver])
    err = O.factorization_error(X, U, V.T, "linear") + O.factorization_error(Y, V, Z.T, "linear")
    np.testing.assert_allclose(err, float(g["fc_%s_err" % solver]), rtol=1e-8)
    np.testing.assert_allclose(U, g["fc_%s_U" % solver], rtol=1e-6, atol=1e-9)


def test_error_metric_sparse_equals_dense():
    rng = np.random.RandomState(0)
    X = np.abs(rng.randn(20, 15))
    X[X < 0.5] = 0
    U, V = np.abs(rng.randn(20, 4)), np.abs(rng.randn(15, 4))
    a = O.factorization_error(X, U, V.T, "linear")
    b = O.factorization_error(sp.csr_matrix(X), U, V.T, "linear")
    np.testing.assert_allclose(a, b, rtol=1e-12)
    np.testing.assert_allclose(a, np.linalg.norm(X - U @ V.T), rtol=1e-12)


def test_readme_case():
    g = load_golden("g1_readme")
    assert int(g["n_iter"]) > 0 and float(g["err"]) < 0.1


CYTHON_CASES = {"lin_log_nn": ("linear", "logit", True, 1.0, None, 0.1, 0.2, False),
                "log_log_free": ("logit", "logit", False, 1.0, None, 0.1, 0.2, True),
                "lin_log_free_sg": ("linear", "logit", False, 0.5, 3, 0.1, 0.2, True)}


@pytest.mark.parametrize("name", sorted(CYTHON_CASES))
def test_cython_twin_steps_match_compiled_reference(name):
    """SURVEY N-cy rows: the oracle's ``cython_variant`` flag reproduces the reference's Cython module
    (compiled out of tree for the fixture): same as the live path except Z's logit Hessian has no l2."""
    xl, yl, nn, ratio, seed, l1, l2, signed = CYTHON_CASES[name]
    g, c = load_golden("g3_newton_steps"), load_golden("g7_cython_steps")
    X = g["Xlog"] if xl == "logit" else g["X"]
    Y = g["Ylog"] if yl == "logit" else g["Y"]
    sfx = "s" if signed else "p"
    U, V, Z = g["U0" + sfx].copy(), g["V0" + sfx].copy(), g["Z0" + sfx].copy()
    if seed is not None:
        np.random.seed(seed)
    O.newton_update_step(X, Y, U, V, Z, 0.3, l1, l2, xl, yl, nn, nn, nn, ratio=ratio, pert=0.2, cython_variant=True)
    for n, a in (("U", U), ("V", V), ("Z", Z)):
        np.testing.assert_allclose(a, c["%s_%s1" % (name, n)], rtol=1e-8, atol=1e-10)
    # and it really differs from the live path where l2 > 0
    assert np.abs(Z - g["%s_dense_Z1" % name]).max() > 1e-3
