"""End-to-end contracts of the drop-in API on the GPU, mirroring the reference's own test
suite (tests/test_cmf.py of smn-ailab/PyCMF; line numbers below refer to it) plus fit-level
parity against golden values minted from the reference."""
import warnings

import numpy as np
import pytest
import scipy.sparse as sp
from sklearn.base import clone

from conftest import load_golden

pytestmark = pytest.mark.gpu
solvers = ["mu", "newton"]


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    from pycmf_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("no GPU visible: the gpu-marked tests need an MI355X")


def CMF(**kw):
    from pycmf_amd import CMF as C
    return C(**kw)


def test_input_shape_compatibility_check():  # :28-34
    X, Y = np.ones((5, 2)), np.ones((5, 2))
    msg = "Expected X.shape[1] == Y.shape[0], found X.shape = {}, Y.shape = {}".format(X.shape, Y.shape)
    with pytest.raises(ValueError) as e:
        CMF(solver='mu', beta_loss=2).fit(X, Y)
    assert msg in str(e.value)


@pytest.mark.parametrize("solver", solvers)
def test_fit_nn_output(solver):  # :38-52
    X = np.c_[5 * np.ones(5) - np.arange(1, 6), 5 * np.ones(5) + np.arange(1, 6)]
    Y = X.T.copy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for init in (None, 'nndsvd', 'nndsvda', 'nndsvdar', 'random'):
            U, V, Z = CMF(n_components=2, solver=solver, x_init=init, y_init=init, random_state=0).fit_transform(X, Y)
            assert not ((U < 0).any() or (V < 0).any() or (Z < 0).any())


@pytest.mark.parametrize("solver", solvers)
def test_fit_close(solver):  # :55-64
    rng = np.random.mtrand.RandomState(42)
    X, Y = np.abs(rng.randn(6, 5)), np.abs(rng.randn(5, 6))
    m = CMF(n_components=5, solver=solver, x_init='nndsvdar', y_init='nndsvdar', random_state=0, max_iter=1000)
    assert m.fit(X, Y).reconstruction_err_ < 0.1


@pytest.mark.parametrize("solver", solvers)
def test_fit_level_parity_with_reference(solver):
    """Same custom start as the golden run: iteration count and final error of the reference."""
    g = load_golden("g4_fit_level")
    m = CMF(n_components=5, solver=solver, x_init="custom", y_init="custom", random_state=0, max_iter=1000)
    U, V, Z = m.fit_transform(g["fc_X"], g["fc_Y"], U=g["fc_U0"].copy(), V=g["fc_V0"].copy(), Z=g["fc_Z0"].copy())
    ref_iter, ref_err = int(g["fc_%s_n_iter" % solver]), float(g["fc_%s_err" % solver])
    # measured (tests/tools/measure_parity.py): mu 210/210 iterations, error 2e-6 rel, U 7e-6; newton 730/730, error 7e-5 rel, U 1e-4
    # (round 2, gradient form of the Newton sweeps: 7e-4 / 5e-3)
    assert m.n_iter_ == ref_iter
    ex = np.linalg.norm(g["fc_X"] - U @ V.T) + np.linalg.norm(g["fc_Y"] - V @ Z.T)
    np.testing.assert_allclose(m.reconstruction_err_, ex, rtol=1e-4)
    np.testing.assert_allclose(m.reconstruction_err_, ref_err, rtol=1e-4 if solver == "mu" else 5e-4)
    np.testing.assert_allclose(U, g["fc_%s_U" % solver], rtol=0, atol=1e-4 if solver == "mu" else 1e-3)


def test_readme_smoke():
    g = load_golden("g1_readme")
    m = CMF(n_components=4, random_state=0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        U, V, Z = m.fit_transform(g["X"], g["Y"])
    assert m.n_iter_ == int(g["n_iter"])
    np.testing.assert_allclose(m.reconstruction_err_, float(g["err"]), rtol=1e-4)   # north_star: 1e-4 rel
    np.testing.assert_allclose(U, g["U"], rtol=0, atol=1e-4)


def test_n_components_greater_n_features():  # :104-109
    rng = np.random.mtrand.RandomState(42)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        CMF(n_components=15, random_state=0, tol=1e-2).fit(np.abs(rng.randn(30, 10)), np.abs(rng.randn(10, 5)))


@pytest.mark.parametrize("solver", solvers)
def test_recover_low_rank_matrix(solver):  # :112-123
    rng = np.random.mtrand.RandomState(42)
    U, V, Z = np.abs(rng.randn(10, 5)), np.abs(rng.randn(8, 5)), np.abs(rng.randn(6, 5))
    m = CMF(n_components=5, solver=solver, x_init='nndsvdar', y_init='nndsvdar', random_state=0, max_iter=1000)
    assert m.fit(U @ V.T, V @ Z.T).reconstruction_err_ < 1.0


def test_loss_decreasing():  # :126-162
    from pycmf_amd import collective_matrix_factorization
    from pycmf_amd.factor_init import initialize_mf
    rng = np.random.mtrand.RandomState(42)
    X, Y = np.abs(rng.randn(20, 15)), np.abs(rng.randn(15, 10))
    U0, V0 = initialize_mf(X, 10, init='random', random_state=42, non_negative=True)
    V0_, Z0 = initialize_mf(Y, 10, init='random', random_state=42, non_negative=True)
    U, V, Z = U0.copy(), ((V0 + V0_) / 2).copy(), Z0.copy()
    px, py = np.sum((X - U @ V.T) ** 2) / 2, np.sum((Y - V @ Z.T) ** 2) / 2
    for _ in range(30):
        U, V, Z, _ = collective_matrix_factorization(X, Y, U, V, Z, x_init='custom', y_init='custom',
                                                     n_components=10, max_iter=1, solver='mu', tol=0., random_state=0)
        lx, ly = np.sum((X - U @ V.T) ** 2) / 2, np.sum((Y - V @ Z.T) ** 2) / 2
        assert max(px - lx, py - ly) > 0
        px, py = lx, ly


@pytest.mark.parametrize("solver", solvers)
def test_l1_regularization(solver):  # :165-195
    rng = np.random.mtrand.RandomState(42)
    X, Y = np.abs(rng.randn(6, 5)), np.abs(rng.randn(5, 4))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        reg = CMF(n_components=3, solver=solver, l1_reg=2., random_state=42).fit_transform(X, Y)
        plain = CMF(n_components=3, solver=solver, l1_reg=0., random_state=42).fit_transform(X, Y)
    assert sum(F[F == 0].size for F in reg) > sum(F[F == 0].size for F in plain)


@pytest.mark.parametrize("solver", solvers)
def test_l2_regularization(solver):  # :198-217
    rng = np.random.mtrand.RandomState(42)
    X, Y = np.abs(rng.randn(6, 5)), np.abs(rng.randn(5, 4))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        plain = CMF(n_components=3, solver=solver, l2_reg=0., random_state=42).fit_transform(X, Y)
        reg = CMF(n_components=3, solver=solver, l2_reg=2., random_state=42).fit_transform(X, Y)
    for Fm, Fr in zip(plain, reg):
        assert Fm.mean() > Fr.mean()


def test_nonnegative_condition_for_newton_solver():  # :220-236
    rng = np.random.mtrand.RandomState(42)
    X, Y = np.abs(rng.randn(6, 5)), np.abs(rng.randn(5, 4))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        U, V, Z = CMF(n_components=3, solver="newton", l2_reg=0., random_state=42, U_non_negative=False,
                      V_non_negative=False, Z_non_negative=False).fit_transform(X, Y)
    assert U.min() < 0 and V.min() < 0 and Z.min() < 0


def test_logit_link_optimization():  # :239-250
    rng = np.random.mtrand.RandomState(42)
    X = 1 / (1 + np.exp(-rng.randn(6, 5)))
    Y = 1 / (1 + np.exp(-rng.randn(5, 4)))
    m = CMF(n_components=5, solver="newton", l2_reg=0., random_state=42, x_link="logit", y_link="logit",
            U_non_negative=False, V_non_negative=False, Z_non_negative=False)
    m.fit_transform(X, Y)
    assert m.reconstruction_err_ < 0.1


def test_logit_link_non_negative_optimization():  # :253-267
    rng = np.random.mtrand.RandomState(42)
    X = rng.randn(6, 5)
    X[X < 0] = 0
    Y = 1 / (1 + np.exp(-rng.randn(5, 4)))
    m = CMF(n_components=5, solver="newton", l2_reg=0., random_state=42, y_link="logit",
            U_non_negative=True, V_non_negative=True, Z_non_negative=False,
            hessian_pertubation=0.2, max_iter=1000)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.fit_transform(X, Y)
    assert m.reconstruction_err_ < 0.1


def test_logit_fit_matches_reference_error():
    """golden lg_*: x linear / y logit, signed factors, 'random' init from random_state=42."""
    g = load_golden("g4_fit_level")
    m = CMF(n_components=5, solver="newton", y_link="logit", random_state=42, max_iter=200,
            U_non_negative=False, V_non_negative=False, Z_non_negative=False)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.fit(g["lg_X"], g["lg_Y"])
    ref = float(g["lg_err"])
    assert m.n_iter_ == int(g["lg_n_iter"])
    assert abs(m.reconstruction_err_ - ref) <= 0.05 * ref + 1e-3


@pytest.mark.parametrize("solver", solvers)
def test_sparse_input(solver):  # :270-291
    rng = np.random.mtrand.RandomState(42)
    X = np.abs(rng.randn(10, 8)); X[:, 2 * np.arange(4)] = 0
    Y = np.abs(rng.randn(8, 5))
    est = CMF(n_components=4, solver=solver, random_state=0, x_init="random", y_init="random", max_iter=50, tol=0)
    a = est.fit_transform(X, Y)
    b = clone(est).fit_transform(sp.csr_matrix(X), Y)
    c = clone(est).fit_transform(sp.csc_matrix(X), sp.csr_matrix(Y))
    for F, G, H in zip(a, b, c):
        np.testing.assert_array_almost_equal(F, G, 6)
        np.testing.assert_array_almost_equal(F, H, 6)


def _sg_model():
    return CMF(n_components=5, solver="newton", x_init='svd', y_init='svd', U_non_negative=False,
               V_non_negative=False, Z_non_negative=False, alpha=0.5, sg_sample_ratio=0.5,
               random_state=0, max_iter=1000)


def test_stochastic_newton_solver():  # :292-300
    rng = np.random.mtrand.RandomState(42)
    X, Y = rng.randn(6, 5), rng.randn(5, 6)
    assert _sg_model().fit(X, Y).reconstruction_err_ < 0.1


def test_stochastic_newton_solver_sparse_input_close():  # :303-314
    rng = np.random.mtrand.RandomState(42)
    A, B = rng.randn(6, 5), rng.randn(5, 6)
    assert _sg_model().fit(sp.csr_matrix(A), sp.csr_matrix(B)).reconstruction_err_ < 0.1


def test_stochastic_newton_solver_sparse_input():  # :317-338 (dense == sparse, identical RNG stream)
    rng = np.random.mtrand.RandomState(36)
    A = np.abs(rng.randn(10, 10)); A[:, 2 * np.arange(5)] = 0
    B = np.abs(rng.randn(10, 5)); B[2 * np.arange(5), :] = 0
    est1 = CMF(n_components=5, solver="newton", x_init='svd', y_init='svd', U_non_negative=False,
               V_non_negative=False, Z_non_negative=False, sg_sample_ratio=0.5, random_state=0, max_iter=1000)
    est2 = clone(est1)
    a = est1.fit_transform(A, B)
    b = est2.fit_transform(sp.csr_matrix(A), sp.csr_matrix(B))
    for F, G in zip(a, b):
        np.testing.assert_array_almost_equal(F, G)


def test_auto_compute_alpha():  # :354-371
    rng = np.random.mtrand.RandomState(36)
    X, Y = rng.randn(10, 10), rng.randn(10, 5)
    kw = dict(n_components=2, solver="newton", x_init='svd', y_init='svd', U_non_negative=False,
              V_non_negative=False, Z_non_negative=False, random_state=0, max_iter=100)
    U1, V1, Z1 = CMF(alpha=0.5, **kw).fit_transform(X, Y)
    U2, V2, Z2 = CMF(alpha="auto", **kw).fit_transform(X, Y)
    assert np.linalg.norm(U2 @ V2.T - X) > np.linalg.norm(U1 @ V1.T - X)
    assert np.linalg.norm(V1 @ Z1.T - Y) > np.linalg.norm(V2 @ Z2.T - Y)


def test_transform_custom_init():  # :67-82
    rs = np.random.RandomState(0)
    X, Y = np.abs(rs.randn(6, 5)), np.abs(rs.randn(5, 1))
    avg = np.sqrt(X.mean() / 4)
    U0, V0 = np.abs(avg * rs.randn(6, 4)), np.abs(avg * rs.randn(5, 4))
    Z0 = np.abs(np.sqrt(Y.mean() / 4) * rs.randn(1, 4))
    CMF(solver='newton', n_components=4, x_init='custom', y_init='custom', random_state=0).fit_transform(X, Y, U=U0, V=V0, Z=Z0)


def test_input_method_compatibility():  # :85-101: every pair of init methods, one MU iteration
    import itertools
    rng = np.random.mtrand.RandomState(0)
    X, Y = np.abs(rng.randn(6, 5)), np.abs(rng.randn(5, 6))
    avg = np.sqrt(X.mean() / 4)
    U0, V0 = np.abs(avg * rng.randn(6, 4)), np.abs(avg * rng.randn(5, 4))
    Z0 = np.abs(np.sqrt(Y.mean() / 4) * rng.randn(6, 4))
    inits = [None, 'random', 'nndsvd', 'nndsvda', 'nndsvdar', 'custom']
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for xi, yi in itertools.product(inits, inits):
            U, V, Z = CMF(n_components=4, solver='mu', x_init=xi, y_init=yi, random_state=0,
                          max_iter=1).fit_transform(X, Y, U=U0.copy(), V=V0.copy(), Z=Z0.copy())
            assert np.isfinite(U).all() and np.isfinite(V).all() and np.isfinite(Z).all()


def test_svd_ncomponents_lt_nfeatures():  # :340-351
    rng = np.random.mtrand.RandomState(42)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        CMF(n_components=3, solver="newton", x_init='svd', y_init='svd', U_non_negative=False, V_non_negative=False,
            Z_non_negative=False, random_state=0, max_iter=1).fit(rng.randn(6, 4), rng.randn(4, 2))


def test_analysis(capsys):  # :411-423 (the reference's own version breaks on current sklearn; ours does not)
    from sklearn.feature_extraction.text import CountVectorizer
    rng = np.random.mtrand.RandomState(36)
    model = CMF(n_components=2, solver="newton", max_iter=1)
    cv = CountVectorizer()
    X_ = sp.csr_matrix(cv.fit_transform(["hello world", "goodbye world", "hello goodbye"]))
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        model.fit_transform(X_.T, np.abs(rng.randn(3, 1)))
    model.print_topic_terms(cv, importances=False)
    model.print_topic_terms(cv, importances=True)
    out = capsys.readouterr().out
    assert "Topic 1" in out and "Topic 2" in out and "hello" in out


@pytest.mark.parametrize("solver", solvers)
def test_transform_after_fit(solver):  # :374-408
    g = load_golden("g4_fit_level")
    X, Y = g["tr_X"], g["tr_Y"]
    m = CMF(n_components=3, solver=solver, x_init="random", y_init="random", random_state=0, max_iter=60)
    U, V, Z = m.fit_transform(X, Y)
    np.testing.assert_allclose(V, g["tr_%s_V" % solver], rtol=5e-3, atol=5e-3)
    Ut, Vt, Zt = m.transform(X, None)
    np.testing.assert_array_equal(Vt, V)           # V untouched when Y is None
    np.testing.assert_allclose(Ut, g["tr_%s_Ut" % solver], rtol=5e-3, atol=5e-3)
    assert clone(m).get_params()["solver"] == solver


@pytest.mark.parametrize("solver", ["mu", "newton"])
def test_transform_after_fit_no_labels(solver):  # :395-408
    rng = np.random.mtrand.RandomState(36)
    X = rng.randn(7, 5)
    Y = rng.randn(5, 3)
    X_new = rng.randn(15, 5)
    m = CMF(n_components=2, solver=solver, x_init="svd", y_init="svd", U_non_negative=False, V_non_negative=False,
            Z_non_negative=False, random_state=0, max_iter=100)
    U_ft, V_ft, Z_ft = m.fit_transform(X, Y)
    U_t, V_t, Z_t = m.transform(X_new, None)
    np.testing.assert_array_equal(V_t, V_ft)
    assert U_t.shape == (15, 2) and np.isfinite(U_t).all()


def test_missing_library_is_loud(monkeypatch):
    from pycmf_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libcmfhip.so")
    with pytest.raises(RuntimeError):
        _lib.load()


def test_device_assisted_randomized_svd_matches_sklearn():
    """above DEVICE_SVD_MIN_CELLS the initialisers' randomized SVD takes its products
    from the GPU copy of the data; same algorithm and test matrix as sklearn -> same factors (fp32)."""
    from sklearn.utils.extmath import randomized_svd
    from pycmf_amd import _lib
    from pycmf_amd.factor_init import DeviceOperand, randomized_svd_device, initialize_mf
    rng = np.random.RandomState(0)
    for shape in ((2600, 1700), (1500, 2900)):     # tall and wide (sklearn transposes the wide one)
        M = np.abs(rng.randn(shape[0], 12) @ rng.randn(12, shape[1])) + 0.05 * np.abs(rng.randn(*shape))
        Y = np.abs(rng.randn(shape[1], 5))
        ctx = _lib.Context(0)
        ctx.set_problem(shape[0], shape[1], 5, 8)
        ctx.set_data(0, M); ctx.set_data(1, Y)
        op = DeviceOperand(ctx, 0, M.shape)
        np.testing.assert_allclose(op.dot(Y), M @ Y, rtol=2e-5)
        np.testing.assert_allclose(op.tdot(M[:, :3]), M.T @ M[:, :3], rtol=2e-5)
        U1, s1, V1 = randomized_svd_device(op, 8, random_state=3)
        U0, s0, V0 = randomized_svd(M, 8, random_state=3)
        np.testing.assert_allclose(s1, s0, rtol=1e-4)
        np.testing.assert_allclose(U1 * s1 @ V1, U0 * s0 @ V0, rtol=0, atol=2e-3 * np.abs(M).max())
        # component signs (svd_flip): sklearn's decision reads M's U in both orientations -- the leading, well
        # separated components must come out with sklearn's sign, not just the product
        for comp in range(3):
            assert np.dot(U1[:, comp], U0[:, comp]) > 0.99 and np.dot(V1[comp], V0[comp]) > 0.99, (shape, comp)
        A1, B1 = initialize_mf(M, 8, init="nndsvd", random_state=3, non_negative=True, operand=op)
        A0, B0 = initialize_mf(M, 8, init="nndsvd", random_state=3, non_negative=True)
        np.testing.assert_allclose(A1 @ B1.T, A0 @ B0.T, rtol=0, atol=5e-3 * np.abs(M).max())
        ctx.close()


def test_fit_with_device_assisted_init():
    rng = np.random.RandomState(1)
    U, V, Z = np.abs(rng.randn(2400, 6)), np.abs(rng.randn(1800, 6)), np.abs(rng.randn(40, 6))
    X, Y = U @ V.T, V @ Z.T
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m = CMF(n_components=6, solver="mu", random_state=0, max_iter=200)
        m.fit(X, Y)
    assert m.reconstruction_err_ < 0.05 * (np.linalg.norm(X) + np.linalg.norm(Y))


def test_driver_level_init_and_v_merge_match_reference():
    """g5 drv_*: what the reference's driver hands to the solver -- 'random' init of both sides from ONE seed
    (RandomState re-created per call, pycmf/cmf.py:110-117), V = mean of the two V candidates (:425-430) -- observed
    through one MU iteration from it."""
    from pycmf_amd import collective_matrix_factorization
    g = load_golden("g5_init")
    U, V, Z, n_it = collective_matrix_factorization(g["M"], g["drv_Y"], n_components=4, x_init="random", y_init="random",
                                                    solver="mu", max_iter=1, random_state=3)
    assert n_it == 1
    for a, name in ((U, "drv_U1"), (V, "drv_V1"), (Z, "drv_Z1")):
        np.testing.assert_allclose(a, g[name], rtol=2e-5, atol=1e-7)


@pytest.mark.parametrize("solver", solvers)
def test_fit_from_the_reference_default_init(solver):
    """g4 fc_*_nndsvdar_*: the reference's own default path (nndsvdar init from random_state=0, then the solver loop):
    same stopping iteration, same reconstruction error."""
    g = load_golden("g4_fit_level")
    m = CMF(n_components=5, solver=solver, x_init="nndsvdar", y_init="nndsvdar", random_state=0, max_iter=1000)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        m.fit(g["fc_X"], g["fc_Y"])
    assert m.n_iter_ == int(g["fc_%s_nndsvdar_n_iter" % solver])
    np.testing.assert_allclose(m.reconstruction_err_, float(g["fc_%s_nndsvdar_err" % solver]), rtol=1e-4)
