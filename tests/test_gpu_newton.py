"""GPU parity tests for the Newton path (HIP through the C ABI) vs golden vectors minted
from the reference and vs the CPU oracle.  Run with -m gpu on an MI355X."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_golden
from test_oracle_golden import NEWTON_CASES

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from pycmf_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("no GPU visible: the gpu-marked tests need an MI355X")
    return _lib


def test_safe_invert_batch(lib):
    from oracle import cmf_oracle as O
    rng = np.random.RandomState(0)
    ctx = lib.Context(0)
    for k in (1, 2, 3, 7, 16, 33, 64, 100, 128, 200, 256):
        n = 5 if k <= 128 else 2
        A = rng.randn(n, k, k + 3)
        H = A @ A.transpose(0, 2, 1) / k
        H[0] *= 0.01          # everything below the perturbation
        if n > 1:
            H[1] -= 0.8 * np.eye(k)  # indefinite: |lambda| branch
        if n > 2:
            H[2][:, -1] = 0; H[2][-1, :] = 0  # singular
        got = ctx.safe_invert_batch(H, 0.2)
        for i in range(n):
            ref = O.safe_invert(H[i], 0.2)
            np.testing.assert_allclose(got[i], ref, rtol=2e-3, atol=2e-4 * np.abs(ref).max())
    ctx.close()


def _solver(name, seed):
    from pycmf_amd.solver_shell import HipNewtonSolver
    xl, yl, nn, ratio, _, l1, l2, signed = NEWTON_CASES[name]
    return HipNewtonSolver(alpha=0.3, l1_reg=l1, l2_reg=l2, x_link=xl, y_link=yl,
                           U_non_negative=nn, V_non_negative=nn, Z_non_negative=nn,
                           hessian_pertubation=0.2, sg_sample_ratio=ratio, random_state=seed)


@pytest.mark.parametrize("name", sorted(NEWTON_CASES))
@pytest.mark.parametrize("fmt", ["dense", "csr"])
def test_newton_golden_steps(lib, name, fmt):
    xl, yl, nn, ratio, seed, l1, l2, signed = NEWTON_CASES[name]
    g = load_golden("g3_newton_steps")
    X = g["Xlog"] if xl == "logit" else g["X"]
    Y = g["Ylog"] if yl == "logit" else g["Y"]
    if fmt == "csr":
        X = sp.csr_matrix(X)
    sfx = "s" if signed else "p"
    U, V, Z = g["U0" + sfx].copy(), g["V0" + sfx].copy(), g["Z0" + sfx].copy()
    s = _solver(name, seed)
    for it in range(1, 4):
        s.update_step(X, Y, U, V, Z, l1, l2, 0.3)
        if it in (1, 3):
            tol = 2e-5 if it == 1 else 5e-5   # measured <= 1e-6 (tests/tools/measure_parity.py); round 2: 5e-4 / 5e-3
            for n, a in (("U", U), ("V", V), ("Z", Z)):
                ref = g["%s_%s_%s%d" % (name, fmt, n, it)]
                np.testing.assert_allclose(a, ref, rtol=tol, atol=tol * max(1.0, np.abs(ref).max()))
    s.release()


@pytest.mark.parametrize("xl,yl,ratio,k", [("linear", "linear", 1.0, 12), ("logit", "linear", 1.0, 9),
                                          ("linear", "logit", 0.5, 20), ("logit", "logit", 0.7, 33),
                                          ("linear", "linear", 1.0, 256), ("logit", "logit", 0.6, 200)])
def test_newton_vs_oracle_midsize(lib, xl, yl, ratio, k):
    """Sizes that span several 256-row tiles and a padded k; identical host-drawn samples."""
    from oracle import cmf_oracle as O
    from pycmf_amd.solver_shell import HipNewtonSolver
    m, d, p = (300, 270, 130) if k < 100 else (270, 90, 40)  # the oracle does one eigh(k x k) per row
    rng = np.random.RandomState(k)
    X = rng.rand(m, d) if xl == "logit" else np.abs(rng.randn(m, d))
    Y = rng.rand(d, p) if yl == "logit" else np.abs(rng.randn(d, p))
    U0, V0, Z0 = 0.3 * rng.randn(m, k), 0.3 * rng.randn(d, k), 0.3 * rng.randn(p, k)
    kw = dict(alpha=0.4, l1_reg=0.01, l2_reg=0.05, x_link=xl, y_link=yl, U_non_negative=False,
              V_non_negative=False, Z_non_negative=False, hessian_pertubation=0.2, sg_sample_ratio=ratio)
    s = HipNewtonSolver(random_state=5, **kw)
    U, V, Z = U0.copy(), V0.copy(), Z0.copy()
    s.update_step(X, Y, U, V, Z, 0.01, 0.05, 0.4)
    s.release()
    o = O.OracleSolver("newton", random_state=5, **kw)
    Ur, Vr, Zr = U0.copy(), V0.copy(), Z0.copy()
    o.update_step(X, Y, Ur, Vr, Zr)
    for a, b in ((U, Ur), (V, Vr), (Z, Zr)):
        np.testing.assert_allclose(a, b, rtol=2e-3, atol=2e-3 * np.abs(b).max())


def test_device_sampler_exact_size_and_uniform(lib):
    """cmf_newton_step_device_sampled draws exactly int(n*ratio) candidates per row; with a logit
    link the masked residual image vanishes exactly on the unsampled entries, which makes the
    masks observable from the outside: after one U-only step with l1=l2=0 a row moves iff ... we
    instead check the end-to-end contract (the reference's own stochastic tests, :292-314)."""
    from pycmf_amd import CMF
    rng = np.random.mtrand.RandomState(42)
    X, Y = rng.randn(6, 5), rng.randn(5, 6)
    m = CMF(n_components=5, solver="newton", x_init='svd', y_init='svd', U_non_negative=False,
            V_non_negative=False, Z_non_negative=False, alpha=0.5, sg_sample_ratio=0.5,
            random_state=0, max_iter=1000, sg_sampler="device")
    assert m.fit(X, Y).reconstruction_err_ < 0.1


def test_device_sampler_matches_host_sampler_statistically(lib):
    """Same problem, host (NumPy stream) vs device sampler: different samples, same quality."""
    from pycmf_amd.solver_shell import HipNewtonSolver
    rng = np.random.RandomState(3)
    m, d, p, k = 400, 300, 150, 8
    X, Y = np.abs(rng.randn(m, d)), rng.rand(d, p)
    U0, V0, Z0 = 0.3 * rng.randn(m, k), 0.3 * rng.randn(d, k), 0.3 * rng.randn(p, k)
    errs = {}
    for sampler in ("numpy", "device"):
        s = HipNewtonSolver(alpha=0.4, l2_reg=0.05, y_link="logit", U_non_negative=False, V_non_negative=False,
                            Z_non_negative=False, sg_sample_ratio=0.5, random_state=1, max_iter=15, tol=0,
                            sg_sampler=sampler)
        U, V, Z = U0.copy(), V0.copy(), Z0.copy()
        s.fit_iterative_update(X, Y, U, V, Z)
        errs[sampler] = s.compute_error(X, Y, U, V, Z)
        s.release()
    assert abs(errs["numpy"] - errs["device"]) < 0.05 * errs["numpy"]


def test_device_sampler_masks(lib):
    """Observe the device-drawn masks directly: U-only step, linear link, l1 = l2 = 0, V = e_1-like
    so that row i's step depends on sum over its sampled columns of (u_i v_j - x_ij) v_j."""
    m, d, p, k = 64, 200, 40, 1
    X = np.zeros((m, d)); Y = np.zeros((d, p))
    U0 = np.zeros((m, k)); V0 = np.ones((d, k)); Z0 = np.ones((p, k))
    X[:, :] = 1.0  # residual = -1 on every sampled entry -> gradient_i = -alpha * (#sampled) ; H_i = alpha * (#sampled)
    ctx = lib.Context(0)
    ctx.set_problem(m, d, p, k)
    ctx.set_data(0, X); ctx.set_data(1, Y)
    for w, F in enumerate((U0, V0, Z0)):
        ctx.set_factor(w, F)
    # with H_i = alpha*s >= pert the Newton step gives u_i = 1 exactly, independent of the sample; use
    # pert huge instead so that H^-1 = 1/pert and the step reveals s: u_i = alpha * s / pert
    ctx.newton_step_device_sampled(0.5, 0.0, 0.0, "linear", "linear", 0, 1, 1e6, 0.37, 1234)
    U = ctx.get_factor(0)[:, 0]
    s = int(d * 0.37)
    np.testing.assert_allclose(U, 0.5 * s / 1e6, rtol=1e-5)
    ctx.close()


def test_newton_large_k_fallback_paths(lib):
    """k_pad = 512: per-row sweeps take the masked-dense GEMM formulation (Khatri-Rao images) and the
    safe inverse runs entirely on the Jacobi kernel with its global-memory workspace."""
    from oracle import cmf_oracle as O
    from pycmf_amd.solver_shell import HipNewtonSolver
    m, d, p, k = 40, 36, 20, 300
    rng = np.random.RandomState(2)
    X, Y = rng.rand(m, d), np.abs(rng.randn(d, p))
    U0, V0, Z0 = 0.1 * rng.randn(m, k), 0.1 * rng.randn(d, k), 0.1 * rng.randn(p, k)
    kw = dict(alpha=0.4, l1_reg=0.0, l2_reg=0.3, x_link="logit", y_link="linear", U_non_negative=False,
              V_non_negative=False, Z_non_negative=False, hessian_pertubation=0.2)
    s = HipNewtonSolver(**kw)
    U, V, Z = U0.copy(), V0.copy(), Z0.copy()
    s.update_step(X, Y, U, V, Z, 0.0, 0.3, 0.4)
    s.release()
    o = O.OracleSolver("newton", **kw)
    Ur, Vr, Zr = U0.copy(), V0.copy(), Z0.copy()
    o.update_step(X, Y, Ur, Vr, Zr)
    for a, b in ((U, Ur), (V, Vr), (Z, Zr)):
        np.testing.assert_allclose(a, b, rtol=5e-3, atol=5e-3 * np.abs(b).max())


@pytest.mark.parametrize("row_kernel", [0, 1])
def test_row_kernel_and_gemm_formulation_agree(lib, row_kernel):
    """The fused gather kernel and the masked-dense GEMM formulation are two evaluations of the same
    sweep: identical host-drawn samples -> same factors."""
    from oracle import cmf_oracle as O
    m, d, p, k = 310, 280, 140, 40
    rng = np.random.RandomState(8)
    X, Y = np.abs(rng.randn(m, d)), rng.rand(d, p)
    U0, V0, Z0 = 0.3 * rng.randn(m, k), 0.3 * rng.randn(d, k), 0.3 * rng.randn(p, k)
    np.random.seed(4)
    masks = {"U": [], "Z": [], "V": []}
    Ur, Vr, Zr = U0.copy(), V0.copy(), Z0.copy()
    O.newton_update_step(X, Y, Ur, Vr, Zr, 0.4, 0.01, 0.05, "linear", "logit", False, False, False,
                         ratio=0.5, pert=0.2, masks=masks)
    ctx = lib.Context(0)
    ctx.set_option("row_kernel", row_kernel)
    ctx.set_problem(m, d, p, k)
    ctx.set_data(0, X); ctx.set_data(1, Y)
    for w, F in enumerate((U0, V0, Z0)):
        ctx.set_factor(w, F)
    ctx.newton_step(0.4, 0.01, 0.05, "linear", "logit", 0, 7, 0.2, 0.5,
                    np.array(masks["U"]), np.array(masks["Z"]),
                    np.array([a for a, _ in masks["V"]]), np.array([b for _, b in masks["V"]]))
    got = [ctx.get_factor(w) for w in range(3)]
    ctx.close()
    for a, b in zip(got, (Ur, Vr, Zr)):
        np.testing.assert_allclose(a, b, rtol=2e-3, atol=2e-3 * np.abs(b).max())


@pytest.mark.parametrize("links", [("linear", "linear"), ("linear", "logit"), ("logit", "linear")])
@pytest.mark.parametrize("shape,ratio", [((37, 53, 29, 12), 0.5), ((131, 97, 66, 40), 0.9), ((70, 300, 50, 256), 0.4)])
def test_shared_partial_sums_match_the_oracle_and_the_row_form(lib, links, shape, ratio):
    """Linear sampled sides: H_i = s sum_{j in S_i} o_j o_j^T (pycmf/cmf_solvers.py:414-428 with the identity link).  Groups
    of R rows share the sums over the samples they have in common (option row_classes = R) and take their gradient from two
    masked GEMMs: same step as the row-by-row form (row_classes = 0) and as the oracle, for NumPy-ordered (unsorted) lists,
    row counts that are not multiples of R, and ratios where nearly every candidate is in every list."""
    from oracle import cmf_oracle as O
    m, d, p, k = shape
    xl, yl = links
    rng = np.random.RandomState(m + k)
    X = rng.rand(m, d) if xl == "logit" else np.abs(rng.randn(m, d))
    Y = rng.rand(d, p) if yl == "logit" else np.abs(rng.randn(d, p))
    sc = (0.5 / k) ** 0.5
    U0, V0, Z0 = sc * rng.randn(m, k), sc * rng.randn(d, k), sc * rng.randn(p, k)
    np.random.seed(4)
    masks = {"U": [], "Z": [], "V": []}
    Ur, Vr, Zr = U0.copy(), V0.copy(), Z0.copy()
    # l2 above the perturbation: every H_i is positive definite beyond the clamp threshold, so the comparison is about the
    # sums, not about how an eigenvalue clamp amplifies their rounding
    O.newton_update_step(X, Y, Ur, Vr, Zr, 0.4, 0.01, 0.3, xl, yl, False, False, False, ratio=ratio, pert=0.2, masks=masks)
    lists = (np.array(masks["U"]), np.array(masks["Z"]), np.array([a for a, _ in masks["V"]]), np.array([b for _, b in masks["V"]]))
    got = {}
    for R in (0, 2, 3, 4, 5, 6, -1):
        ctx = lib.Context(0)
        ctx.set_option("row_classes", R)
        ctx.set_problem(m, d, p, k)
        ctx.set_data(0, X); ctx.set_data(1, Y)
        for w, F in enumerate((U0, V0, Z0)):
            ctx.set_factor(w, F)
        ctx.newton_step(0.4, 0.01, 0.3, xl, yl, 0, 7, 0.2, ratio, *lists)
        got[R] = [ctx.get_factor(w) for w in range(3)]
        ctx.close()
    for R in (2, 3, 4, 5, 6, -1):
        for a, b, o in zip(got[R], got[0], (Ur, Vr, Zr)):
            np.testing.assert_allclose(a, b, rtol=0, atol=2e-4 * np.abs(b).max())   # float32 sums in another order, then solved
            np.testing.assert_allclose(a, o, rtol=0, atol=2e-3 * np.abs(o).max())


def test_shared_partial_sums_with_many_candidates(lib):
    """100 000 candidates per list: the class-list kernel keeps one LDS byte per candidate plus its histograms, so the
    automatic group size backs off until that fits (and beyond 131 072 candidates the side runs row by row); same U sweep
    as the row-by-row form either way."""
    m, d, p, k, ratio = 10, 100000, 6, 8, 0.5
    rng = np.random.RandomState(5)
    X, Y = np.abs(rng.randn(m, d)), np.abs(rng.randn(d, p))
    U0, V0, Z0 = 0.1 * rng.randn(m, k), 0.1 * rng.randn(d, k), 0.1 * rng.randn(p, k)
    got = {}
    for R in (-1, 6, 0):
        ctx = lib.Context(0)
        ctx.set_option("row_classes", R)
        ctx.set_problem(m, d, p, k)
        ctx.set_data(0, X); ctx.set_data(1, Y)
        for w, F in enumerate((U0, V0, Z0)):
            ctx.set_factor(w, F)
        ctx.newton_step_device_sampled(0.5, 0.0, 0.3, "linear", "linear", 0, 1, 0.2, ratio, 3)   # U sweep: lists over d
        got[R] = ctx.get_factor(0)
        ctx.close()
    for R in (-1, 6):
        np.testing.assert_allclose(got[R], got[0], rtol=0, atol=2e-4 * np.abs(got[0]).max())
    assert np.abs(got[0] - U0).max() > 0


@pytest.mark.parametrize("l2,rank", [(0.05, 0), (0.01, 150)])
def test_group_certificates_do_not_change_the_step(lib, l2, rank):
    """Rows of half a class group share the threshold test of _safe_invert (pycmf/cmf_solvers.py:346-356) when the part of
    their Hessians they have in common already exceeds the perturbation (a positive semi-definite part bounds lambda_min
    from below).  The step must be the one every row's own test gives: bit-identical here, both when the certificates hold
    (well-conditioned factors) and when they fail and every row falls back to its own test (rank-deficient V, l2 under
    the perturbation: the rows take the clamp route)."""
    from oracle import cmf_oracle as O
    m, d, p, k, ratio = 64, 6000, 40, 200, 0.5
    rng = np.random.RandomState(21)
    X, Y = np.abs(rng.randn(m, d)), rng.rand(d, p)
    sc = (0.8 / k) ** 0.5
    U0, V0, Z0 = sc * rng.randn(m, k), sc * rng.randn(d, k), sc * rng.randn(p, k)
    if rank:
        V0[:, rank:] = 0.0     # V^T D V has rank <= 150 < k for every sample set
    got = {}
    for certs in (1, 0):
        ctx = lib.Context(0)
        ctx.set_option("row_certificates", certs)
        ctx.set_problem(m, d, p, k)
        ctx.set_data(0, X); ctx.set_data(1, Y)
        for w, F in enumerate((U0, V0, Z0)):
            ctx.set_factor(w, F)
        ctx.newton_step_device_sampled(0.5, 0.0, l2, "linear", "logit", 0, 1, 0.2, ratio, 7)   # the U sweep
        got[certs] = ctx.get_factor(0)
        ctx.close()
    assert np.isfinite(got[1]).all()
    np.testing.assert_array_equal(got[1], got[0])
    assert np.abs(got[1] - U0).max() > 0


@pytest.mark.parametrize("k", [24, 200])
def test_global_certificate_when_l2_reaches_the_perturbation(lib, k):
    """l2 >= hessian_pertubation with non-negative weights: every per-row Hessian is (positive semi-definite) + l2 I, so
    _safe_invert (pycmf/cmf_solvers.py:346-356) never clamps and no row needs its threshold test -- the reference's own sparse
    Newton settings (l2_reg = 5).  Same step with the certificate (default) and with every row testing itself, bit for bit,
    on the Z sweep (logit, row by row) and the V sweep (shared linear X side + per-row logit Y side); and the oracle's step."""
    from oracle import cmf_oracle as O
    m, d, p = 90, 700, 30
    rng = np.random.RandomState(k)
    X, Y = np.abs(rng.randn(m, d)), (rng.rand(d, p) < 0.2).astype(np.float64)
    sc = (0.8 / k) ** 0.5
    U0, V0, Z0 = sc * np.abs(rng.randn(m, k)), sc * np.abs(rng.randn(d, k)), sc * rng.randn(p, k)
    got = {}
    for certs in (1, 0):
        ctx = lib.Context(0)
        ctx.set_option("row_certificates", certs)
        ctx.set_option("lowrank_rows", 0)      # the general per-row path on both sides of the comparison (p < k here)
        ctx.set_problem(m, d, p, k)
        ctx.set_data(0, X); ctx.set_data(1, Y)
        for w, F in enumerate((U0, V0, Z0)):
            ctx.set_factor(w, F)
        ctx.newton_step(0.5, 0.05, 0.6, "linear", "logit", 3, 7, 0.2, 1.0)
        got[certs] = [ctx.get_factor(w) for w in range(3)]
        ctx.close()
    U, V, Z = U0.copy(), V0.copy(), Z0.copy()
    O.newton_update_step(X, Y, U, V, Z, 0.5, 0.05, 0.6, "linear", "logit", True, True, False, 1.0, 0.2)
    for a, b, o in zip(got[1], got[0], (U, V, Z)):
        np.testing.assert_array_equal(a, b)
        np.testing.assert_allclose(a, o, rtol=0, atol=5e-4 * np.abs(o).max())


@pytest.mark.parametrize("name", ["lin_log_nn", "log_log_free", "lin_log_free_sg"])
def test_cython_variant_matches_compiled_reference(lib, name):
    """HipNewtonSolver(cython_variant=True) reproduces the reference's Cython twin (g7 fixture)."""
    from pycmf_amd.solver_shell import HipNewtonSolver
    xl, yl, nn, ratio, seed, l1, l2, signed = NEWTON_CASES[name]
    g, c = load_golden("g3_newton_steps"), load_golden("g7_cython_steps")
    X = g["Xlog"] if xl == "logit" else g["X"]
    Y = g["Ylog"] if yl == "logit" else g["Y"]
    sfx = "s" if signed else "p"
    U, V, Z = g["U0" + sfx].copy(), g["V0" + sfx].copy(), g["Z0" + sfx].copy()
    s = HipNewtonSolver(alpha=0.3, l1_reg=l1, l2_reg=l2, x_link=xl, y_link=yl, U_non_negative=nn, V_non_negative=nn,
                        Z_non_negative=nn, hessian_pertubation=0.2, sg_sample_ratio=ratio, random_state=seed,
                        cython_variant=True)
    s.update_step(X, Y, U, V, Z, l1, l2, 0.3)
    s.release()
    for n, a in (("U", U), ("V", V), ("Z", Z)):
        ref = c["%s_%s1" % (name, n)]
        np.testing.assert_allclose(a, ref, rtol=5e-4, atol=5e-4 * max(1.0, np.abs(ref).max()))


@pytest.mark.parametrize("fmt", ["dense", "csr"])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_newton_protocol_matches_oracle(lib, world, fmt):
    """SURVEY 8(e), Newton with linear links: `world` shards as separate contexts on this GPU, the all-reduce
    emulated by summing their partial buffers; every shard must land on the oracle's unsharded step
    (cmf_solvers.py:510-522), with dense and with native-CSR row blocks of X."""
    import torch
    from oracle import cmf_oracle as O
    from pycmf_amd.sharded import ShardedNewtonLinear, HipNewtonShardBackend, shard_bounds
    m, d, p, k = 610, 280, 330, 20
    rng = np.random.RandomState(31)
    X = np.abs(rng.randn(m, d)) * (rng.rand(m, d) < 0.15)
    Y = np.abs(rng.randn(d, p))
    U0, V0, Z0 = 0.3 * rng.randn(m, k), 0.3 * rng.randn(d, k), 0.3 * rng.randn(p, k)
    alpha, l1, l2, pert, nn = 0.35, 0.01, 0.05, 0.2, 0b101
    o = O.OracleSolver("newton", alpha=alpha, l1_reg=l1, l2_reg=l2, x_link="linear", y_link="linear",
                       U_non_negative=True, V_non_negative=False, Z_non_negative=True,
                       hessian_pertubation=pert, sg_sample_ratio=1.0)
    Ur, Vr, Zr = U0.copy(), V0.copy(), Z0.copy()
    for _ in range(2):
        o.update_step(X, Y, Ur, Vr, Zr)
    shards, bounds = [], []
    for r in range(world):
        r0, r1 = shard_bounds(m, world, r)
        c0, c1 = shard_bounds(p, world, r)
        ctx = lib.Context(0)
        ctx.set_problem(r1 - r0, d, c1 - c0, k)
        ctx.set_data(0, sp.csr_matrix(X[r0:r1]) if fmt == "csr" else X[r0:r1])
        ctx.set_data(1, Y[:, c0:c1])
        ctx.set_factor(0, U0[r0:r1]); ctx.set_factor(1, V0); ctx.set_factor(2, Z0[c0:c1])
        backend = HipNewtonShardBackend(ctx, alpha, nn, pert)
        buf = torch.zeros(backend.buf_elems(), dtype=torch.float32, device="cuda:0")
        shards.append((ctx, backend, buf)); bounds.append((r0, r1, c0, c1))
    torch.cuda.synchronize()   # the fills run on PyTorch's stream, the contexts launch on their own

    def all_reduce_emulated(_):
        for ctx, _, _ in shards:
            ctx.sync()
        total = torch.stack([b for _, _, b in shards]).sum(0)
        for _, _, b in shards:
            b.copy_(total)
        torch.cuda.synchronize()

    for _ in range(2):
        # the three phases of ShardedNewtonLinear.step, run phase by phase over the shards
        for ctx, backend, buf in shards:
            backend.update_uz(l1, l2, 7)
            backend.partials(buf)
        all_reduce_emulated(None)
        for ctx, backend, buf in shards:
            backend.apply_v(buf, l1, l2)
            ctx.sync()
    for (ctx, _, _), (r0, r1, c0, c1) in zip(shards, bounds):
        for got, ref in ((ctx.get_factor(1), Vr), (ctx.get_factor(0), Ur[r0:r1]), (ctx.get_factor(2), Zr[c0:c1])):
            np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())
        ctx.close()
    # and the driver object itself on a single shard equals the fused step
    ctx = lib.Context(0)
    ctx.set_problem(m, d, p, k)
    ctx.set_data(0, sp.csr_matrix(X) if fmt == "csr" else X); ctx.set_data(1, Y)
    ctx.set_factor(0, U0); ctx.set_factor(1, V0); ctx.set_factor(2, Z0)
    backend = HipNewtonShardBackend(ctx, alpha, nn, pert)
    drv = ShardedNewtonLinear(backend, torch.zeros(backend.buf_elems(), dtype=torch.float32, device="cuda:0"))
    torch.cuda.synchronize()
    for _ in range(2):
        drv.step(l1, l2, 7)
    ctx.sync()
    for got, ref in ((ctx.get_factor(0), Ur), (ctx.get_factor(1), Vr), (ctx.get_factor(2), Zr)):
        np.testing.assert_allclose(got, ref, rtol=2e-3, atol=2e-3 * np.abs(ref).max())
    ctx.close()


def test_symmetric_block_row_kernel_matches_full(lib):
    """k_pad = 256: the row kernel that accumulates only the blocks on or above the diagonal of every H_i and
    mirrors them must give the factors of the full-block kernel (same device-drawn samples), for both accumulate
    modes (the V sweep adds its Y-side Hessian onto the X-side one) and ragged sample counts.  Every row draws
    more samples than k, so the Hessians are well conditioned and the comparison is not about the eigenvalue clamp."""
    m, d, p, k = 500, 601, 450, 256
    rng = np.random.RandomState(12)
    X, Y = rng.rand(m, d), rng.rand(d, p)
    U0, V0, Z0 = 0.2 * rng.randn(m, k), 0.2 * rng.randn(d, k), 0.2 * rng.randn(p, k)
    out = []
    for sym in (0, 1, 3, 4):  # full blocks | upper blocks, two images | one sqrt-weighted image | ... with 16-wide diagonal sub-blocks (default)
        ctx = lib.Context(0)
        ctx.set_option("row_symmetric", sym)
        ctx.set_problem(m, d, p, k)
        ctx.set_data(0, X); ctx.set_data(1, Y)
        for w, F in enumerate((U0, V0, Z0)):
            ctx.set_factor(w, F)
        # one step: a second one starts from factors of magnitude 2 (saturated sigmoids, near-singular Hessians)
        # and amplifies the fp32 summation-order difference between the two kernels
        ctx.newton_step_device_sampled(0.4, 0.01, 0.05, "logit", "logit", 0, 7, 0.2, 0.63, 99)
        out.append([ctx.get_factor(w) for w in range(3)])
        ctx.close()
    for other in out[1:]:
        for a, b in zip(out[0], other):  # fp32 Hessians summed in a different order, then solved
            np.testing.assert_allclose(a, b, rtol=1e-3, atol=1e-4 * np.abs(a).max())


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("xl,yl,ratio,classes", [("linear", "logit", 0.5, 0), ("logit", "linear", 1.0, 4), ("linear", "linear", 0.7, 0),
                                                 ("linear", "logit", 0.5, 4), ("linear", "linear", 0.7, 3), ("linear", "logit", 0.5, 6)])
def test_row_sharded_newton_is_bit_identical(lib, world, xl, yl, ratio, classes):
    """per-row Newton sweeps sharded by rows (two contexts per rank: U/Z sweeps on the rank's rows of X
    and columns of Y, V sweep on its columns of X and rows of Y; factor rows exchanged in between).  `world` ranks
    emulated on this GPU (all contexts on one stream), the in-place all-gather of equal blocks emulated by copying
    every rank's block into every staging tensor: every rank must hold exactly the factors of the unsharded
    iteration (the device sampler keys by global row).  Linear sampled sides in their shared-partial-sum form
    (`row_classes` > 0: groups of rows share outer-product sums, the gradient is a split-K GEMM) add the same terms in an
    order that depends on the shard's extent: those agree to float32 rounding instead."""
    import torch
    from pycmf_amd.sharded import HipNewtonRowsBackend, ShardedNewtonRows, block_bounds
    exact = not (classes and ratio < 1 and "linear" in (xl, yl))
    m, d, p, k = 211, 157, 93, 24
    rng = np.random.RandomState(5)
    X = rng.rand(m, d) if xl == "logit" else np.abs(rng.randn(m, d))
    Y = rng.rand(d, p) if yl == "logit" else np.abs(rng.randn(d, p))
    U0, V0, Z0 = 0.3 * rng.randn(m, k), 0.3 * rng.randn(d, k), 0.3 * rng.randn(p, k)
    alpha, l1, l2, pert, nn = 0.4, 0.01, 0.05, 0.2, 0b010

    def step(ctx, mask, seed):
        if ratio < 1:
            ctx.newton_step_device_sampled(alpha, l1, l2, xl, yl, nn, mask, pert, ratio, seed)
        else:
            ctx.newton_step(alpha, l1, l2, xl, yl, nn, mask, pert, 1.0)

    ref = lib.Context(0)
    ref.set_option("row_classes", classes)
    ref.set_problem(m, d, p, k)
    ref.set_data(0, X); ref.set_data(1, Y)
    for w, F in enumerate((U0, V0, Z0)):
        ref.set_factor(w, F)
    for it in range(2):
        step(ref, 7, 40 + it)
    want = [ref.get_factor(w) for w in range(3)]
    ref.close()

    ranks = []
    stream = torch.cuda.Stream(device="cuda:0")
    sh = stream.cuda_stream
    for r in range(world):
        r0, r1 = block_bounds(m, world, r)
        q0, q1 = block_bounds(d, world, r)
        c0, c1 = block_bounds(p, world, r)
        a = lib.Context(0, sh)
        a.set_option("row_classes", classes)
        a.set_problem(r1 - r0, d, c1 - c0, k)
        a.set_data(0, X[r0:r1]); a.set_data(1, Y[:, c0:c1])
        a.set_factor(0, U0[r0:r1]); a.set_factor(1, V0); a.set_factor(2, Z0[c0:c1])
        b = lib.Context(0, sh)
        b.set_option("row_classes", classes)
        b.set_problem(m, q1 - q0, p, k)
        b.set_data(0, X[:, q0:q1]); b.set_data(1, Y[q0:q1])
        b.set_factor(0, U0); b.set_factor(1, V0[q0:q1]); b.set_factor(2, Z0)
        be = HipNewtonRowsBackend(a, b, (r0, r1, q0, q1, c0, c1), (m, d, p), alpha, xl, yl, nn, pert, ratio)
        staging = [torch.full((world * -(-n // world), be.k_pad), float("nan"), dtype=torch.float32, device="cuda:0")
                   for n in (m, d, p)]
        ranks.append((be, staging))
    torch.cuda.synchronize()

    def gather(which):  # what ShardedNewtonRows._gather does on every rank, block copies standing in for RCCL
        with torch.cuda.stream(stream):
            for be, st in ranks:
                be.export_rows(which, st[which])
            c = ranks[0][1][which].shape[0] // world
            for r, (_, src) in enumerate(ranks):           # all_gather_into_tensor(full, full[r*c:(r+1)*c]) on every rank
                for q, (_, dst) in enumerate(ranks):
                    if q != r:
                        dst[which][r * c:(r + 1) * c].copy_(src[which][r * c:(r + 1) * c])
            for be, st in ranks:
                be.import_rows(which, st[which])

    for it in range(2):
        for be, _ in ranks:
            be.sweep_uz(l1, l2, 7, 40 + it)
        gather(0); gather(2)
        for be, _ in ranks:
            be.sweep_v(l1, l2, 40 + it)
        gather(1)
    for be, _ in ranks:
        r0, r1, q0, q1, c0, c1 = be.bounds
        for got, ref_rows in ((be.ctx_uz.get_factor(0), want[0][r0:r1]), (be.ctx_uz.get_factor(2), want[2][c0:c1]),
                              (be.ctx_uz.get_factor(1), want[1]), (be.ctx_v.get_factor(1), want[1][q0:q1])):
            if exact:
                np.testing.assert_array_equal(got, ref_rows)
            else:
                np.testing.assert_allclose(got, ref_rows, rtol=0, atol=2e-5 * np.abs(ref_rows).max())
        if exact:
            np.testing.assert_array_equal(be.ctx_v.get_factor(0), want[0])
        else:
            np.testing.assert_allclose(be.ctx_v.get_factor(0), want[0], rtol=0, atol=2e-5 * np.abs(want[0]).max())
        be.ctx_uz.close(); be.ctx_v.close()
    # the driver object on a single rank (world 1: gathers degenerate to copies between the two contexts)
    a = lib.Context(0, sh); a.set_option("row_classes", classes); a.set_problem(m, d, p, k); a.set_data(0, X); a.set_data(1, Y)
    b = lib.Context(0, sh); b.set_option("row_classes", classes); b.set_problem(m, d, p, k); b.set_data(0, X); b.set_data(1, Y)
    for c in (a, b):
        for w, F in enumerate((U0, V0, Z0)):
            c.set_factor(w, F)
    be = HipNewtonRowsBackend(a, b, (0, m, 0, d, 0, p), (m, d, p), alpha, xl, yl, nn, pert, ratio)
    drv = ShardedNewtonRows(be, [torch.zeros((n, be.k_pad), dtype=torch.float32, device="cuda:0") for n in (m, d, p)])
    torch.cuda.synchronize()
    for it in range(2):
        drv.step(l1, l2, 7, 40 + it)
    for w in range(3):
        np.testing.assert_array_equal(a.get_factor(w), want[w])
    a.close(); b.close()


@pytest.mark.parametrize("k", [200, 100])
@pytest.mark.parametrize("ns", [1, 0])
def test_flagged_hessians_newton_schulz_and_jacobi(lib, ns, k):
    """k_pad = 256 / 128 with rank-deficient per-row Hessians (fewer samples than k, logit U sweep without l2): every row
    fails the `lambda_min >= pert` test, so the eigenvalue clamp of _safe_invert (:346-356) really acts.  Both
    treatments of the flagged rows -- the Newton-Schulz spectral clamp (default) and the Jacobi eigen-solver --
    must reproduce the fp64 oracle."""
    from oracle import cmf_oracle as O
    m, d, p = 71, 96 if k == 200 else 60, 41     # odd row counts: the last 256 x 256 image of k_pad = 128 is half empty
    rng = np.random.RandomState(21)
    X, Y = rng.rand(m, d), rng.rand(d, p)
    U0, V0, Z0 = 0.15 * rng.randn(m, k), 0.15 * rng.randn(d, k), 0.15 * rng.randn(p, k)
    alpha, l1, l2, pert = 0.4, 0.0, 0.05, 0.2
    Ur, Vr, Zr = U0.copy(), V0.copy(), Z0.copy()
    O.newton_update_step(X, Y, Ur, Vr, Zr, alpha, l1, l2, "logit", "logit", False, False, False, 1.0, pert)
    ctx = lib.Context(0)
    ctx.set_option("newton_schulz", ns)
    ctx.set_problem(m, d, p, k)
    ctx.set_data(0, X); ctx.set_data(1, Y)
    for w, F in enumerate((U0, V0, Z0)):
        ctx.set_factor(w, F)
    ctx.newton_step(alpha, l1, l2, "logit", "logit", 0, 7, pert, 1.0)
    got = [ctx.get_factor(w) for w in range(3)]
    ctx.close()
    for a, b in zip(got, (Ur, Vr, Zr)):
        np.testing.assert_allclose(a, b, rtol=2e-3, atol=2e-3 * np.abs(b).max())


@pytest.mark.parametrize("k", [256, 120])
@pytest.mark.parametrize("ns", [1, 0])
def test_clamped_shared_hessian(lib, ns, k):
    """Linear links, no sampling, d < k: the ONE shared Hessian alpha V^T V + l2 I of the U sweep is rank deficient
    plus l2 < pert, so its eigenvalue clamp acts.  Spectral-clamp route and chip-wide Jacobi route vs the oracle."""
    from oracle import cmf_oracle as O
    m, d, p = 300, k // 2, 40
    rng = np.random.RandomState(k)
    X, Y = np.abs(rng.randn(m, d)), np.abs(rng.randn(d, p))
    U0, V0, Z0 = 0.1 * rng.randn(m, k), 0.1 * rng.randn(d, k), 0.1 * rng.randn(p, k)
    Ur = U0.copy()
    O.newton_sweep_U(Ur, V0, X, 0.5, 0.0, 0.05, "linear", False, 1.0, 0.2)
    ctx = lib.Context(0)
    ctx.set_option("newton_schulz", ns)
    ctx.set_problem(m, d, p, k)
    ctx.set_data(0, X); ctx.set_data(1, Y)
    for w, F in enumerate((U0, V0, Z0)):
        ctx.set_factor(w, F)
    ctx.newton_step(0.5, 0.0, 0.05, "linear", "linear", 0, 1, 0.2, 1.0)   # U sweep only
    got = ctx.get_factor(0)
    ctx.close()
    np.testing.assert_allclose(got, Ur, rtol=2e-3, atol=2e-3 * np.abs(Ur).max())


def test_two_class_sides_with_different_group_sizes_across_chunks(lib):
    """a V sweep with TWO linear sampled sides whose automatic group sizes differ (m in [4096, 8192): groups
    of 4 rows; p >= 16384: groups of 6) and more V rows than one Hessian chunk holds.  Every chunk must start on a group
    boundary of BOTH sides (multiples of lcm(256, 4, 6) = 768), otherwise rows past the first chunk receive the class images
    of the wrong group.  Checked against the row-by-row form (row_classes = 0) on the same device-drawn lists
    (pycmf/cmf_solvers.py:452-486 with identity links)."""
    m, d, p, k, ratio = 4200, 2000, 16500, 8, 0.5
    rng = np.random.RandomState(12)
    X, Y = np.abs(rng.randn(m, d)).astype(np.float32), np.abs(rng.randn(d, p)).astype(np.float32)
    U0, V0, Z0 = 0.2 * rng.randn(m, k), 0.2 * rng.randn(d, k), 0.2 * rng.randn(p, k)
    got = {}
    for R, chunk in ((0, 0), (-1, 256), (-1, 0)):
        ctx = lib.Context(0)
        ctx.set_option("row_classes", R)
        ctx.set_option("row_chunk", chunk)
        ctx.set_problem(m, d, p, k)
        ctx.set_data(0, X); ctx.set_data(1, Y)
        for w, F in enumerate((U0, V0, Z0)):
            ctx.set_factor(w, F)
        ctx.newton_step_device_sampled(0.5, 0.0, 0.3, "linear", "linear", 0, 2, 0.2, ratio, 7)   # V sweep only
        got[(R, chunk)] = ctx.get_factor(1)
        ctx.close()
    ref = got[(0, 0)]
    assert np.abs(ref - V0).max() > 0
    for key in ((-1, 256), (-1, 0)):
        np.testing.assert_allclose(got[key], ref, rtol=0, atol=2e-4 * np.abs(ref).max())


@pytest.mark.parametrize("p,k", [(40, 100), (64, 200), (20, 256), (6, 80)])
def test_lowrank_per_row_side_woodbury_form(lib, p, k):
    """V sweep with a shared (linear, unsampled) X side and a logit Y side of p < k label columns, l2 >= hessian_pertubation:
    H_i = (alpha U^T U + l2 I) + Z^T C_i Z is a rank-p update of ONE shared matrix, so g H_i^-1 comes from a p x p system per
    row (Woodbury) instead of a k x k factorisation per row.  Same iterates as the general per-row path (option off) and as the
    float64 oracle (pycmf/cmf_solvers.py:432-486); the reference's own sparse Newton shape (samples/toxic_comments.ipynb:
    6 label columns, l2_reg = 5)."""
    from oracle import cmf_oracle as O
    m, d = 300, (500 if k <= 100 else 160)      # the oracle eigen-decomposes a k x k matrix per V row
    rng = np.random.RandomState(p + k)
    X = np.abs(rng.randn(m, d)) * (rng.rand(m, d) < 0.2)
    Y = (rng.rand(d, p) < 0.15).astype(np.float64)
    sc = (0.5 / k) ** 0.5
    U0, V0, Z0 = sc * np.abs(rng.randn(m, k)), sc * np.abs(rng.randn(d, k)), sc * rng.randn(p, k)
    args = (0.4, 0.05, 0.7, "linear", "logit", 3, 7, 0.2, 1.0)
    got = {}
    for low in (1, 2, 0):
        ctx = lib.Context(0)
        ctx.set_option("lowrank_rows", low)
        if p == 40:
            ctx.set_option("row_chunk", 100)       # several batches of p x p systems
        ctx.set_problem(m, d, p, k)
        ctx.set_data(0, X); ctx.set_data(1, Y)
        for w, F in enumerate((U0, V0, Z0)):
            ctx.set_factor(w, F)
        for _ in range(2):
            ctx.newton_step(*args)
        got[low] = [ctx.get_factor(w) for w in range(3)]
        ctx.close()
    U, V, Z = U0.copy(), V0.copy(), Z0.copy()
    for _ in range(2):
        O.newton_update_step(X, Y, U, V, Z, 0.4, 0.05, 0.7, "linear", "logit", True, True, False, 1.0, 0.2)
    for a, a2, b, o in zip(got[1], got[2], got[0], (U, V, Z)):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-4 * np.abs(b).max())
        np.testing.assert_allclose(a2, a, rtol=0, atol=2e-5 * np.abs(a).max())   # the p x p systems in registers / through memory
        np.testing.assert_allclose(a, o, rtol=0, atol=5e-4 * np.abs(o).max())
    assert np.abs(got[1][1] - got[0][1]).max() > 0 or k <= 64      # the two paths really are different code


@pytest.mark.parametrize("k,link", [(24, "logit"), (200, "logit"), (200, "linear"), (100, "logit")])
def test_split_row_launches_for_few_rows_with_long_lists(lib, k, link):
    """A Z sweep with fewer rows than CUs and thousands of samples per row (BASELINE configs[4]: 64 label columns over 1e5 rows
    of V): every row's samples are split over several workgroups that write partial Hessians / gradients, summed in chunk order
    (option row_split).  Same step as one workgroup per row (float32 sums in another order) and as the oracle
    (pycmf/cmf_solvers.py:488-508)."""
    from oracle import cmf_oracle as O
    m, d, p = 40, 9000, 12
    rng = np.random.RandomState(k)
    X = np.abs(rng.randn(m, d)).astype(np.float32)
    Y = (rng.rand(d, p) < 0.3).astype(np.float64) if link == "logit" else np.abs(rng.randn(d, p))
    sc = (0.5 / k) ** 0.5
    U0, V0, Z0 = sc * rng.randn(m, k), sc * rng.randn(d, k), sc * rng.randn(p, k)
    got = {}
    for split in (1, 0):
        ctx = lib.Context(0)
        ctx.set_option("row_split", split)
        ctx.set_problem(m, d, p, k)
        ctx.set_data(0, X); ctx.set_data(1, Y)
        for w, F in enumerate((U0, V0, Z0)):
            ctx.set_factor(w, F)
        if link == "logit":
            ctx.newton_step(0.5, 0.01, 0.3, "linear", "logit", 0, 4, 0.2, 1.0)                 # Z sweep, all d samples per row
        else:
            ctx.newton_step_device_sampled(0.5, 0.01, 0.3, "linear", "linear", 0, 4, 0.2, 0.6, 5)   # sampled: explicit lists
        got[split] = ctx.get_factor(2)
        ctx.close()
    np.testing.assert_allclose(got[1], got[0], rtol=0, atol=2e-4 * np.abs(got[0]).max())
    assert np.abs(got[1] - Z0).max() > 0
    if link == "logit":
        U, V, Z = U0.copy(), V0.copy(), Z0.copy()
        O.newton_update_step(X.astype(np.float64), Y, U, V, Z, 0.5, 0.01, 0.3, "linear", "logit", False, False, False, 1.0, 0.2,
                             update_U=False, update_V=False)
        np.testing.assert_allclose(got[1], Z, rtol=0, atol=1e-3 * np.abs(Z).max())


@pytest.mark.parametrize("k", [129, 200, 256])
def test_blocked_mfma_cholesky_matches_the_rank1_kernel(lib, k):
    """Per-row solves at k_pad = 256: the blocked Cholesky on the matrix pipe (cmf_chol_mfma.hip.h; valid orders that end inside a
    32-column panel, and the full 256) against the rank-1 register kernel it replaces (option chol_mfma = 0) -- the same
    threshold decisions, the same steps to float32 round-off of two different elimination orders -- and both against the
    float64 oracle (pycmf/cmf_solvers.py:346-356, :321-326)."""
    from oracle import cmf_oracle as O
    rng = np.random.RandomState(k)
    m, d, p = 96, 420, 24
    X, Y = np.abs(rng.randn(m, d)), rng.rand(d, p)
    sc = 0.4 / np.sqrt(k / 8.0)
    U, V, Z = sc * rng.randn(m, k), sc * rng.randn(d, k), sc * rng.randn(p, k)
    args = (0.5, 0.0, 0.3, "logit", "logit", 0, 7, 0.2, 1.0)
    outs = []
    for opt in (1, 0):
        ctx = lib.Context(0)
        ctx.set_option("chol_mfma", opt)
        ctx.set_problem(m, d, p, k)
        ctx.set_data(0, X); ctx.set_data(1, Y)
        for w, F in enumerate((U, V, Z)):
            ctx.set_factor(w, F)
        ctx.newton_step(*args)
        outs.append(([ctx.get_factor(w) for w in range(3)], ctx.newton_clamp_stats()))
        ctx.close()
    assert outs[0][1][0] == outs[1][1][0]                 # the same rows went to the spectral clamp
    for a, b in zip(outs[0][0], outs[1][0]):   # (float32 solves of Hessians with cond ~ 1e3 in two elimination orders: cond x eps32)
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-4 * np.abs(b).max())
    Ur, Vr, Zr = U.copy(), V.copy(), Z.copy()
    O.newton_update_step(X, Y, Ur, Vr, Zr, 0.5, 0.0, 0.3, "logit", "logit", False, False, False, 1.0, 0.2)
    for a, ref in zip(outs[0][0], (Ur, Vr, Zr)):
        np.testing.assert_allclose(a, ref, rtol=0, atol=2e-4 * np.abs(ref).max())
