"""The float32 limit of the per-row spectral clamp, made visible: `cmf_newton_clamp_stats` counts the row Hessians whose smallest
eigenvalue was below the perturbation and keeps the largest ||H||_F / pert among them; `HipNewtonSolver` warns when that ratio
leaves the range in which the stated tolerances hold (tests/tools/fuzz_campaign.py).  Reference: `_safe_invert`,
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


def _step(lib, X, Y, U, V, Z, k, args, options=()):
    m, d, p = U.shape[0], V.shape[0], Z.shape[0]
    ctx = lib.Context(0)
    ctx.set_problem(m, d, p, k)
    for n, v in options:
        ctx.set_option(n, v)
    ctx.set_data(0, X); ctx.set_data(1, Y)
    for w, F in enumerate((U, V, Z)):
        ctx.set_factor(w, F)
    ctx.newton_step(*args)
    out = [ctx.get_factor(w) for w in range(3)], ctx.newton_clamp_stats(reset=True)
    assert ctx.newton_clamp_stats() == (0, 0.0, 0)
    ctx.close()
    return out


def test_ill_conditioned_rows_are_redone_in_float64(lib):
    from oracle import cmf_oracle as O
    # (a) l2 >= pert: every Hessian is certified positive definite above the threshold, the clamp never acts
    X, Y, U, V, Z = _problem(0, 60, 50, 40, 8)
    _, stats = _step(lib, X, Y, U, V, Z, 8, (0.5, 0.0, 0.5, "linear", "logit", 0, 7, 0.2, 1.0))
    assert stats == (0, 0.0, 0)
    # (b) more components than samples and no l2: rank-deficient Hessians, every V row is clamped; after the U sweep's 1 / pert
    # steps ||H|| / pert is ~1e5 -- beyond what float32 Hessians resolve (tests/tools/fuzz_campaign.py found this case)
    m, d, p, k = 40, 103, 2, 100
    X, Y, U, V, Z = _problem(1, m, d, p, k)
    args = (0.75, 2.0, 0.0, "linear", "logit", 2, 7, 0.2, 1.0)
    Ur, Vr, Zr = U.copy(), V.copy(), Z.copy()
    O.newton_update_step(X, Y, Ur, Vr, Zr, 0.75, 2.0, 0.0, "linear", "logit", False, True, False, 1.0, 0.2)
    # float32 only: recorded, and visibly off on V (U and Z, well conditioned, agree)
    got, (rows, ratio, refined) = _step(lib, X, Y, U, V, Z, k, args, [("refine_rows", 0)])
    assert rows >= d and ratio > 1e4 and refined == 0
    for w, ref in ((0, Ur), (2, Zr)):
        np.testing.assert_allclose(got[w], ref, rtol=0, atol=1e-4 * np.abs(ref).max())
    err32 = np.abs(got[1] - Vr).max() / np.abs(Vr).max()
    assert 1e-3 < err32 < 0.1
    # default: those rows are redone in float64 -- nothing is left recorded, and V agrees with the float64 oracle
    got, (rows, ratio, refined) = _step(lib, X, Y, U, V, Z, k, args)
    assert refined >= d and ratio <= 1e4
    for w, ref in enumerate((Ur, Vr, Zr)):
        np.testing.assert_allclose(got[w], ref, rtol=0, atol=1e-4 * np.abs(ref).max())
    assert np.abs(got[1] - Vr).max() / np.abs(Vr).max() < 0.02 * err32


@pytest.mark.parametrize("k,pert", [(100, 0.2), (200, 0.2), (200, 2.0)])
def test_batched_refinement_equals_row_by_row(lib, k, pert):
    """The batched float64 refinement (cmf_refine64.hip.h: Hessians of all listed rows on the float64 matrix pipe, batched Cholesky test,
    blocked float64 substitutions, batched Newton-Schulz images for the rows the clamp acts on -- pert 2 makes it act on every row)
    against the one-row-at-a-time form of round 3, every row forced through either: the same steps to float64 round-off of the
    different summation orders, and both at the float64 oracle."""
    from oracle import cmf_oracle as O
    m, d, p = 70, 333, 9
    X, Y, U, V, Z = _problem(3, m, d, p, k)
    args = (0.6, 0.01, 0.05, "linear", "logit", 0, 7, pert, 1.0)
    outs = []
    for batched in (1, 0):
        got, (rows, ratio, refined) = _step(lib, X, Y, U, V, Z, k, args, [("refine_rows_batched", batched), ("refine_rows_ratio", 1), ("refine_rows_cond", 1), ("refine_rows_tol_ppm", 0)])   # (tol 0: the ratios alone decide -- every row)
        assert refined >= d + p                      # every Z and V row (logit / two-sided sweeps) went through the float64 path
        outs.append(got)
    for a, b in zip(*outs):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-6 * np.abs(b).max())
    Ur, Vr, Zr = U.copy(), V.copy(), Z.copy()
    O.newton_update_step(X, Y, Ur, Vr, Zr, 0.6, 0.01, 0.05, "linear", "logit", False, False, False, 1.0, pert)
    for a, ref in zip(outs[0], (Ur, Vr, Zr)):
        np.testing.assert_allclose(a, ref, rtol=0, atol=2e-5 * np.abs(ref).max())


def test_refinement_across_chunks_and_with_sampling(lib):
    """Rows to refine in every chunk of a sweep split into several (option row_chunk), index lists in play (ratio 0.5) and
    a native CSR X: chunk-relative row numbers, absolute list rows and the sparse target term of the float64 path."""
    import scipy.sparse as sp
    from oracle import cmf_oracle as O
    rng = np.random.RandomState(5)
    m, d, p, k = 48, 600, 6, 70
    X = np.abs(rng.randn(m, d)); X[rng.rand(m, d) < 0.8] = 0.0
    Y = (rng.rand(d, p) < 0.3).astype(float)
    sc = 0.4 / np.sqrt(k / 8.0)
    U, V, Z = sc * rng.randn(m, k), np.abs(sc * rng.randn(d, k)), sc * rng.randn(p, k)
    ratio, alpha, l1, l2, pert = 0.5, 0.6, 0.1, 0.0, 0.002
    np.random.seed(11)
    masks = {"U": [], "Z": [], "V": []}
    Ur, Vr, Zr = U.copy(), V.copy(), Z.copy()
    O.newton_update_step(X, Y, Ur, Vr, Zr, alpha, l1, l2, "linear", "logit", False, True, False, ratio, pert, masks=masks)
    su, sm, sp_ = int(d * ratio), int(m * ratio), int(p * ratio)
    lists = [np.array(masks["U"], dtype=np.int32).reshape(m, su), np.array(masks["Z"], dtype=np.int32).reshape(p, su),
             np.array([a for a, _ in masks["V"]], dtype=np.int32).reshape(d, sm),
             np.array([b for _, b in masks["V"]], dtype=np.int32).reshape(d, sp_)]
    outs = []
    for opts in ([("row_chunk", 256)], []):
        ctx = lib.Context(0)
        ctx.set_problem(m, d, p, k)
        for n, v in opts:
            ctx.set_option(n, v)
        ctx.set_data(0, sp.csr_matrix(X)); ctx.set_data(1, Y)
        for w, F in enumerate((U, V, Z)):
            ctx.set_factor(w, F)
        ctx.newton_step(alpha, l1, l2, "linear", "logit", 2, 7, pert, ratio, *lists)
        rows, ratio_left, refined = ctx.newton_clamp_stats()
        assert refined >= d and ratio_left <= 1e4          # k > samples per row, l2 = 0: every V row qualifies
        outs.append([ctx.get_factor(w) for w in range(3)])
        ctx.close()
    for w, ref in enumerate((Ur, Vr, Zr)):
        np.testing.assert_allclose(outs[0][w], ref, rtol=0, atol=1e-3 * np.abs(ref).max())
        np.testing.assert_array_equal(outs[0][w], outs[1][w])  # the chunking changes nothing


@pytest.mark.parametrize("line", range(10))
def test_flagged_campaign_cases_with_refinement(lib, line):
    """The cases tests/tools/fuzz_campaign.py flagged in round 3 (V off by 1e-2 .. 0.4 in float32: clamped rows with ||H|| / pert >= 6e4,
    and -- the last three -- plain Cholesky solves of Hessians with condition numbers >= 3e4), replayed with the default
    float64 refinement (fused row path for k_pad <= 256, masked-dense path above): within 2e-3 of the float64 oracle."""
    import json, os, sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(here, "tools"))
    import fuzz_campaign as FC
    case = json.loads(open(os.path.join(here, "tools", "fuzz_flagged.jsonl")).read().splitlines()[line])["case"]
    case["options"] = {}
    info = {}
    err = FC.run_case(case, case["seed"], info)
    assert info["refined_rows"] > 0 and info["clamp_ratio"] <= 1e4
    assert max(err) < 2e-3, (err, info)


def test_solver_warns_when_the_clamp_leaves_the_float32_range(monkeypatch):
    from pycmf_amd.solver_shell import HipNewtonSolver
    m, d, p, k = 40, 103, 2, 100
    X, Y, U, V, Z = _problem(1, m, d, p, k)
    kw = dict(max_iter=1, tol=0, alpha=0.75, x_link="linear", y_link="logit", U_non_negative=False, V_non_negative=True,
              Z_non_negative=False, hessian_pertubation=0.2)
    # default: the ill-conditioned rows are redone in float64, no warning
    s = HipNewtonSolver(l1_reg=2.0, l2_reg=0.0, **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        s.fit_iterative_update(X, Y, U.copy(), V.copy(), Z.copy())
    assert s.refined_rows_ >= d and s.clamp_ratio_ <= HipNewtonSolver.CLAMP_RATIO_WARN
    # refinement off: recorded and warned
    monkeypatch.setenv("PYCMF_AMD_REFINE_ROWS", "0")
    s = HipNewtonSolver(l1_reg=2.0, l2_reg=0.0, **kw)
    with pytest.warns(RuntimeWarning, match="float32"):
        s.fit_iterative_update(X, Y, U.copy(), V.copy(), Z.copy())
    assert s.clamped_rows_ >= d and s.clamp_ratio_ > HipNewtonSolver.CLAMP_RATIO_WARN and s.refined_rows_ == 0
    monkeypatch.delenv("PYCMF_AMD_REFINE_ROWS")
    # the same problem with l2 >= pert: nothing clamped
    s = HipNewtonSolver(l1_reg=2.0, l2_reg=0.5, **kw)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        s.fit_iterative_update(X, Y, U.copy(), V.copy(), Z.copy())
    assert s.clamped_rows_ == 0 and s.refined_rows_ == 0
