"""The rank-one route of `_safe_invert`'s clamp (csrc/cmf_rank1clamp.hip.h): per-row Hessians with ONE eigenvalue above the threshold
and every other one below it -- the steady state of BASELINE configs[2] at the reference's default l2 = 0 -- get
step = g . Q diag(1 / max(|lambda|, pert)) Q^T (pycmf/cmf_solvers.py:346-356 as :321-326 applies it) from the top eigenpair and a
Cholesky certificate for the rest; anything else keeps its flag for the tridiagonal eigen-solve.  Against the oracle's float64
`safe_invert` through the sweeps' own dispatch (`cmf_safe_solve_batch`, method 0); `cmf_newton_clamp_routes` says which route ran."""
import numpy as np
import pytest

from oracle import cmf_oracle as O

pytestmark = pytest.mark.gpu


def _spectrum_matrix(rng, ev):
    n = len(ev)
    Q, _ = np.linalg.qr(rng.randn(n, n))
    return (Q * ev) @ Q.T


def _solve(H, g, pert, **options):
    from pycmf_amd import _lib
    ctx = _lib.Context(0)
    try:
        for name, val in options.items():
            ctx.set_option(name, val)
        out = ctx.safe_solve_batch(H, g, pert, method=0)
        return out, ctx.newton_clamp_routes()
    finally:
        ctx.close()


def _reference(H, g, pert):
    H32 = H.astype(np.float32).astype(np.float64)
    return np.stack([g[b] @ O.safe_invert(H32[b], pert) for b in range(H.shape[0])])


@pytest.mark.parametrize("k", [256, 200])
def test_one_outlier_rows_take_the_rank_one_route(k):
    """One eigenvalue of 5e2 .. 2e3, the rest in [0, 0.19] (pert = 0.2): every row is served by the rank-one route, to float32
    round-off of the float64 result; the same rows with the route switched off go through the eigen-solve and agree."""
    rng = np.random.RandomState(3)
    Hs = [_spectrum_matrix(rng, np.concatenate([np.sort(rng.uniform(0.0, 0.19, k - 1)), [top]])) for top in (2087.0, 512.0, 900.0, 35.0, 1500.0)]
    Hs.append(_spectrum_matrix(rng, np.concatenate([np.zeros(k - 1), [700.0]])))          # exactly rank one
    H = np.stack(Hs)
    g = rng.randn(len(Hs), k)
    ref = _reference(H, g, 0.2)
    got, (eig_rows, r1_rows) = _solve(H, g, 0.2)
    assert r1_rows == len(Hs) and eig_rows == 0
    err = np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)
    assert err.max() < 5e-6, err
    again, _ = _solve(H, g, 0.2)
    np.testing.assert_array_equal(got, again)
    off, (eig_rows, r1_rows) = _solve(H, g, 0.2, rank1_clamp=0)
    assert r1_rows == 0 and eig_rows == len(Hs)
    assert (np.abs(off - ref).max(axis=1) / np.abs(ref).max(axis=1)).max() < 5e-4   # (the eigen-solve's bound at ||H|| / pert = 1e4: test_gpu_eigclamp.py)
    assert (np.abs(off - got).max(axis=1) / np.abs(ref).max(axis=1)).max() < 5e-4


def test_rows_outside_the_regime_keep_their_flag():
    """Two large eigenvalues (the power iteration does not converge, or the certificate fails), a bulk that crosses the threshold,
    an eigenvalue closer to the threshold than float32 resolves, Gram matrices with many eigenvalues above it: none may be served
    by the rank-one route, all must still come out right (eigen-solve); rows of the regime in the same batch are served by it."""
    rng = np.random.RandomState(4)
    k = 256
    bulk = lambda hi: np.sort(rng.uniform(0.0, hi, k - 2))
    outside = [
        _spectrum_matrix(rng, np.concatenate([bulk(0.19), [800.0, 900.0]])),      # two outliers of the same size
        _spectrum_matrix(rng, np.concatenate([bulk(0.19), [0.5, 900.0]])),        # a second eigenvalue just above the threshold
        _spectrum_matrix(rng, np.concatenate([bulk(0.27), [0.1, 2.27]])),         # the C3X regime: the bulk straddles the threshold
        _spectrum_matrix(rng, np.concatenate([bulk(0.19), [0.19995, 2087.0]])),   # within 4 eps32 lambda_1 = 5e-4 of the threshold
    ]
    B = rng.randn(300, k) * 0.05
    outside.append(B.T @ B)
    inside = [_spectrum_matrix(rng, np.concatenate([np.sort(rng.uniform(0.0, 0.19, k - 1)), [top]])) for top in (640.0, 1234.0, 77.0)]
    H = np.stack(outside + inside)
    g = rng.randn(H.shape[0], k)
    ref = _reference(H, g, 0.2)
    got, (eig_rows, r1_rows) = _solve(H, g, 0.2)
    assert r1_rows == len(inside) and eig_rows == len(outside), (eig_rows, r1_rows)
    err = np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)
    assert err[len(outside):].max() < 5e-6, err
    assert err[:len(outside)].max() < 5e-4, err     # row 3: an eigenvalue 5e-5 below the threshold, resolved to eps32 ||H|| = 2.5e-4 (continuous clamp)


def test_whole_matrix_below_the_threshold_and_unclamped_rows():
    """lambda_1 < pert: everything is clamped, the inverse is I / pert (the route's formula with max(lambda_1, pert)); rows with
    lambda_min >= pert are solved by plain Cholesky and reach no clamp route at all."""
    rng = np.random.RandomState(5)
    k = 256
    Hs = [_spectrum_matrix(rng, np.sort(rng.uniform(0.0, 0.15, k))), 0.05 * np.eye(k) + 1e-3 * _spectrum_matrix(rng, rng.uniform(0, 1, k)),
          _spectrum_matrix(rng, rng.uniform(0.5, 3.0, k))]
    H = np.stack(Hs)
    g = rng.randn(len(Hs), k)
    ref = _reference(H, g, 0.2)
    got, (eig_rows, r1_rows) = _solve(H, g, 0.2)
    assert eig_rows + r1_rows == 2
    err = np.abs(got - ref).max(axis=1) / np.abs(ref).max(axis=1)
    assert err.max() < 5e-6, err
