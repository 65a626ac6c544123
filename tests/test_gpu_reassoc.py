"""Re-associated form of the shared-Hessian Newton sweeps (csrc/cmf_newton.hip.h, "pre-conditioned operand"):
F <- clamp(F (I - H Hinv) + T (s O Hinv) - l1 sign(F) Hinv) is the reference's F - grad Hinv
(pycmf/cmf_solvers.py:396-410, :436-450, :321-326) with the float64 inverse applied to the OTHER factor before the
float32 data contraction, so that the rounding of that contraction is not multiplied by cond(H)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from pycmf_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("no GPU visible: the gpu-marked tests need an MI355X")
    return _lib


def _run(lib, X, Y, F0, steps, args, **opts):
    ctx = lib.Context(0)
    for k_, v_ in opts.items():
        ctx.set_option(k_, v_)
    ctx.set_problem(X.shape[0], X.shape[1], Y.shape[1], F0[0].shape[1])
    ctx.set_data(0, X); ctx.set_data(1, Y)
    for w, F in enumerate(F0):
        ctx.set_factor(w, F)
    for _ in range(steps):
        ctx.newton_step(*args)
    out = [ctx.get_factor(w) for w in range(3)]
    ctx.close()
    return out


_REG = [(0.0, 0.4, False), (0.0, 0.002, True), (0.05, 0.01, False), (0.03, 0.3, True)]


# k_pad 32 / 64 / 128 / 256 with every regularisation case; k_pad 512 (the unfused path, 30 s per case) with one of them
@pytest.mark.parametrize("k,l1,l2,nn", [(k,) + r for k in (20, 48, 100, 200) for r in _REG] + [(300,) + _REG[2]])
def test_reassociated_sweeps_match_the_oracle(lib, k, l1, l2, nn):
    """Two full linear Newton iterations against the float64 oracle, element-wise, for every k_pad class (small-tile
    update kernel, fused 256-row epilogue, unfused k_pad > 256), with and without the l1 term, clamped (l2 far under the
    perturbation, non-negative factors: rank-deficient Grams) and unclamped inverses.  The gradient form (option off)
    runs beside it."""
    from oracle import cmf_oracle as O
    rng = np.random.RandomState(k)
    m, d, p = 420, 380, 200
    Ut, Vt, Zt = np.abs(rng.randn(m, 6)), np.abs(rng.randn(d, 6)), np.abs(rng.randn(p, 6))
    X = Ut @ Vt.T + 0.1 * np.abs(rng.randn(m, d))
    Y = Vt @ Zt.T + 0.1 * np.abs(rng.randn(d, p))
    if not nn:
        X, Y = X - X.mean(), Y - Y.mean()
    sc = np.sqrt(np.abs(X).mean() / k)
    draw = (lambda *s: np.abs(rng.randn(*s))) if nn else rng.randn
    F0 = [sc * draw(m, k), sc * draw(d, k), sc * draw(p, k)]
    U, V, Z = [f.copy() for f in F0]
    for _ in range(2):
        O.newton_update_step(X, Y, U, V, Z, 0.45, l1, l2, "linear", "linear", nn, nn, nn, 1.0, 0.2)
    args = (0.45, l1, l2, "linear", "linear", 7 if nn else 0, 7, 0.2, 1.0)
    new = _run(lib, X, Y, F0, 2, args)
    old = _run(lib, X, Y, F0, 2, args, newton_reassoc=0)
    for a, b, o in zip(new, old, (U, V, Z)):
        scale = np.abs(o).max()
        np.testing.assert_allclose(a, o, rtol=0, atol=1e-4 * scale)   # measured <= 4e-5 (l2 = 0.002, non-negative: cond 1e5)
        # the gradient form multiplies the float32 rounding of its data contractions by cond(H): up to 0.4 * scale in the
        # clamped cases here -- it must stay finite and never be the closer one by more than rounding
        assert np.isfinite(b).all()
        assert np.abs(a - o).max() <= np.abs(b - o).max() + 1e-6 * scale


def test_reassociated_sweep_in_three_stages_equals_the_fused_step(lib):
    """cmf_newton_v_gram / _products / _finish called one after the other (what a row-sharded run does around its two
    collectives) is the V sweep of cmf_newton_step."""
    rng = np.random.RandomState(8)
    m, d, p, k = 300, 260, 150, 40
    X, Y = np.abs(rng.randn(m, d)), np.abs(rng.randn(d, p))
    F0 = [0.2 * np.abs(rng.randn(m, k)), 0.2 * np.abs(rng.randn(d, k)), 0.2 * np.abs(rng.randn(p, k))]
    want = _run(lib, X, Y, F0, 1, (0.3, 0.02, 0.05, "linear", "linear", 7, 7, 0.2, 1.0))
    ctx = lib.Context(0)
    ctx.set_problem(m, d, p, k)
    ctx.set_data(0, X); ctx.set_data(1, Y)
    for w, F in enumerate(F0):
        ctx.set_factor(w, F)
    ctx.newton_uz_update(0.3, 0.02, 0.05, 7, 5, 0.2)
    mp, dp, pp, kp = ctx.geometry()
    gbuf = ctx.scratch(kp * kp * 8)
    pbuf = ctx.scratch(dp * kp * 4)
    ctx.newton_v_gram(0.3, gbuf.data_ptr())
    ctx.newton_v_products(0.3, 0.05, 0.2, gbuf.data_ptr(), pbuf.data_ptr())
    ctx.newton_v_finish(pbuf.data_ptr(), 0.02, 7)
    got = [ctx.get_factor(w) for w in range(3)]
    ctx.close()
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)
