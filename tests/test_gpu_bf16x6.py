"""Optional arithmetic of the data passes at k_pad = 256 / 128 (cmf_set_option "gemm_arith" = 1): three bf16 planes per fp32
operand, six cross products on the bf16 matrix pipe -- must be indistinguishable from the fp32 MFMA path at fp32
tolerance, and match the fp64 oracle like it."""
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


def _run(lib, arith, X, Y, U, V, Z, steps, l1=0.0, l2=0.0):
    ctx = lib.Context(0)
    ctx.set_option("gemm_arith", arith)
    ctx.set_option("gemm_arith_min_tiles", 1)
    m, k = U.shape
    ctx.set_problem(m, V.shape[0], Z.shape[0], k)
    ctx.set_data(0, X); ctx.set_data(1, Y)
    for w, F in enumerate((U, V, Z)):
        ctx.set_factor(w, F)
    for _ in range(steps):
        ctx.mu_step(l1, l2, 7)
    out = [ctx.get_factor(w) for w in range(3)]
    err = ctx.residual_sq()
    ctx.close()
    return out, err


@pytest.mark.parametrize("k,shape", [(256, (700, 530, 300)), (200, (700, 530, 300)), (256, (300, 4300, 260)),
                                     (128, (700, 530, 300)), (100, (300, 4300, 260))])
def test_bf16x6_mu_matches_fp32_path_and_oracle(lib, k, shape):
    from oracle import cmf_oracle as O
    m, d, p = shape                  # ragged; k = 200 pads to 256; d = 4300 makes X V / Y^T V split their reduction into slabs
    rng = np.random.RandomState(9)
    X, Y = np.abs(rng.randn(m, d)), np.abs(rng.randn(d, p))
    s = np.sqrt(X.mean() / k)
    U, V, Z = s * np.abs(rng.randn(m, k)), s * np.abs(rng.randn(d, k)), s * np.abs(rng.randn(p, k))
    got6, e6 = _run(lib, 1, X, Y, U, V, Z, 3, 0.01, 0.02)
    got32, e32 = _run(lib, 0, X, Y, U, V, Z, 3, 0.01, 0.02)
    Ur, Vr, Zr = U.copy(), V.copy(), Z.copy()
    for _ in range(3):
        O.mu_update_step(X, Y, Ur, Vr, Zr, 0.01, 0.02)
    for a, b, ref in zip(got6, got32, (Ur, Vr, Zr)):
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-7 * np.abs(ref).max())   # the two arithmetics agree at fp32 level
        np.testing.assert_allclose(a, ref, rtol=2e-4, atol=1e-6 * np.abs(ref).max())  # and with the fp64 oracle
    np.testing.assert_allclose(e6, e32, rtol=1e-5)


def test_bf16x6_signed_data_and_wide_dynamic_range(lib):
    """The split must be exact for any finite fp32 value: signed entries, magnitudes from 1e-6 to 1e+4."""
    from oracle import cmf_oracle as O
    m, d, p, k = 520, 300, 260, 256
    rng = np.random.RandomState(10)
    X = rng.randn(m, d) * 10.0 ** rng.uniform(-6, 4, size=(m, d))
    Y = rng.randn(d, p) * 10.0 ** rng.uniform(-6, 4, size=(d, p))
    U, V, Z = rng.randn(m, k), rng.randn(d, k), rng.randn(p, k)
    got6, _ = _run(lib, 1, X, Y, U, V, Z, 1)
    got32, _ = _run(lib, 0, X, Y, U, V, Z, 1)
    Ur, Vr, Zr = U.copy(), V.copy(), Z.copy()
    O.mu_update_step(X.astype(np.float32).astype(np.float64), Y.astype(np.float32).astype(np.float64), Ur, Vr, Zr)
    for a, b, ref in zip(got6, got32, (Ur, Vr, Zr)):
        # signed sums cancel: compare against the oracle with the error scale of the fp32 path itself
        scale = np.abs(ref).max()
        assert np.abs(a - ref).max() <= 4.0 * max(np.abs(b - ref).max(), 1e-6 * scale)


@pytest.mark.parametrize("k", [256, 200])
def test_bf16x6_row_kernel_matches_fp32_row_kernel(lib, k):
    """Per-row Newton sweeps at k_pad = 256 with gemm_arith = 1: the Hessians accumulate on the bf16 matrix pipe from
    three-plane splits of sqrt(w_j) o_j fetched with transposing LDS reads (row_hess6_kernel).  Same device-drawn samples
    -> the factors of the fp32 row kernel; ragged sample counts and both accumulate modes (the V sweep) included."""
    m, d, p = 500, 601, 450
    rng = np.random.RandomState(12)
    X, Y = rng.rand(m, d), rng.rand(d, p)
    U0, V0, Z0 = 0.2 * rng.randn(m, k), 0.2 * rng.randn(d, k), 0.2 * rng.randn(p, k)
    out = []
    for arith in (0, 1):
        ctx = lib.Context(0)
        ctx.set_option("gemm_arith", arith)
        ctx.set_problem(m, d, p, k)
        ctx.set_data(0, X); ctx.set_data(1, Y)
        for w, F in enumerate((U0, V0, Z0)):
            ctx.set_factor(w, F)
        ctx.newton_step_device_sampled(0.4, 0.01, 0.05, "logit", "logit", 0, 7, 0.2, 0.63, 99)
        out.append([ctx.get_factor(w) for w in range(3)])
        ctx.close()
    for a, b in zip(*out):
        np.testing.assert_allclose(b, a, rtol=1e-3, atol=1e-4 * np.abs(a).max())


def test_bf16x6_newton_matches_oracle(lib):
    """The same arithmetic against the fp64 oracle with host-drawn samples (linear x, logit y, ratio 0.5, k = 256)."""
    from oracle import cmf_oracle as O
    m, d, p, k = 270, 90, 40, 256
    rng = np.random.RandomState(k)
    X, Y = np.abs(rng.randn(m, d)), rng.rand(d, p)
    U0, V0, Z0 = 0.3 * rng.randn(m, k), 0.3 * rng.randn(d, k), 0.3 * rng.randn(p, k)
    np.random.seed(4)
    masks = {"U": [], "Z": [], "V": []}
    Ur, Vr, Zr = U0.copy(), V0.copy(), Z0.copy()
    O.newton_update_step(X, Y, Ur, Vr, Zr, 0.4, 0.01, 0.05, "linear", "logit", False, False, False, ratio=0.5, pert=0.2, masks=masks)
    ctx = lib.Context(0)
    ctx.set_option("gemm_arith", 1)
    ctx.set_option("gemm_arith_min_tiles", 1)
    ctx.set_problem(m, d, p, k)
    ctx.set_data(0, X); ctx.set_data(1, Y)
    for w, F in enumerate((U0, V0, Z0)):
        ctx.set_factor(w, F)
    ctx.newton_step(0.4, 0.01, 0.05, "linear", "logit", 0, 7, 0.2, 0.5, np.array(masks["U"]), np.array(masks["Z"]),
                    np.array([a for a, _ in masks["V"]]), np.array([b for _, b in masks["V"]]))
    got = [ctx.get_factor(w) for w in range(3)]
    ctx.close()
    for a, b in zip(got, (Ur, Vr, Zr)):
        np.testing.assert_allclose(a, b, rtol=2e-3, atol=2e-3 * np.abs(b).max())
