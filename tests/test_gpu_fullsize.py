"""Checks at BASELINE.json's FULL shapes (configs[1]..[4]) through the C ABI on one MI355X.

The fp64 oracle finishes a whole iteration only for C2, so C2 is compared with it directly; C3, C4 and C5
are covered by properties of the update rules that do not depend on the size:

* MU is invariant under the gauge (U, V, Z) -> (a U, V / a, a Z), bit-exactly in fp32 for a power of two
  (pycmf/cmf_solvers.py:230-246: numerators and denominators scale by the same power of two);
* the MU objective does not increase (the reference's own test: tests/test_cmf.py:160);
* a sharded iteration (partial buffers summed across shards) equals the unsharded one;
* the device sampler is a pure function of (seed, sweep, row): same seed -> bit-identical step,
  and the symmetric-block row kernel agrees with the full-block one.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from pycmf_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("no GPU visible: the gpu-marked tests need an MI355X")
    return _lib


def _synthetic(lib, m, d, p, k, r0=0, c0=0, rows=None, cols=None, gauge=1.0):
    """bench.py's synthetic problem (|N(0,1)| data, 'random' init rule), optionally a shard of it."""
    rows = m if rows is None else rows
    cols = p if cols is None else cols
    ctx = lib.Context(0)
    ctx.set_problem(rows, d, cols, k)
    ctx.fill_data_synthetic(0, 42, r0, 0)
    ctx.fill_data_synthetic(1, 43, 0, c0)
    scale = (0.7979 / k) ** 0.5
    ctx.fill_factor_synthetic(lib.CMF_U, 101, r0, scale * gauge)
    ctx.fill_factor_synthetic(lib.CMF_V, 102, 0, scale / gauge)
    ctx.fill_factor_synthetic(lib.CMF_Z, 103, c0, scale * gauge)
    return ctx


def _err(ctx):
    ex, ey = ctx.residual_sq()
    return 0.5 * ex ** 0.5 + 0.5 * ey ** 0.5


def test_c2_full_size_mu_matches_oracle(lib):
    """BASELINE configs[1] (16384 x 8192 / 8192 x 4096, k = 128): FIVE MU iterations against the fp64 oracle in the
    reference's operation order, element-wise after every iteration (no drift: the distance stays at float32 round-off) and on
    the relative residuals at the end (north_star: within 1e-4 rel.)."""
    from oracle import cmf_oracle as O
    m, d, p, k = 16384, 8192, 4096, 128
    ctx = _synthetic(lib, m, d, p, k)
    X, Y = ctx.get_data(0).astype(np.float64), ctx.get_data(1).astype(np.float64)
    U, V, Z = (ctx.get_factor(w) for w in range(3))
    for it in range(5):
        ctx.mu_step(0.0, 0.0, 7)
        O.mu_update_step(X, Y, U, V, Z)
        for w, ref in enumerate((U, V, Z)):
            np.testing.assert_allclose(ctx.get_factor(w), ref, rtol=2e-4, atol=1e-6 * np.abs(ref).max(), err_msg="iteration %d" % (it + 1))
    ex, ey = ctx.residual_sq()
    x2, y2 = ctx.data_sq()
    rx_ref = np.linalg.norm(X - U @ V.T) / np.linalg.norm(X)
    ry_ref = np.linalg.norm(Y - V @ Z.T) / np.linalg.norm(Y)
    assert abs((ex / x2) ** 0.5 - rx_ref) <= 1e-4 * rx_ref
    assert abs((ey / y2) ** 0.5 - ry_ref) <= 1e-4 * ry_ref
    ctx.close()


def test_c4_full_size_mu_properties(lib):
    """BASELINE configs[3] (65536^2, k = 256, the headline): monotone objective, bit-exact gauge invariance and
    shard independence of one MU iteration."""
    import torch
    from pycmf_amd.sharded import shard_bounds
    m = d = p = 65536
    k = 256
    ctx = _synthetic(lib, m, d, p, k)
    errs = [_err(ctx)]
    ctx.mu_step(0.0, 0.0, 7)
    first = [ctx.get_factor(w) for w in range(3)]
    errs.append(_err(ctx))
    for _ in range(2):
        ctx.mu_step(0.0, 0.0, 7)
        errs.append(_err(ctx))
    assert all(np.isfinite(errs)) and all(b <= a * (1 + 1e-6) for a, b in zip(errs, errs[1:])), errs
    assert errs[-1] < 0.999 * errs[0]

    # gauge (2 U, V / 2, 2 Z): every product of the step scales by a power of two, so fp32 rounding is unchanged
    scale = (0.7979 / k) ** 0.5
    ctx.fill_factor_synthetic(lib.CMF_U, 101, 0, scale * 2.0)
    ctx.fill_factor_synthetic(lib.CMF_V, 102, 0, scale / 2.0)
    ctx.fill_factor_synthetic(lib.CMF_Z, 103, 0, scale * 2.0)
    ctx.mu_step(0.0, 0.0, 7)
    for w, g in ((0, 2.0), (1, 0.5), (2, 2.0)):
        np.testing.assert_array_equal(ctx.get_factor(w), g * first[w])
    ctx.close()

    # two shards (rows of X / U, columns of Y / rows of Z), all-reduce emulated by summing the partial buffers
    shards = []
    for r in range(2):
        r0, r1 = shard_bounds(m, 2, r)
        c0, c1 = shard_bounds(p, 2, r)
        sc = _synthetic(lib, m, d, p, k, r0, c0, r1 - r0, c1 - c0)
        buf = torch.zeros(sc.v_buf_elems(), dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()   # the fill runs on PyTorch's stream, the context launches on its own
        sc.mu_v_partials(buf.data_ptr())
        shards.append((sc, buf, r0, r1, c0, c1))
    for sc, *_ in shards:
        sc.sync()
    total = shards[0][1] + shards[1][1]
    torch.cuda.synchronize()
    for sc, buf, r0, r1, c0, c1 in shards:
        buf.copy_(total)
        torch.cuda.synchronize()
        sc.mu_v_apply(buf.data_ptr(), 0.0, 0.0)
        sc.mu_uz_update(0.0, 0.0, 7)
        sc.sync()
        # a different split-K partition of the same sums: fp32 round-off only
        np.testing.assert_allclose(sc.get_factor(1), first[1], rtol=1e-4, atol=0)
        np.testing.assert_allclose(sc.get_factor(0), first[0][r0:r1], rtol=1e-4, atol=0)
        np.testing.assert_allclose(sc.get_factor(2), first[2][c0:c1], rtol=1e-4, atol=0)
        sc.close()


def test_c3_full_size_newton_properties(lib):
    """BASELINE configs[2] (32768 x 16384 / 16384 x 8192, k = 256, y logit, sg_sample_ratio 0.5, device sampler):
    the step is a pure function of the seed, the two row-kernel variants agree, and so do the forms of the linear sampled
    X side -- row by row, shared partial sums over groups of 4 and of 6 rows (the default), with and without the groups'
    threshold certificates."""
    m, d, p, k = 32768, 16384, 8192, 256
    args = (0.5, 0.0, 0.1, "linear", "logit", 0, 7, 0.2, 0.5)
    out = {}
    for name, sym, seed, classes, certs in (("a", 1, 1000, -1, 1), ("b", 1, 1000, -1, 1), ("full", 0, 1000, -1, 1), ("other", 1, 1001, -1, 1),
                                            ("rows", 1, 1000, 0, 1), ("four", 1, 1000, 4, 1), ("nocert", 1, 1000, -1, 0),
                                            ("one", 3, 1000, -1, 1), ("d16", 4, 1000, -1, 1), ("d16b", 4, 1000, -1, 1), ("d16rows", 4, 1000, 0, 1)):
        ctx = _synthetic(lib, m, d, p, k)
        ctx.set_option("row_symmetric", sym)
        ctx.set_option("row_classes", classes)
        ctx.set_option("row_certificates", certs)
        ctx.newton_step_device_sampled(*args, seed)
        out[name] = [ctx.get_factor(w) for w in range(3)]
        assert all(np.isfinite(F).all() for F in out[name])
        ctx.close()
    for w in range(3):
        np.testing.assert_array_equal(out["a"][w], out["b"][w])
        # fp32 Hessians summed in a different order, then solved
        np.testing.assert_allclose(out["full"][w], out["a"][w], rtol=1e-3, atol=1e-4 * np.abs(out["a"][w]).max())
        assert np.abs(out["other"][w] - out["a"][w]).max() > 1e-6 * np.abs(out["a"][w]).max()
        # the same sums in another order (float32), then solved
        np.testing.assert_allclose(out["rows"][w], out["a"][w], rtol=0, atol=1e-4 * np.abs(out["a"][w]).max())
        np.testing.assert_allclose(out["four"][w], out["a"][w], rtol=0, atol=1e-4 * np.abs(out["a"][w]).max())
        # the single-image kernel, and the default: its diagonal blocks as 16-wide sub-blocks (class images and row by row)
        np.testing.assert_array_equal(out["d16"][w], out["d16b"][w])
        for name in ("one", "d16", "d16rows"):
            np.testing.assert_allclose(out[name][w], out["a"][w], rtol=0, atol=1e-4 * np.abs(out["a"][w]).max())
        # a certificate only replaces a test it implies: nothing changes
        np.testing.assert_array_equal(out["nocert"][w], out["a"][w])


def test_c5_full_size_sparse_shard_independence(lib):
    """BASELINE configs[4] (CSR 1e6 x 1e5 with 1e8 non-zeros, Y 1e5 x 64, k = 256, Newton, linear links, native CSR):
    two nnz-balanced row shards + the emulated all-reduce give the unsharded iteration."""
    import scipy.sparse as sp
    import torch
    from pycmf_amd.sharded import HipNewtonShardBackend, shard_bounds
    m, d, p, k, npr = 1000000, 100000, 64, 256, 100
    rng = np.random.default_rng(42)
    indices = rng.integers(0, d, size=m * npr, dtype=np.int32)
    indices.reshape(m, npr).sort(axis=1)  # canonical CSR rows (duplicates allowed: they add up, like scipy's)
    data = np.ones(m * npr)
    scale = (npr / d / k) ** 0.5
    alpha, l1, l2, pert = 0.5, 0.0, 0.1, 0.2

    def make(r0, r1, c0, c1):
        ctx = lib.Context(0)
        ctx.set_option("sparse_mode", 2)
        ctx.set_problem(r1 - r0, d, c1 - c0, k)
        # copies: the upload canonicalises (sum_duplicates) in place
        X = sp.csr_matrix((data[r0 * npr:r1 * npr].copy(), indices[r0 * npr:r1 * npr].copy(),
                           np.arange(0, (r1 - r0) * npr + 1, npr, dtype=np.int64)), shape=(r1 - r0, d))
        ctx.set_data(0, X)
        ctx.fill_data_synthetic(1, 43, 0, c0)
        ctx.fill_factor_synthetic(lib.CMF_U, 101, r0, scale)
        ctx.fill_factor_synthetic(lib.CMF_V, 102, 0, scale)
        ctx.fill_factor_synthetic(lib.CMF_Z, 103, c0, scale)
        return ctx

    full = make(0, m, 0, p)
    e0 = full.residual_sq()
    full.newton_step(alpha, l1, l2, "linear", "linear", 0, 7, pert, 1.0)
    ref = [full.get_factor(w) for w in range(3)]
    e1 = full.residual_sq()
    assert all(np.isfinite(F).all() for F in ref) and e1[0] < e0[0] and e1[1] < e0[1]
    full.close()
    shards = []
    for r in range(2):
        r0, r1 = shard_bounds(m, 2, r)
        c0, c1 = shard_bounds(p, 2, r)
        ctx = make(r0, r1, c0, c1)
        be = HipNewtonShardBackend(ctx, alpha, 0, pert)
        buf = torch.zeros(be.buf_elems(), dtype=torch.float32, device="cuda:0")
        torch.cuda.synchronize()   # the fill runs on PyTorch's stream, the context launches on its own
        be.update_uz(l1, l2, 7)
        be.partials(buf)
        shards.append((ctx, be, buf, r0, r1, c0, c1))
    for ctx, *_ in shards:
        ctx.sync()
    total = shards[0][2] + shards[1][2]
    torch.cuda.synchronize()
    for ctx, be, buf, r0, r1, c0, c1 in shards:
        buf.copy_(total)
        torch.cuda.synchronize()
        be.apply_v(buf, l1, l2)
        ctx.sync()
        for w, sl in ((1, slice(None)), (0, slice(r0, r1)), (2, slice(c0, c1))):
            np.testing.assert_allclose(ctx.get_factor(w), ref[w][sl], rtol=1e-3, atol=1e-5 * np.abs(ref[w]).max())
        ctx.close()


def test_c4_full_size_bf16x6_matches_fp32_path(lib):
    """BASELINE configs[3] with the optional arithmetic (gemm_arith = 1: three bf16 planes per fp32 operand, six cross
    products on the bf16 matrix pipe): one full-size MU iteration gives the factors of the fp32-MFMA path to fp32
    round-off, the objective decreases identically, and the power-of-two gauge stays bit-exact."""
    m = d = p = 65536
    k = 256
    out = {}
    for arith in (0, 1):
        ctx = _synthetic(lib, m, d, p, k)
        ctx.set_option("gemm_arith", arith)
        e0 = _err(ctx)
        ctx.mu_step(0.0, 0.0, 7)
        out[arith] = ([ctx.get_factor(w) for w in range(3)], e0, _err(ctx))
        if arith == 1:
            scale = (0.7979 / k) ** 0.5
            ctx.fill_factor_synthetic(lib.CMF_U, 101, 0, scale * 2.0)
            ctx.fill_factor_synthetic(lib.CMF_V, 102, 0, scale / 2.0)
            ctx.fill_factor_synthetic(lib.CMF_Z, 103, 0, scale * 2.0)
            ctx.mu_step(0.0, 0.0, 7)
            for w, g in ((0, 2.0), (1, 0.5), (2, 2.0)):
                np.testing.assert_array_equal(ctx.get_factor(w), g * out[1][0][w])
        ctx.close()
    for a, b in zip(out[0][0], out[1][0]):  # 65536-term fp32 sums accumulated in a different order
        np.testing.assert_allclose(b, a, rtol=1e-4, atol=0)
    assert out[1][2] < out[1][1] and abs(out[1][2] - out[0][2]) <= 1e-6 * out[0][2]


# ---------------------------------------------------------------------------------------------------------------------
# float64 recomputation of SAMPLED ROWS at the full BASELINE sizes: the inputs of a sweep are read back from the device
# (factors whole, the rows / columns of X and Y that the sampled factor rows touch), those rows are recomputed on the host
# with the oracle's arithmetic, and compared with what the device wrote.  Unlike the property checks above this ties the
# full-size kernels to the reference's formulas directly: an index wrap past 2^31 or a tile never visited shows up here.
def _spread(n, count, rng):
    """first, last and a few random rows"""
    return sorted(set([0, n - 1] + [int(v) for v in rng.randint(0, n, size=count)]))


def test_c4_full_size_mu_sampled_rows_vs_fp64(lib):
    """BASELINE configs[3] (65536^2, k = 256, MU): rows of V, U, Z after one update against pycmf/cmf_solvers.py:230-263
    evaluated in float64 on the device's own inputs."""
    m = d = p = 65536
    k = 256
    eps = float(np.finfo(np.float32).eps)
    rng = np.random.RandomState(0)
    ctx = _synthetic(lib, m, d, p, k)
    U0, V0, Z0 = (ctx.get_factor(w) for w in range(3))
    ctx.mu_step(0.0, 0.0, lib.CMF_UPD_V)
    V1 = ctx.get_factor(1)
    G = U0.T @ U0 + Z0.T @ Z0                                    # :245
    for j in _spread(d, 30, rng):
        xcol = ctx.get_data_block(0, 0, m, j, 1)[:, 0].astype(np.float64)
        yrow = ctx.get_data_block(1, j, 1, 0, p)[0].astype(np.float64)
        num = xcol @ U0 + yrow @ Z0                              # :244
        den = V0[j] @ G
        den[den == 0] = eps                                      # :219
        np.testing.assert_allclose(V1[j], V0[j] * num / den, rtol=2e-4, atol=0, err_msg="V row %d" % j)
    ctx.mu_step(0.0, 0.0, lib.CMF_UPD_U | lib.CMF_UPD_Z)
    U1, Z1 = ctx.get_factor(0), ctx.get_factor(2)
    G2 = V1.T @ V1
    for i in _spread(m, 30, rng):
        xrow = ctx.get_data_block(0, i, 1, 0, d)[0].astype(np.float64)
        den = U0[i] @ G2                                         # (U V^T) V, :233
        den[den == 0] = eps
        np.testing.assert_allclose(U1[i], U0[i] * (xrow @ V1) / den, rtol=2e-4, atol=0, err_msg="U row %d" % i)
    for c in _spread(p, 30, rng):
        ycol = ctx.get_data_block(1, 0, d, c, 1)[:, 0].astype(np.float64)
        den = Z0[c] @ G2                                         # :239
        den[den == 0] = eps
        np.testing.assert_allclose(Z1[c], Z0[c] * (ycol @ V1) / den, rtol=2e-4, atol=0, err_msg="Z row %d" % c)
    ctx.close()


def _tile_sums(F):
    """column sums of every 256-row tile of a rows x k array (rows padded with zeros to a multiple of 256)"""
    rows, k = F.shape
    pad = (-rows) % 256
    if pad:
        F = np.vstack([F, np.zeros((pad, k))])
    return F.reshape(-1, 256, k).sum(axis=1)


@pytest.mark.parametrize("shape", [(65536, 65536, 65536, 256), (16384, 8192, 4096, 128)], ids=["c4", "c2"])
def test_full_size_mu_every_output_tile_checksum(lib, shape):
    """BASELINE configs[3] and [1] at full size: EVERY 256-row output tile of the four data passes of one MU iteration
    (X^T U + Y Z, X V, Y^T V: pycmf/cmf_solvers.py:244, :232, :238) and every element of the three fused epilogues (:212-228).
    The products are recovered in float64 from the factors the device wrote -- V1 = V0 * P / (V0 G)  =>  P = V1 (V0 G) / V0 -- and the
    column sums of each tile of P are compared with (block sums of X, Y)^T times the factor, the block sums coming from plain
    float64 reductions of the device's data (cmf_data_block_sums_f64: nothing shared with the GEMM kernels).  A tile never
    visited, a K-step dropped (1 / 2048 of a tile's sum at C4) or an index wrap would show in that tile's row; the sampled-row test
    below checks single elements of a few rows, this one checks all tiles."""
    m, d, p, k = shape
    ctx = _synthetic(lib, m, d, p, k)
    U0, V0, Z0 = (ctx.get_factor(w) for w in range(3))
    mp, dp, pp, kp = ctx.geometry()
    sx_cols = ctx.data_block_sums(0, 1)[:m]           # [m, d_pad / 256]: sums of X over blocks of 256 columns
    sx_rows = ctx.data_block_sums(0, 0)[:, :d]        # [m_pad / 256, d]: sums of X over blocks of 256 rows
    sy_cols = ctx.data_block_sums(1, 1)[:d]           # [d, p_pad / 256]
    sy_rows = ctx.data_block_sums(1, 0)[:, :p]        # [d_pad / 256, p]
    # the block sums themselves against the device's total (one more independent reduction)
    tx, ty = ctx.data_sum()
    assert abs(sx_cols.sum() - tx) <= 1e-9 * tx and abs(sx_rows.sum() - tx) <= 1e-9 * tx
    assert abs(sy_cols.sum() - ty) <= 1e-9 * ty and abs(sy_rows.sum() - ty) <= 1e-9 * ty

    def check(rec, ref, what):
        got = _tile_sums(rec)
        assert got.shape == ref.shape, what
        err = np.abs(got - ref) / np.abs(ref).max(axis=1, keepdims=True)
        worst = np.unravel_index(np.argmax(err), err.shape)
        assert err.max() <= 2e-5, "%s: tile %d column %d off by %.3g of the tile's largest column sum" % (what, worst[0], worst[1], err.max())

    def recover(F1, F0, den, exact):
        """numerator = F1 den / F0 element by element; the handful of entries the synthetic start left at exactly zero (F1 = F0 = 0
        whatever the numerator) are filled in from `exact(row, column)`, the float64 value of that numerator entry"""
        zr, zc = np.nonzero(F0 == 0)
        assert len(zr) <= 64, "%d zero entries in a synthetic factor" % len(zr)
        with np.errstate(invalid="ignore", divide="ignore"):
            P = F1 * den / F0
        for i, c in zip(zr, zc):
            assert F1[i, c] == 0
            P[i, c] = exact(int(i), int(c))
        return P

    xcol = lambda j: ctx.get_data_block(0, 0, m, j, 1)[:, 0].astype(np.float64)
    xrow = lambda i: ctx.get_data_block(0, i, 1, 0, d)[0].astype(np.float64)
    ycol = lambda c: ctx.get_data_block(1, 0, d, c, 1)[:, 0].astype(np.float64)
    yrow = lambda j: ctx.get_data_block(1, j, 1, 0, p)[0].astype(np.float64)
    for it in range(2):      # two iterations: the second starts from factors the device itself produced
        ctx.mu_step(0.0, 0.0, lib.CMF_UPD_V)
        V1 = ctx.get_factor(1)
        P = recover(V1, V0, V0 @ (U0.T @ U0 + Z0.T @ Z0),                  # X^T U + Y Z as the device formed it (:244-245)
                    lambda j, c: xcol(j) @ U0[:, c] + yrow(j) @ Z0[:, c])
        check(P, sx_cols.T @ U0 + sy_rows @ Z0, "iteration %d, X^T U + Y Z" % (it + 1))
        ctx.mu_step(0.0, 0.0, lib.CMF_UPD_U | lib.CMF_UPD_Z)
        U1, Z1 = ctx.get_factor(0), ctx.get_factor(2)
        G2 = V1.T @ V1
        check(recover(U1, U0, U0 @ G2, lambda i, c: xrow(i) @ V1[:, c]), sx_rows @ V1, "iteration %d, X V" % (it + 1))          # :232-233
        check(recover(Z1, Z0, Z0 @ G2, lambda q, c: ycol(q) @ V1[:, c]), sy_cols.T @ V1, "iteration %d, Y^T V" % (it + 1))      # :238-239
        U0, V0, Z0 = U1, V1, Z1
    ctx.close()


def test_c4_residual_metric_vs_fp64(lib):
    """The device error metric (cmf_residual_sq: the quantity of the stopping test and of reconstruction_err_,
    pycmf/cmf_solvers.py:36-42, :128-130) at BASELINE configs[3]: the value of the whole 65536^2 problem equals the sum over eight
    row / column slabs (each slab a context of its own: the NT pass with other extents), and the first slab -- 8192 rows of X with
    their rows of U, 8192 columns of Y with their rows of Z, after one MU update -- equals ||X_s - U_s V^T||^2, ||Y_s - V Z_s^T||^2
    evaluated in float64 on the host from the device's own data to 1e-5."""
    m = d = p = 65536
    k = 256
    ctx = _synthetic(lib, m, d, p, k)
    whole = np.array(ctx.residual_sq())
    ctx.close()
    parts = np.zeros(2)
    for s in range(8):
        sc = _synthetic(lib, m, d, p, k, s * 8192, s * 8192, 8192, 8192)
        parts += np.array(sc.residual_sq())
        if s == 0:
            sc.mu_step(0.0, 0.0, 7)                                     # the metric on updated factors as well
            ex, ey = sc.residual_sq()
            U, V, Z = (sc.get_factor(w) for w in range(3))
            X = sc.get_data(0).astype(np.float64)
            rx = float(((X - U @ V.T) ** 2).sum())
            del X
            Y = sc.get_data(1).astype(np.float64)
            ry = float(((Y - V @ Z.T) ** 2).sum())
            del Y
            assert abs(ex - rx) <= 1e-5 * rx and abs(ey - ry) <= 1e-5 * ry, (ex, rx, ey, ry)
        sc.close()
    np.testing.assert_allclose(parts, whole, rtol=1e-6)


def _feed(monkeypatch, O, lists):
    """make the oracle's sampler hand out the given index lists, in order (it draws one per row, U / Z sweeps, or two per
    row, V sweep: X side then Y side -- pycmf/cmf_solvers.py:414, :494, :455-456)"""
    it = iter(lists)
    monkeypatch.setattr(O, "draw_sample", lambda n, ratio: next(it))


@pytest.mark.parametrize("x_link", ["linear", "logit"])
def test_c3_full_size_newton_sampled_rows_vs_fp64(lib, monkeypatch, x_link):
    """BASELINE configs[2] (32768 x 16384 / 16384 x 8192, k = 256, y logit, sg_sample_ratio 0.5, device sampler): rows of
    U, Z and V after their sweeps against the oracle's per-row arithmetic (pycmf/cmf_solvers.py:394-508) on the index lists the
    device drew for exactly those rows (cmf_sample_lists) -- which also ties the drawn lists to what the row kernels consumed.
    x_link = 'logit': "sigmoid link" on BOTH sides (the U sweep's logit Hessian carries no l2, :426-428; no row shares partial
    sums with its neighbours: every weight is the row's own)."""
    from oracle import cmf_oracle as O
    from threadpoolctl import threadpool_limits
    m, d, p, k = 32768, 16384, 8192, 256
    alpha, l1, l2, pert, ratio, seed = 0.5, 0.0, 0.1, 0.2, 0.5, 1000
    rng = np.random.RandomState(1)
    ctx = _synthetic(lib, m, d, p, k)
    ctx.fill_data_synthetic(1, 43, 0, 0, 1)            # targets of the logit side: sigmoid(N(0,1)), as bench.py's c3
    if x_link == "logit":
        ctx.fill_data_synthetic(0, 42, 0, 0, 1)
    limit = threadpool_limits(limits=1)                # the oracle's 256 x 256 eigh is 15 x slower on a multi-threaded BLAS
    U0, V0, Z0 = (ctx.get_factor(w) for w in range(3))

    def tol(ref):
        return dict(rtol=0, atol=2e-3 * np.abs(ref).max())

    ctx.newton_step_device_sampled(alpha, l1, l2, x_link, "logit", 0, lib.CMF_UPD_U, pert, ratio, seed)
    U1 = ctx.get_factor(0)
    rows = _spread(m, 14, rng)
    lists = [ctx.sample_lists(0, seed, ratio, i, 1)[0] for i in rows]
    assert all(len(s) == int(d * ratio) and len(np.unique(s)) == len(s) for s in lists)
    Xs = np.vstack([ctx.get_data_block(0, i, 1, 0, d) for i in rows]).astype(np.float64)
    Us = U0[rows].copy()
    _feed(monkeypatch, O, lists)
    O.newton_sweep_U(Us, V0, Xs, alpha, l1, l2, x_link, False, ratio, pert)
    np.testing.assert_allclose(U1[rows], Us, **tol(Us))
    assert np.abs(U1[rows] - U0[rows]).max() > 1e-3 * np.abs(U0).max()

    ctx.newton_step_device_sampled(alpha, l1, l2, x_link, "logit", 0, lib.CMF_UPD_Z, pert, ratio, seed)
    Z1 = ctx.get_factor(2)
    cols = _spread(p, 14, rng)
    lists = [ctx.sample_lists(1, seed, ratio, c, 1)[0] for c in cols]
    Ys = np.hstack([ctx.get_data_block(1, 0, d, c, 1) for c in cols]).astype(np.float64)
    Zs = Z0[cols].copy()
    _feed(monkeypatch, O, lists)
    O.newton_sweep_Z(Zs, V0, Ys, alpha, l1, l2, "logit", False, ratio, pert)
    np.testing.assert_allclose(Z1[cols], Zs, **tol(Zs))

    ctx.newton_step_device_sampled(alpha, l1, l2, x_link, "logit", 0, lib.CMF_UPD_V, pert, ratio, seed)
    V1 = ctx.get_factor(1)
    rows = _spread(d, 10, rng)
    lists = []
    for q in rows:
        lists += [ctx.sample_lists(2, seed, ratio, q, 1)[0], ctx.sample_lists(3, seed, ratio, q, 1)[0]]
    Xs = np.hstack([ctx.get_data_block(0, 0, m, q, 1) for q in rows]).astype(np.float64)
    Ys = np.vstack([ctx.get_data_block(1, q, 1, 0, p) for q in rows]).astype(np.float64)
    Vs = V0[rows].copy()
    _feed(monkeypatch, O, lists)
    O.newton_sweep_V(Vs, U1, Z1, Xs, Ys, alpha, l1, l2, x_link, "logit", False, ratio, pert)
    np.testing.assert_allclose(V1[rows], Vs, **tol(Vs))
    limit.restore_original_limits()
    ctx.close()


@pytest.mark.parametrize("done", [10, 13])
def test_c3_full_size_in_the_clamp_regime_rows_vs_fp64(lib, monkeypatch, done):
    """BASELINE configs[2] at the reference's DEFAULT l2_reg = 0 (pycmf/cmf.py:622) where it actually lives: from
    iteration 7 on `_safe_invert`'s clamp (pycmf/cmf_solvers.py:346-356) acts on EVERY row of U and Z -- 255 of 256 eigenvalues
    of each Hessian below the perturbation (tools/r06_spectrum_probe.py).  Ten iterations on the device, then the sweeps of the
    eleventh one by one: 12 rows of each factor against the oracle's per-row float64 arithmetic on the lists the device drew, at
    the full-size tolerance (2e-3 of the factor's largest entry: float32 Hessians of ||H|| / pert = 1e4), the clamp asserted to
    have acted on every row of the U and Z sweeps (in float32, or redone in float64 where the error bound asked for it).
    done = 10: the transition (the bulk of U's spectrum is crossing the threshold; most rows are redone in float64).  done = 13: the
    steady state every later iteration runs in -- all rows of U and Z through the rank-one route (csrc/cmf_rank1clamp.hip.h),
    asserted via `cmf_newton_clamp_routes`."""
    from oracle import cmf_oracle as O
    from threadpoolctl import threadpool_limits
    m, d, p, k = 32768, 16384, 8192, 256
    alpha, l1, l2, pert, ratio = 0.5, 0.0, 0.0, 0.2, 0.5
    rng = np.random.RandomState(2)
    ctx = _synthetic(lib, m, d, p, k)
    ctx.fill_data_synthetic(1, 43, 0, 0, 1)
    for it in range(done):
        ctx.newton_step_device_sampled(alpha, l1, l2, "linear", "logit", 0, 7, pert, ratio, 1000 + it)
    before = ctx.newton_clamp_stats(full=True)
    routes0 = ctx.newton_clamp_routes()
    seed = 1000 + done
    limit = threadpool_limits(limits=1)
    U0, V0, Z0 = (ctx.get_factor(w) for w in range(3))

    def tol(ref):
        return dict(rtol=0, atol=2e-3 * np.abs(ref).max())

    ctx.newton_step_device_sampled(alpha, l1, l2, "linear", "logit", 0, lib.CMF_UPD_U, pert, ratio, seed)
    after_u = ctx.newton_clamp_stats(full=True)
    assert (after_u[0] - before[0]) + (after_u[2] - before[2]) == m, (before, after_u)     # every row of U: clamped (float32) or refined
    routes_u = ctx.newton_clamp_routes()
    if done == 13:
        assert routes_u[1] - routes0[1] == m and routes_u[0] == routes0[0] and after_u[2] == before[2], (routes0, routes_u, before, after_u)
    U1 = ctx.get_factor(0)
    rows = _spread(m, 12, rng)
    lists = [ctx.sample_lists(0, seed, ratio, i, 1)[0] for i in rows]
    Xs = np.vstack([ctx.get_data_block(0, i, 1, 0, d) for i in rows]).astype(np.float64)
    Us = U0[rows].copy()
    _feed(monkeypatch, O, lists)
    O.newton_sweep_U(Us, V0, Xs, alpha, l1, l2, "linear", False, ratio, pert)
    np.testing.assert_allclose(U1[rows], Us, **tol(Us))
    below = [int((np.linalg.eigvalsh(alpha * V0[s].T @ V0[s]) < pert).sum()) for s in lists[:2]]
    assert min(below) >= 200, below                        # the regime itself: nearly the whole spectrum under the threshold
    assert np.abs(U1[rows] - U0[rows]).max() > 1e-4 * np.abs(U0).max()

    ctx.newton_step_device_sampled(alpha, l1, l2, "linear", "logit", 0, lib.CMF_UPD_Z, pert, ratio, seed)
    after_z = ctx.newton_clamp_stats(full=True)
    assert (after_z[0] - after_u[0]) + (after_z[2] - after_u[2]) == p, (after_u, after_z)
    if done == 13:
        assert ctx.newton_clamp_routes()[1] - routes_u[1] == p
    Z1 = ctx.get_factor(2)
    cols = _spread(p, 12, rng)
    lists = [ctx.sample_lists(1, seed, ratio, c, 1)[0] for c in cols]
    Ys = np.hstack([ctx.get_data_block(1, 0, d, c, 1) for c in cols]).astype(np.float64)
    Zs = Z0[cols].copy()
    _feed(monkeypatch, O, lists)
    O.newton_sweep_Z(Zs, V0, Ys, alpha, l1, l2, "logit", False, ratio, pert)
    np.testing.assert_allclose(Z1[cols], Zs, **tol(Zs))

    ctx.newton_step_device_sampled(alpha, l1, l2, "linear", "logit", 0, lib.CMF_UPD_V, pert, ratio, seed)
    V1 = ctx.get_factor(1)
    rows = _spread(d, 12, rng)
    lists = []
    for q in rows:
        lists += [ctx.sample_lists(2, seed, ratio, q, 1)[0], ctx.sample_lists(3, seed, ratio, q, 1)[0]]
    Xs = np.hstack([ctx.get_data_block(0, 0, m, q, 1) for q in rows]).astype(np.float64)
    Ys = np.vstack([ctx.get_data_block(1, q, 1, 0, p) for q in rows]).astype(np.float64)
    Vs = V0[rows].copy()
    _feed(monkeypatch, O, lists)
    O.newton_sweep_V(Vs, U1, Z1, Xs, Ys, alpha, l1, l2, "linear", "logit", False, ratio, pert)
    np.testing.assert_allclose(V1[rows], Vs, **tol(Vs))
    limit.restore_original_limits()
    ctx.close()


def _c5_problem(lib, y_kind=0, y_param=0.0):
    import scipy.sparse as sp
    m, d, p, k, npr = 1000000, 100000, 64, 256, 100
    rng = np.random.default_rng(42)
    indices = rng.integers(0, d, size=m * npr, dtype=np.int32)
    indices.reshape(m, npr).sort(axis=1)
    X = sp.csr_matrix((np.ones(m * npr), indices, np.arange(0, m * npr + 1, npr, dtype=np.int64)), shape=(m, d))
    X.sum_duplicates()
    ctx = lib.Context(0)
    ctx.set_option("sparse_mode", 2)
    ctx.set_problem(m, d, p, k)
    ctx.set_data(0, X)
    ctx.fill_data_synthetic(1, 43, 0, 0, y_kind, y_param)
    scale = (npr / d / k) ** 0.5
    ctx.fill_factor_synthetic(lib.CMF_U, 101, 0, scale)
    ctx.fill_factor_synthetic(lib.CMF_V, 102, 0, scale)
    ctx.fill_factor_synthetic(lib.CMF_Z, 103, 0, scale)
    return ctx, X, (m, d, p, k)


@pytest.mark.parametrize("y_link", ["linear", "logit"])
def test_c5_full_size_native_csr_sampled_rows_vs_fp64(lib, y_link):
    """BASELINE configs[4] (CSR X 1e6 x 1e5 with 1e8 non-zeros kept native, Y 1e5 x 64, k = 256, Newton).  'linear': the
    bench's c5 workload.  'logit': the reference's own sparse Newton settings (samples/toxic_comments.ipynb:853-856:
    y_link='logit', Y in {0,1}, l1 = 2, l2 = 5, U and V non-negative) -- a dense image of X (400 GB) cannot exist, so passing
    at all shows that X stayed CSR.  Two iterations; after every sweep sampled rows are recomputed with the oracle's
    arithmetic (pycmf/cmf_solvers.py:394-508) from the device's own inputs."""
    from oracle import cmf_oracle as O
    logit = y_link == "logit"
    alpha, pert = 0.5, 0.2
    l1, l2, nnm = (2.0, 5.0, 3) if logit else (0.0, 0.1, 0)
    nn = [bool(nnm & 1), bool(nnm & 2), bool(nnm & 4)]
    ctx, X, (m, d, p, k) = _c5_problem(lib, 2 if logit else 0, 0.1)
    Y = ctx.get_data(1).astype(np.float64)
    rng = np.random.RandomState(2)
    F = [ctx.get_factor(w) for w in range(3)]
    e_prev = None
    for it in range(2):
        # U sweep
        ctx.newton_step(alpha, l1, l2, "linear", y_link, nnm, lib.CMF_UPD_U, pert, 1.0)
        U1 = ctx.get_factor(0)
        rows = _spread(m, 4, rng)
        Us = F[0][rows].copy()
        O.newton_sweep_U(Us, F[1], X[rows], alpha, l1, l2, "linear", nn[0], 1.0, pert)
        np.testing.assert_allclose(U1[rows], Us, rtol=0, atol=2e-3 * max(np.abs(Us).max(), 1e-3 * np.abs(F[0]).max()), err_msg="U it %d" % it)
        # Z sweep
        ctx.newton_step(alpha, l1, l2, "linear", y_link, nnm, lib.CMF_UPD_Z, pert, 1.0)
        Z1 = ctx.get_factor(2)
        cols = _spread(p, 2, rng)
        Zs = F[2][cols].copy()
        O.newton_sweep_Z(Zs, F[1], Y[:, cols], alpha, l1, l2, y_link, nn[2], 1.0, pert)
        np.testing.assert_allclose(Z1[cols], Zs, rtol=0, atol=2e-3 * np.abs(Zs).max(), err_msg="Z it %d" % it)
        # V sweep
        ctx.newton_step(alpha, l1, l2, "linear", y_link, nnm, lib.CMF_UPD_V, pert, 1.0)
        V1 = ctx.get_factor(1)
        rows = _spread(d, 1, rng)
        Vs = F[1][rows].copy()
        O.newton_sweep_V(Vs, U1, Z1, X[:, rows], Y[rows], alpha, l1, l2, "linear", y_link, nn[1], 1.0, pert)
        np.testing.assert_allclose(V1[rows], Vs, rtol=0, atol=2e-3 * np.abs(Vs).max(), err_msg="V it %d" % it)
        F = [U1, V1, Z1]
        assert all(np.isfinite(A).all() for A in F)
    ctx.close()


def test_c5_shape_native_csr_mu_sampled_rows_vs_fp64(lib):
    """The C5 shape under the MU solver (native CSR X 1e6 x 1e5, 1e8 non-zeros: SpMM numerators, SDDMM error): rows of V, U and Z
    after one update recomputed in float64 (pycmf/cmf_solvers.py:230-263), and the sparse error expansion of sklearn
    (cmf_solvers.py:40) against the float64 value on the same factors."""
    eps = float(np.finfo(np.float32).eps)
    ctx, X, (m, d, p, k) = _c5_problem(lib)
    Y = ctx.get_data(1).astype(np.float64)
    rng = np.random.RandomState(3)
    U0, V0, Z0 = (ctx.get_factor(w) for w in range(3))
    ctx.mu_step(0.0, 0.0, lib.CMF_UPD_V)
    V1 = ctx.get_factor(1)
    G = U0.T @ U0 + Z0.T @ Z0
    rows = _spread(d, 4, rng)
    num = np.asarray((X[:, rows].T @ U0)) + Y[rows] @ Z0
    den = V0[rows] @ G
    den[den == 0] = eps
    np.testing.assert_allclose(V1[rows], V0[rows] * num / den, rtol=2e-4, atol=0)
    ctx.mu_step(0.0, 0.0, lib.CMF_UPD_U | lib.CMF_UPD_Z)
    U1, Z1 = ctx.get_factor(0), ctx.get_factor(2)
    G2 = V1.T @ V1
    rows = _spread(m, 6, rng)
    den = U0[rows] @ G2
    den[den == 0] = eps
    np.testing.assert_allclose(U1[rows], U0[rows] * np.asarray(X[rows] @ V1) / den, rtol=2e-4, atol=0)
    den = Z0 @ G2
    den[den == 0] = eps
    np.testing.assert_allclose(Z1, Z0 * (Y.T @ V1) / den, rtol=2e-4, atol=0)
    ex2, ey2 = ctx.residual_sq("linear", "linear")
    # sklearn's expansion ||X||^2 - 2 sum_nnz x_ij (u_i . v_j) + <U^T U, V^T V> (cmf_solvers.py:40) on a row block of 20000 rows in
    # float64 (the whole sum costs a minute of host time; the SDDMM kernel is the same for every row, and
    # test_gpu_sparse.py checks the total at mid size): device value of the block = total of a second context holding that block
    blk = slice(400000, 420000)
    sub = lib.Context(0)
    sub.set_option("sparse_mode", 2)
    sub.set_problem(20000, d, p, k)
    sub.set_data(0, X[blk]); sub.set_data(1, Y)
    sub.set_factor(0, U1[blk]); sub.set_factor(1, V1); sub.set_factor(2, Z1)
    bx2, _ = sub.residual_sq("linear", "linear")
    sub.close()
    coo = X[blk].tocoo()
    cross = float(np.einsum("ij,ij->i", U1[blk][coo.row], V1[coo.col]) @ coo.data)
    want = float(coo.data @ coo.data) - 2.0 * cross + float(np.sum((U1[blk].T @ U1[blk]) * (V1.T @ V1)))
    np.testing.assert_allclose(bx2, want, rtol=1e-4)
    assert 0.0 < ex2 < float(X.data @ X.data)
    np.testing.assert_allclose(ey2, float(np.sum((Y - V1 @ Z1.T) ** 2)), rtol=1e-4)
    ctx.close()
