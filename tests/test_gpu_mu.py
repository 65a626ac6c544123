"""GPU parity tests for the MU path: HIP (through the C ABI) vs the CPU oracle and
the golden vectors minted from the reference.  Run with -m gpu on an MI355X."""
import numpy as np
import pytest
import scipy.sparse as sp

from conftest import load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from pycmf_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("no GPU visible: the gpu-marked tests need an MI355X")
    return _lib


def _step(lib, X, Y, U, V, Z, l1=0.0, l2=0.0, iters=1, mask=7):
    ctx = lib.Context(0)
    m, k = U.shape
    ctx.set_problem(m, V.shape[0], Z.shape[0], k)
    ctx.set_data(0, X)
    ctx.set_data(1, Y)
    for w, F in enumerate((U, V, Z)):
        ctx.set_factor(w, F)
    for _ in range(iters):
        ctx.mu_step(l1, l2, mask)
    out = [ctx.get_factor(w) for w in range(3)]
    ctx.close()
    return out


@pytest.mark.parametrize("tag,l1,l2", [("plain", 0.0, 0.0), ("reg", 0.3, 0.7)])
@pytest.mark.parametrize("fmt", ["dense", "csr"])
@pytest.mark.parametrize("iters", [1, 10])
def test_mu_golden_steps(lib, tag, l1, l2, fmt, iters):
    g = load_golden("g2_mu_steps")
    X = sp.csr_matrix(g["X"]) if fmt == "csr" else g["X"]
    U, V, Z = _step(lib, X, g["Y"], g["U0"], g["V0"], g["Z0"], l1, l2, iters)
    tol = 2e-5 if iters == 1 else 2e-4  # float32 device arithmetic vs float64 reference
    for n, a in (("U", U), ("V", V), ("Z", Z)):
        np.testing.assert_allclose(a, g["%s_%s_%s%d" % (tag, fmt, n, iters)], rtol=tol, atol=1e-6)


def test_mu_signed_zero_den_partial(lib):
    g = load_golden("g2_mu_steps")
    U, V, Z = _step(lib, g["sX"], g["sY"], g["sU0"], g["sV0"], g["sZ0"])
    for n, a in (("U", U), ("V", V), ("Z", Z)):
        np.testing.assert_allclose(a, g["s%s1" % n], rtol=1e-3, atol=1e-4)  # signed data: cancellation
    U, V, Z = _step(lib, g["zX"], g["zY"], g["zU0"], g["zV0"], g["zZ0"])
    for n, a in (("U", U), ("V", V), ("Z", Z)):
        np.testing.assert_allclose(a, g["z%s1" % n], rtol=2e-5, atol=1e-6)
    U, V, Z = _step(lib, g["X"], g["Y"], g["U0"], g["V0"], g["Z0"], mask=1)
    for n, a in (("U", U), ("V", V), ("Z", Z)):
        np.testing.assert_allclose(a, g["p%s1" % n], rtol=2e-5, atol=1e-6)


def _problem(seed, m, d, p, k):
    rng = np.random.RandomState(seed)
    X, Y = np.abs(rng.randn(m, d)), np.abs(rng.randn(d, p))
    s = np.sqrt(X.mean() / k)
    return X, Y, s * np.abs(rng.randn(m, k)), s * np.abs(rng.randn(d, k)), s * np.abs(rng.randn(p, k))


def test_mu_midsize_residual_parity(lib):
    """G6: 256x192 / 192x96, k=32, 20 iterations; north_star tolerance: residuals within
    1e-4 relative of the float64 reference."""
    g = load_golden("g6_mid_mu")
    rng = np.random.RandomState(42)
    X, Y = np.abs(rng.randn(256, 192)), np.abs(rng.randn(192, 96))
    U0, V0, Z0 = np.abs(rng.randn(256, 32)), np.abs(rng.randn(192, 32)), np.abs(rng.randn(96, 32))
    s = np.sqrt(X.mean() / 32)
    U, V, Z = _step(lib, X, Y, U0 * s, V0 * s, Z0 * s, iters=20)
    ex, ey = np.linalg.norm(X - U @ V.T), np.linalg.norm(Y - V @ Z.T)
    rx, ry = g["errs"][-1]
    assert abs(ex - rx) <= 1e-4 * rx
    assert abs(ey - ry) <= 1e-4 * ry
    np.testing.assert_allclose(U, g["U"], rtol=2e-3, atol=1e-5)


@pytest.mark.parametrize("m,d,p,k", [(5, 4, 1, 4), (300, 260, 70, 5), (513, 257, 255, 37),
                                      (700, 300, 520, 130), (1024, 768, 512, 256), (64, 2000, 40, 64),
                                      (1000, 700, 300, 100), (260, 1030, 2050, 128)])
def test_mu_vs_oracle_ragged(lib, m, d, p, k):
    from oracle import cmf_oracle as O
    X, Y, U0, V0, Z0 = _problem(m + k, m, d, p, k)
    U, V, Z = _step(lib, X, Y, U0, V0, Z0, 0.01, 0.02, iters=3)
    Ur, Vr, Zr = U0.copy(), V0.copy(), Z0.copy()
    for _ in range(3):
        O.mu_update_step(X, Y, Ur, Vr, Zr, 0.01, 0.02)
    for a, b in ((U, Ur), (V, Vr), (Z, Zr)):
        np.testing.assert_allclose(a, b, rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize("m,d,p,k", [(1000, 700, 300, 100), (513, 1030, 2050, 128), (3000, 257, 5000, 70), (256, 256, 256, 65)])
def test_paired_data_passes_match_split_form(lib, m, d, p, k):
    """k_pad = 128: the two data passes of an MU half-iteration as ONE balanced launch (cmf_gemm_pair.hip.h: work units cut
    across output tiles and across the two products) against the split launches it replaces and against the oracle
    (cmf_solvers.py:230-246).  Shapes whose quotas cross tile and product boundaries at odd places."""
    from oracle import cmf_oracle as O
    X, Y, U0, V0, Z0 = _problem(7 * m + k, m, d, p, k)
    out = {}
    for pair in (0, 1):
        ctx = lib.Context(0)
        ctx.set_option("pair_passes", pair)
        ctx.set_problem(m, d, p, k)
        ctx.set_data(0, X)
        ctx.set_data(1, Y)
        for w, F in enumerate((U0, V0, Z0)):
            ctx.set_factor(w, F)
        ctx.kernel_timing(1)
        ctx.kernel_timing_reset()
        for _ in range(3):
            ctx.mu_step(0.01, 0.02, 7)
        launches = ctx.kernel_time("gemm_pair")[1]
        ctx.kernel_timing(False)
        assert launches == (6 if pair else 0), "the paired launch must be the path that ran (or not, when switched off)"
        out[pair] = [ctx.get_factor(w) for w in range(3)]
        ctx.close()
    Ur, Vr, Zr = U0.copy(), V0.copy(), Z0.copy()
    for _ in range(3):
        O.mu_update_step(X, Y, Ur, Vr, Zr, 0.01, 0.02)
    for a, b, r in zip(out[1], out[0], (Ur, Vr, Zr)):
        np.testing.assert_allclose(a, b, rtol=2e-5, atol=1e-7)  # same products, another order of the float32 partial sums
        np.testing.assert_allclose(a, r, rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize("link", ["linear", "logit"])
def test_residual_sq(lib, link):
    from oracle import cmf_oracle as O
    X, Y, U, V, Z = _problem(3, 333, 270, 129, 20)
    if link == "logit":
        X, Y = 1 / (1 + np.exp(-X)), 1 / (1 + np.exp(-Y))
        U, V, Z = U - U.mean(), V - V.mean(), Z - Z.mean()
    ctx = lib.Context(0)
    ctx.set_problem(333, 270, 129, 20)
    ctx.set_data(0, X); ctx.set_data(1, Y)
    for w, F in enumerate((U, V, Z)):
        ctx.set_factor(w, F)
    ex2, ey2 = ctx.residual_sq(link, link)
    x2, y2 = ctx.data_sq()
    Xd = ctx.get_data(0)
    ctx.close()
    np.testing.assert_allclose(np.sqrt(ex2), O.factorization_error(X, U, V.T, link), rtol=1e-5)
    np.testing.assert_allclose(np.sqrt(ey2), O.factorization_error(Y, V, Z.T, link), rtol=1e-5)
    np.testing.assert_allclose(x2, np.sum(X.astype(np.float32).astype(np.float64) ** 2), rtol=1e-6)
    np.testing.assert_array_equal(Xd, X.astype(np.float32))


def test_strided_and_fortran_inputs(lib):
    X, Y, U0, V0, Z0 = _problem(9, 40, 30, 20, 6)
    ref = _step(lib, X, Y, U0, V0, Z0)
    got = _step(lib, np.asfortranarray(X), Y.T.copy().T, np.asfortranarray(U0), V0[::1], Z0.T.copy().T)
    for a, b in zip(ref, got):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("k", [24, 100])
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_protocol_matches_unsharded(lib, world, k):
    """run `world` shards as separate contexts on this one GPU, emulate the
    all-reduce by summing their partial buffers on the device, and compare with the
    unsharded step (same kernels, different partition)."""
    import torch
    from pycmf_amd.sharded import ShardedMU, HipShardBackend, shard_bounds
    m, d, p = 700, 300, 530   # (k = 100: k_pad = 128, the shards' U / Z updates take the paired launch of cmf_gemm_pair.hip.h)
    X, Y, U0, V0, Z0 = _problem(77, m, d, p, k)
    ref = _step(lib, X, Y, U0, V0, Z0, 0.01, 0.02, iters=2)
    ctxs, bufs, bounds = [], [], []
    for r in range(world):
        r0, r1 = shard_bounds(m, world, r)
        c0, c1 = shard_bounds(p, world, r)
        ctx = lib.Context(0)
        ctx.set_problem(r1 - r0, d, c1 - c0, k)
        ctx.set_data(0, X[r0:r1]); ctx.set_data(1, Y[:, c0:c1])
        ctx.set_factor(0, U0[r0:r1]); ctx.set_factor(1, V0); ctx.set_factor(2, Z0[c0:c1])
        ctxs.append(ctx); bounds.append((r0, r1, c0, c1))
        bufs.append(torch.zeros(ctx.v_buf_elems(), dtype=torch.float32, device="cuda:0"))
    torch.cuda.synchronize()   # the fills run on PyTorch's stream, the contexts launch on their own
    for _ in range(2):
        for ctx, b in zip(ctxs, bufs):
            ctx.mu_v_partials(b.data_ptr())
            ctx.sync()
        total = torch.stack(bufs).sum(0)
        for ctx, b in zip(ctxs, bufs):
            b.copy_(total)
            torch.cuda.synchronize()
            ctx.mu_v_apply(b.data_ptr(), 0.01, 0.02)
            ctx.mu_uz_update(0.01, 0.02, 7)
            ctx.sync()
    for ctx, (r0, r1, c0, c1) in zip(ctxs, bounds):
        np.testing.assert_allclose(ctx.get_factor(1), ref[1], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(ctx.get_factor(0), ref[0][r0:r1], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(ctx.get_factor(2), ref[2][c0:c1], rtol=1e-5, atol=1e-7)
        ctx.close()


def test_synthetic_fill_is_partition_independent(lib):
    """bench.py's generator: a shard filled with global offsets equals the slice of the
    full matrix."""
    full = lib.Context(0); full.set_problem(300, 200, 120, 8)
    full.fill_data_synthetic(0, 42); full.fill_data_synthetic(1, 43)
    Xf, Yf = full.get_data(0), full.get_data(1)
    part = lib.Context(0); part.set_problem(100, 200, 40, 8)
    part.fill_data_synthetic(0, 42, 150, 0); part.fill_data_synthetic(1, 43, 0, 60)
    np.testing.assert_array_equal(part.get_data(0), Xf[150:250])
    np.testing.assert_array_equal(part.get_data(1), Yf[:, 60:100])
    assert Xf.min() >= 0 and abs(Xf.mean() - 0.7979) < 0.02 and abs((Xf ** 2).mean() - 1.0) < 0.03
    full.close(); part.close()


@pytest.mark.parametrize("solver", ["mu", "newton"])
def test_graph_replay_matches_eager(lib, solver):
    """cmf_set_option("graph", 1): the MU step and the linear-link Newton step (k <= 64: the float64 shared inverse decides
    on the device, no host round trip) are captured into a hipGraph on their second call and replayed from then on; the
    iterates must be the eager ones bit for bit."""
    rng = np.random.RandomState(4)
    m, d, p, k = 300, 200, 120, 24
    X, Y = np.abs(rng.randn(m, d)), np.abs(rng.randn(d, p))
    F0 = [0.3 * np.abs(rng.randn(n, k)) + 0.05 for n in (m, d, p)]
    outs = []
    for graph in (0, 1):
        ctx = lib.Context(0)
        ctx.set_option("graph", graph)
        ctx.set_problem(m, d, p, k)
        ctx.set_data(0, X); ctx.set_data(1, Y)
        for w, F in enumerate(F0):
            ctx.set_factor(w, F)
        for _ in range(6):
            if solver == "mu":
                ctx.mu_step(0.01, 0.02, 7)
            else:
                ctx.newton_step(0.5, 0.01, 0.3, "linear", "linear", 7, 7, 0.2, 1.0)
        outs.append([ctx.get_factor(w) for w in range(3)])
        ctx.close()
    for a, b in zip(*outs):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("solver", ["mu", "newton"])
def test_in_place_updates_on_column_tiles_are_deterministic(lib, solver):
    """a fused factor update with few row tiles runs on 256 x 64 column tiles (option narrow_update, default on).  Where
    the update is IN PLACE -- the MU epilogue F <- F * P / reg(F G) (pycmf/cmf_solvers.py:212-228) and the re-associated Newton sweep
    F <- clamp(F E + ...) -- the column workgroups of a row tile all stream the whole rows of F, so they must not see each other's
    output: the launch writes a scratch image that is copied over F afterwards.  2048 / 1536 / 1024 rows at k = 200 (k_pad = 256: 8 / 6 /
    4 row tiles, four column tiles each): twenty repetitions, each compared BIT FOR BIT with the one-column-tile form, on a busy device
    (a second context hammers another stream)."""
    rng = np.random.RandomState(11)
    m, d, p, k = 2048, 1536, 1024, 200
    X, Y = np.abs(rng.randn(m, d)), np.abs(rng.randn(d, p))
    s = np.sqrt(X.mean() / k)
    U0, V0, Z0 = s * np.abs(rng.randn(m, k)), s * np.abs(rng.randn(d, k)), s * np.abs(rng.randn(p, k))
    noise = lib.Context(0)                    # keeps the CUs busy from another stream while the updates run
    noise.set_problem(4096, 4096, 4096, 200)
    noise.fill_data_synthetic(0, 1, 0, 0); noise.fill_data_synthetic(1, 2, 0, 0)
    for w, seed in ((0, 3), (1, 4), (2, 5)):
        noise.fill_factor_synthetic(w, seed, 0, 0.05)

    def run(narrow):
        ctx = lib.Context(0)
        ctx.set_problem(m, d, p, k)
        ctx.set_option("narrow_update", narrow)
        ctx.set_data(0, X); ctx.set_data(1, Y)
        for w, F in enumerate((U0, V0, Z0)):
            ctx.set_factor(w, F)
        for _ in range(3):
            noise.mu_step(0.0, 0.0, 7)        # asynchronous: runs beside what follows
            if solver == "mu":
                ctx.mu_step(0.01, 0.02, 7)
            else:                             # linear links, clamp active (non-negative factors, small l2): F E + T (O Hinv) in place
                ctx.newton_step(0.5, 0.0, 0.001, "linear", "linear", 7, 7, 0.2, 1.0)
        out = [ctx.get_factor(w) for w in range(3)]
        ctx.close()
        return out

    ref = run(0)
    assert all(np.isfinite(F).all() for F in ref)
    for rep in range(20):
        got = run(1)
        for name, a, b in zip("UVZ", got, ref):
            assert np.array_equal(a, b), "repetition %d: %s differs from the one-column-tile form in %d entries" % (rep, name, int((a != b).sum()))
    noise.sync()
    noise.close()
