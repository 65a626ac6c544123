"""Error behaviour of the C ABI and its Python wrapper on a live GPU: wrong call order, bad sizes, missing
data.  Every failure must be a loud exception with the library's message, never a silent fallback."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from pycmf_amd import _lib
    if _lib.device_count() < 1:
        pytest.fail("no GPU visible: the gpu-marked tests need an MI355X")
    return _lib


def test_call_order_and_arguments(lib):
    ctx = lib.Context(0)
    with pytest.raises(ValueError, match="cmf_set_problem has not been called"):
        ctx.mu_step(0.0, 0.0, 7)
    with pytest.raises(ValueError, match="bad problem size"):
        ctx.set_problem(4, 3, 2, 0)
    ctx.set_problem(4, 3, 2, 2)
    with pytest.raises(ValueError, match="X and Y must be set"):
        ctx.mu_step(0.0, 0.0, 7)
    with pytest.raises(ValueError, match="X and Y must be set|X must be set"):
        ctx.newton_step(0.5, 0, 0, "linear", "linear", 7, 7, 0.2, 1.0)
    X, Y = np.ones((4, 3)), np.ones((3, 2))
    ctx.set_data(0, X)
    ctx.set_data(1, Y)
    with pytest.raises(ValueError, match="needs the sample index lists"):
        ctx.newton_step(0.5, 0, 0, "linear", "linear", 7, 7, 0.2, 0.5)   # ratio < 1 without lists
    with pytest.raises(ValueError, match="unknown option"):
        ctx.set_option("no_such_knob", 1)
    with pytest.raises(ValueError, match="gemm_pipe must be"):
        ctx.set_option("gemm_pipe", 99)
    with pytest.raises(ValueError, match="gemm_arith must be"):
        ctx.set_option("gemm_arith", 2)
    with pytest.raises(ValueError, match="operand has 5 rows, expected 3"):
        ctx.data_matmul(0, False, np.ones((5, 2)))
    ctx.close()


def test_bad_device_and_csr(lib):
    with pytest.raises(ValueError, match="out of range"):
        lib.Context(9999)
    import scipy.sparse as sp
    ctx = lib.Context(0)
    ctx.set_option("sparse_mode", 2)
    ctx.set_problem(3, 4, 2, 2)
    bad = sp.csr_matrix((np.ones(2), np.array([0, 7]), np.array([0, 1, 2, 2])), shape=(3, 8))  # column 7 >= d
    with pytest.raises(ValueError, match="column index out of range"):
        ctx.set_data(0, bad)
    ctx.close()


def test_solver_shape_mismatch_messages(lib):
    from pycmf_amd.solver_shell import HipMUSolver
    s = HipMUSolver()
    X, Y = np.ones((5, 4)), np.ones((4, 3))
    with pytest.raises(ValueError, match="X has shape"):
        s.update_step(X, Y, np.ones((6, 2)), np.ones((4, 2)), np.ones((3, 2)), 0, 0, 0.5)
    s.release()


def test_empty_and_degenerate_shapes(lib):
    """p = 1 (README shape), k = 1, and an all-zero X: finite results, zeros stay zero."""
    from pycmf_amd import CMF
    rng = np.random.RandomState(0)
    X, Y = np.abs(rng.randn(5, 4)), np.abs(rng.randn(4, 1))
    for solver in ("mu", "newton"):
        U, V, Z = CMF(n_components=1, solver=solver, x_init="random", y_init="random", random_state=0, max_iter=20).fit_transform(X, Y)
        assert U.shape == (5, 1) and V.shape == (4, 1) and Z.shape == (1, 1)
        assert np.isfinite(U).all() and np.isfinite(V).all() and np.isfinite(Z).all()
    ctx = lib.Context(0)
    ctx.set_problem(5, 4, 1, 2)
    ctx.set_data(0, np.zeros((5, 4))); ctx.set_data(1, Y)
    for w, F in enumerate((np.ones((5, 2)), np.ones((4, 2)), np.ones((1, 2)))):
        ctx.set_factor(w, F)
    ctx.mu_step(0.0, 0.0, 7)
    assert np.isfinite(ctx.get_factor(0)).all() and (ctx.get_factor(0) == 0).all()   # X = 0 -> U numerator 0
    ctx.close()


def test_single_rank_bench_never_imports_torch():
    """bench.py at N = 1 (the driver's headline run) must not pull PyTorch in: its cold import was seen to take many
    minutes on some boxes, inside the driver's own clock around the run."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_no_torch_check.py"), "--workload", "tiny", "--steps", "2",
                          "--warmup", "1", "--no-cpu-baseline"], cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "torch imported: False" in out.stdout, out.stdout
    # ... and neither does a rank of an N > 1 run: the sharded drivers + the RCCL communicator inside libcmfhip (one rank here)
    env = dict(os.environ, CMF_BENCH_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29531")
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "bench_no_torch_check.py"), "--gpus", "1", "--workload", "tiny",
                          "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], cwd=root, capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "torch imported: False" in out.stdout, out.stdout


def test_csr_input_is_validated_canonicalised_and_left_alone(lib):
    """malformed indptr is rejected before anything reads through it; duplicate entries are merged (||A||^2 of
    the sparse error expansion is a sum over MERGED entries, like scipy's sum_duplicates); re-setting data on one problem
    frees the previous images; the caller's matrix is never canonicalised in place."""
    import scipy.sparse as sp
    m, d, p, k = 6, 5, 3, 2
    ctx = lib.Context(0)
    ctx.set_option("sparse_mode", 2)
    ctx.set_problem(m, d, p, k)
    h = ctx._h
    ip = np.array([0, 2, 1, 3, 3, 3, 3], dtype=np.int64)           # not monotonic
    idx = np.array([0, 1, 2], dtype=np.int32)
    val = np.ones(3)
    rc = ctx._lib.cmf_set_data_csr(h, 0, ip.ctypes.data_as(lib._pi64), idx.ctypes.data_as(lib._pi32), val.ctypes.data_as(lib._pd), 3)
    assert rc == 1 and b"not monotonic" in ctx._lib.cmf_last_error()
    ip = np.array([0, 1, 2, 3, 3, 3, 2], dtype=np.int64)           # indptr[rows] != nnz (and not monotonic at the end)
    rc = ctx._lib.cmf_set_data_csr(h, 0, ip.ctypes.data_as(lib._pi64), idx.ctypes.data_as(lib._pi32), val.ctypes.data_as(lib._pd), 3)
    assert rc == 1
    # duplicates and unsorted columns, straight through the C ABI (the Python wrapper would canonicalise a copy first)
    ip = np.array([0, 3, 3, 4, 4, 4, 4], dtype=np.int64)
    idx = np.array([4, 1, 4, 2], dtype=np.int32)                    # row 0: (4, 2.0) (1, 1.0) (4, 3.0)
    val = np.array([2.0, 1.0, 3.0, 4.0])
    for _ in range(2):                                              # twice: the second call replaces the first image
        assert ctx._lib.cmf_set_data_csr(h, 0, ip.ctypes.data_as(lib._pi64), idx.ctypes.data_as(lib._pi32), val.ctypes.data_as(lib._pd), 4) == 0
    ctx.set_data(1, np.ones((d, p)))
    x2, _ = ctx.data_sq()
    assert x2 == 1.0 + 25.0 + 16.0                                  # (2 + 3)^2 for the merged entry
    dense = np.zeros((m, d)); dense[0, 4] = 5.0; dense[0, 1] = 1.0; dense[2, 2] = 4.0
    np.testing.assert_array_equal(ctx.get_data(0), dense.astype(np.float32))
    # the wrapper leaves a non-canonical caller matrix untouched
    A = sp.csr_matrix((val.copy(), idx.copy(), ip.copy()), shape=(m, d))
    assert not A.has_canonical_format
    before = (A.data.copy(), A.indices.copy(), A.indptr.copy())
    ctx.set_data(0, A)
    for a, b in zip(before, (A.data, A.indices, A.indptr)):
        np.testing.assert_array_equal(a, b)
    ctx.close()


def test_scratch_survives_set_problem_and_diagnostics_are_gated(lib):
    """cmf_scratch_alloc buffers live until cmf_scratch_free / cmf_ctx_destroy (a re-sized problem does not free
    them under the drivers that hold one); the timing-only knobs that produce wrong results need CMF_DIAG=1."""
    ctx = lib.Context(0)
    ctx.set_problem(40, 30, 20, 4)
    buf = ctx.scratch(4 * ctx.v_buf_elems())
    ctx.set_problem(50, 30, 20, 4)                                  # used to free the scratch behind the driver's back
    rng = np.random.RandomState(0)
    ctx.set_data(0, np.abs(rng.randn(50, 30))); ctx.set_data(1, np.abs(rng.randn(30, 20)))
    for w, n in enumerate((50, 30, 20)):
        ctx.set_factor(w, np.abs(rng.randn(n, 4)))
    ctx.mu_v_partials(buf.data_ptr())                               # writes the (still live) buffer
    ctx.mu_v_apply(buf.data_ptr(), 0.0, 0.0)
    assert np.isfinite(ctx.get_factor(1)).all()
    buf.release()                                                   # and it is still known to the context
    with pytest.raises(ValueError, match="CMF_DIAG=1"):      # ... and are not even compiled into the default build
        ctx.set_option("row_diag", 1)
    with pytest.raises(ValueError, match="CMF_DIAG=1"):
        ctx.set_option("chol_diag", 2)
    with pytest.raises(NotImplementedError, match="diagnostic builds only"):
        ctx.debug_clock()
    ctx.set_option("row_diag", 0)
    ctx.close()


def test_strided_and_float32_uploads_round_trip(lib):
    """The pinned, multi-threaded staging path with every host layout the reference can hand over (C-ordered,
    F-ordered views from `B.T`, sliced, float32): what comes back is the float32 rounding of what went in, factors are
    written back into the caller's (strided) arrays in place."""
    rng = np.random.RandomState(1)
    m, d, p, k = 700, 2300, 90, 70                                  # 1.6e6 cells: several host threads, rows not a multiple of anything
    big = rng.randn(2 * d, 2 * d)
    ctx = lib.Context(0)
    ctx.set_problem(m, d, p, k)
    for X in (big[:m, :d], np.asfortranarray(big[:m, :d]), big[:2 * m:2, ::2], big[:m, :d].astype(np.float32), big.T[:m, :d]):
        ctx.set_data(0, X)
        np.testing.assert_array_equal(ctx.get_data(0), np.asarray(X, dtype=np.float32))
    F = rng.randn(k, m)
    Ft = F.T                                                         # F-ordered view, like the reference's Z
    ctx.set_factor(0, Ft)
    out = np.zeros((2 * m, 2 * k))
    view = out[::2, ::2]
    ctx.get_factor_into(0, view)
    np.testing.assert_array_equal(view, Ft.astype(np.float32).astype(np.float64))
    assert (out[1::2] == 0).all() and (out[:, 1::2] == 0).all()
    ctx.close()


def test_sample_index_out_of_range_is_rejected(lib):
    """Caller-supplied sample lists (sg_sample_ratio < 1, NumPy stream): an index outside the candidates is an error, not
    a gather from somewhere else (both formulations of the per-row sweeps)."""
    rng = np.random.RandomState(0)
    m, d, p, k = 12, 10, 6, 4
    X, Y = np.abs(rng.randn(m, d)), np.abs(rng.randn(d, p))
    for row_kernel in (1, 0):
        ctx = lib.Context(0)
        ctx.set_option("row_kernel", row_kernel)
        ctx.set_problem(m, d, p, k)
        ctx.set_data(0, X); ctx.set_data(1, Y)
        for w, n in enumerate((m, d, p)):
            ctx.set_factor(w, 0.1 * rng.randn(n, k))
        u = np.tile(np.arange(5, dtype=np.int32), (m, 1)); z = np.tile(np.arange(5, dtype=np.int32), (p, 1))
        vx = np.tile(np.arange(6, dtype=np.int32), (d, 1)); vy = np.tile(np.arange(3, dtype=np.int32), (d, 1))
        ctx.newton_step(0.5, 0.0, 0.1, "linear", "logit", 0, 7, 0.2, 0.5, u, z, vx, vy)      # in range: fine
        bad = u.copy(); bad[3, 2] = d
        with pytest.raises(ValueError, match="outside"):
            ctx.newton_step(0.5, 0.0, 0.1, "linear", "logit", 0, 1, 0.2, 0.5, bad, z, vx, vy)
        neg = vx.copy(); neg[0, 0] = -1
        with pytest.raises(ValueError, match="outside"):
            ctx.newton_step(0.5, 0.0, 0.1, "linear", "logit", 0, 2, 0.2, 0.5, u, z, neg, vy)
        ctx.close()
