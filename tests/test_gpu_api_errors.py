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
