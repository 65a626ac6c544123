"""cmf_run: the outer loop of the reference (pycmf/cmf_solvers.py:132-195) inside the C ABI -- error at init, one update_step per
iteration, the check every 10th iteration, early stop -- against the same loop driven from Python call by call."""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problem(seed, m, d, p, k, logit=False):
    rng = np.random.RandomState(seed)
    X = np.abs(rng.randn(m, d))
    Y = rng.rand(d, p) if logit else np.abs(rng.randn(d, p))
    s = np.sqrt(X.mean() / k)
    return X, Y, [s * np.abs(rng.randn(n, k)) for n in (m, d, p)]


@pytest.mark.parametrize("solver,kw", [
    ("mu", dict(l1_reg=0.01, l2_reg=0.02, max_iter=300, tol=1e-4)),
    ("mu", dict(max_iter=37, tol=0)),                                   # runs out: n_iter = max_iter, no check at all
    ("mu", dict(max_iter=0, tol=1e-4)),                                 # no iteration: n_iter = 0
    ("mu", dict(l2_reg=0.02, max_iter=25, tol=0, _k=100)),              # k_pad = 128: the paired data-pass launch inside the replayed graph
    ("newton", dict(alpha=0.4, l2_reg=0.3, max_iter=60, tol=1e-4)),     # linear links: shared Hessians, graph-capturable at k <= 64
    ("newton", dict(alpha=0.4, l2_reg=0.05, y_link="logit", U_non_negative=False, V_non_negative=False, Z_non_negative=False,
                    max_iter=30, tol=1e-4)),
    ("newton", dict(alpha=0.5, l2_reg=0.1, sg_sample_ratio=0.5, sg_sampler="device", random_state=5, max_iter=20, tol=1e-4)),
])
def test_c_loop_equals_python_loop(solver, kw, monkeypatch):
    from pycmf_amd.solver_shell import HipMUSolver, HipNewtonSolver
    cls = HipMUSolver if solver == "mu" else HipNewtonSolver
    kw = dict(kw)
    X, Y, F0 = _problem(1, 310, 170, 90, kw.pop("_k", 12), logit=kw.get("y_link") == "logit")
    outs = []
    for host_loop in ("1", "0"):
        monkeypatch.setenv("PYCMF_AMD_HOST_LOOP", host_loop)
        s = cls(**kw)
        U, V, Z = (f.copy() for f in F0)
        _, _, _, n_iter = s.fit_iterative_update(X, Y, U, V, Z)
        err = s.compute_error(X, Y, U, V, Z)
        s.release()
        outs.append((n_iter, U, V, Z, err))
    (n0, U0, V0, Z0, e0), (n1, U1, V1, Z1, e1) = outs
    assert n0 == n1
    if kw["max_iter"] == 0:
        assert n1 == 0
    if kw.get("tol") == 0:
        assert n1 == kw["max_iter"]
    # the same launches in the same order (the C loop replays them from a hipGraph where the step is capturable): bit-identical
    for a, b in ((U0, U1), (V0, V1), (Z0, Z1)):
        np.testing.assert_array_equal(a, b)
    assert e0 == e1


def test_trace_and_verbose_lines(capsys):
    """The C loop returns the error at init and at every check with the elapsed seconds; the solver shell prints the reference's
    verbose lines (pycmf/cmf_solvers.py:178-181, :190-193) from that trace."""
    from pycmf_amd import _lib
    from pycmf_amd.solver_shell import HipMUSolver
    X, Y, F0 = _problem(2, 200, 150, 60, 8)
    ctx = _lib.Context(0)
    ctx.set_problem(200, 150, 60, 8)
    ctx.set_data(0, X); ctx.set_data(1, Y)
    for w, F in enumerate(F0):
        ctx.set_factor(w, F)
    n_iter, errs, secs = ctx.run("mu", 45, 1e-9)
    assert n_iter == 45 and len(errs) == 5 and len(secs) == 5            # init + checks at 10, 20, 30, 40
    assert all(a > b for a, b in zip(errs, errs[1:]))                      # MU decreases the objective
    assert all(a <= b for a, b in zip(secs, secs[1:]))
    ex2, ey2 = ctx.residual_sq()
    assert 0.5 * np.sqrt(ex2) + 0.5 * np.sqrt(ey2) < errs[-1]
    ctx.close()
    s = HipMUSolver(max_iter=25, tol=1e-9, verbose=1)
    U, V, Z = (f.copy() for f in F0)
    s.fit_iterative_update(X, Y, U, V, Z)
    s.release()
    out = capsys.readouterr().out.splitlines()
    assert len(out) == 3 and out[0].startswith("Epoch 10 reached after ") and ", error: " in out[0]
    assert out[1].startswith("Epoch 20 reached after ") and out[2].startswith("Epoch 25 reached after ") and out[2].endswith("seconds.")


def test_reference_benchmark_shape_solve_time():
    """The reference's own benchmark shape (benchmarks/benchmark_cmf.py:42-48: 2000 x 150 / 150 x 10, k = 10, 10 iterations; 0.23-0.50 s
    on 8 vCPU): time of the solve itself (factors resident, one call of the C loop), best of 5."""
    from pycmf_amd import _lib
    X, Y, F0 = _problem(3, 2000, 150, 10, 10)
    ctx = _lib.Context(0)
    ctx.set_problem(2000, 150, 10, 10)
    ctx.set_data(0, X); ctx.set_data(1, Y)
    best = {}
    for solver in ("mu", "newton"):
        times = []
        for rep in range(6):
            for w, F in enumerate(F0):
                ctx.set_factor(w, F)
            ctx.sync()
            t0 = time.perf_counter()
            n_iter, errs, _ = ctx.run(solver, 10, 1e-12, l2=0.1 if solver == "newton" else 0.0, nn_mask=7)
            times.append(time.perf_counter() - t0)
            assert n_iter == 10 and len(errs) == 2 and errs[1] < errs[0]
        best[solver] = min(times[1:])
    ctx.close()
    print("solve time, 10 iterations at 2000 x 150 / 150 x 10, k = 10: mu %.2f ms, newton %.2f ms" % (best["mu"] * 1e3, best["newton"] * 1e3))
    assert best["mu"] < 0.02 and best["newton"] < 0.03


@pytest.mark.parametrize("k,mask", [(12, 7), (100, 7), (256, 7), (40, 5), (40, 3)])
def test_step_with_trace_form_error_equals_step_and_error_pass(k, mask):
    """cmf_mu_step_error: the error metric of the loop's check (pycmf/cmf_solvers.py:36-42, :175-187) from the
    step's own products, ||X||^2 - 2 <U, X V> + <U^T U, V^T V>, against the NT error pass over X and Y on the same factors --
    1e-6 relative on the squared residuals (float32 products, float64 sums) -- and the factors of the step itself against the plain
    step's.  mask 5 / 3: a factor that is not updated has no numerator product to reuse: that side takes the NT pass."""
    from pycmf_amd import _lib
    X, Y, F0 = _problem(3, 700, 330, 210, k)
    res = {}
    for fused in (1, 0):
        ctx = _lib.Context(0)
        ctx.set_problem(700, 330, 210, k)
        ctx.set_data(0, X); ctx.set_data(1, Y)
        for w, F in enumerate(F0):
            ctx.set_factor(w, F)
        for _ in range(3):
            ctx.mu_step(0.01, 0.02, 7)
        if fused:
            sq = ctx.mu_step_error(0.01, 0.02, mask)
        else:
            ctx.mu_step(0.01, 0.02, mask)
            sq = ctx.residual_sq("linear", "linear")
        res[fused] = (sq, [ctx.get_factor(w) for w in range(3)])
        ctx.close()
    for a, b in zip(res[1][0], res[0][0]):
        assert abs(a - b) <= 1e-6 * b, (res[1][0], res[0][0])
    for a, b in zip(res[1][1], res[0][1]):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-6 * np.abs(b).max())


def test_trace_form_falls_back_where_the_expansion_would_cancel():
    """An (almost) exact fit: e^2 << 1e-3 ||X||^2 -- the expansion's three terms of size ||X||^2 cannot resolve it in float32; the
    call must return what the NT pass returns (it IS the NT pass then)."""
    from pycmf_amd import _lib
    rng = np.random.RandomState(2)
    U, V, Z = np.abs(rng.randn(300, 5)), np.abs(rng.randn(200, 5)), np.abs(rng.randn(120, 5))
    X, Y = U @ V.T, V @ Z.T
    out = {}
    for fused in (1, 0):
        ctx = _lib.Context(0)
        ctx.set_problem(300, 200, 120, 5)
        ctx.set_data(0, X); ctx.set_data(1, Y)
        for w, F in enumerate((U * 1.001, V, Z * 0.999)):
            ctx.set_factor(w, F)
        if fused:
            out[fused] = ctx.mu_step_error(0.0, 0.0, 7)
        else:
            ctx.set_option("trace_error", 0)
            out[fused] = ctx.mu_step_error(0.0, 0.0, 7)
        ctx.close()
    assert out[1][0] < 1e-3 * (X ** 2).sum()
    for a, b in zip(out[1], out[0]):
        assert abs(a - b) <= 1e-5 * b + 1e-12, out
