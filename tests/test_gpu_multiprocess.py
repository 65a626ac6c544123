"""Two real processes, one per rank, sharing the single GPU of the test box: the sharded fit must reproduce the
single-process fit.  RCCL refuses two ranks on one device, so the ranks use the host-staged test double of the collectives
(pycmf_amd/comm.py, CMF_COMM_BACKEND=host: the same interface, staged through host memory); the RCCL path itself runs
with one rank (test_bench_rccl_single_rank, test_rccl_abi_single_rank).  No PyTorch in any rank."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from pycmf_amd.sharded import fit_mu_sharded, shard_bounds
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
d = np.load(%(data)r)
X, Y, U, V, Z = d["X"], d["Y"], d["U"].copy(), d["V"].copy(), d["Z"].copy()
r0, r1 = shard_bounds(X.shape[0], world, rank)
c0, c1 = shard_bounds(Y.shape[1], world, rank)
Ur, Zr = U[r0:r1].copy(), Z[c0:c1].copy()
Ur, V, Zr, n_iter = fit_mu_sharded(X[r0:r1], Y[:, c0:c1], Ur, V, Zr, l1_reg=0.01, l2_reg=0.02, max_iter=40, tol=1e-4, device=0)
np.savez(%(out)r + str(rank) + ".npz", U=Ur, V=V, Z=Zr, n_iter=n_iter, r=np.array([r0, r1, c0, c1]))
assert "torch" not in sys.modules
'''


def _rank_env(r, world, port, tmp_path):
    return dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                CMF_COMM_BACKEND="host", CMF_COMM_DIR=str(tmp_path), CMF_COMM_KEY="t%d" % port, CMF_COMM_TIMEOUT="120")


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize("world,mode", [(2, "rsag"), (3, "rsag"), (2, "allreduce"), (3, "allreduce")])
def test_sharded_fit_two_processes(tmp_path, world, mode):
    """fit_mu_sharded on real rank processes.  'rsag' (the default): reduce-scatter of the partial, V epilogue on the rank's row
    block, all-gather of V in place on the factor -- d_pad = 512 gives blocks of 256 rows, so with three ranks the last block is
    EMPTY (its rank applies nothing and contributes a zero Gram); 'allreduce': one all-reduce, replicated epilogue."""
    from pycmf_amd import _lib
    from pycmf_amd.solver_shell import HipMUSolver
    if _lib.device_count() < 1:
        pytest.fail("no GPU visible")
    rng = np.random.RandomState(0)
    m, d, p, k = 530, 300, 410, 12
    X, Y = np.abs(rng.randn(m, d)), np.abs(rng.randn(d, p))
    s = np.sqrt(X.mean() / k)
    U, V, Z = s * np.abs(rng.randn(m, k)), s * np.abs(rng.randn(d, k)), s * np.abs(rng.randn(p, k))
    data = str(tmp_path / "data.npz")
    np.savez(data, X=X, Y=Y, U=U, V=V, Z=Z)
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "data": data, "out": str(tmp_path / "out")})
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(script)],
                              env=dict(_rank_env(r, world, port, tmp_path), PYCMF_AMD_MU_COLLECTIVE=mode),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [q.communicate(timeout=1800)[0].decode() for q in procs]
    for r, (q, o) in enumerate(zip(procs, outs)):
        assert q.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
    # single-process reference through the ordinary solver shell
    ref = HipMUSolver(l1_reg=0.01, l2_reg=0.02, max_iter=40, tol=1e-4)
    Ur, Vr, Zr = U.copy(), V.copy(), Z.copy()
    _, _, _, n_ref = ref.fit_iterative_update(X, Y, Ur, Vr, Zr)
    ref.release()
    for r in range(world):
        o = np.load(str(tmp_path / "out") + "%d.npz" % r)
        r0, r1, c0, c1 = o["r"]
        assert int(o["n_iter"]) == n_ref
        np.testing.assert_allclose(o["V"], Vr, rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(o["U"], Ur[r0:r1], rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(o["Z"], Zr[c0:c1], rtol=2e-4, atol=1e-6)


NEWTON_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from pycmf_amd.sharded import fit_newton_sharded, block_bounds
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
d_ = np.load(%(data)r)
X, Y, U, V, Z = d_["X"], d_["Y"], d_["U"].copy(), d_["V"].copy(), d_["Z"].copy()
r0, r1 = block_bounds(X.shape[0], world, rank)
q0, q1 = block_bounds(X.shape[1], world, rank)
c0, c1 = block_bounds(Y.shape[1], world, rank)
Ur, Zr = U[r0:r1].copy(), Z[c0:c1].copy()
Ur, V, Zr, n_iter = fit_newton_sharded(X[r0:r1], X[:, q0:q1], Y[:, c0:c1], Y[q0:q1], Ur, V, Zr, alpha=0.4, l1_reg=0.01,
                                       l2_reg=0.05, x_link="linear", y_link="logit", U_non_negative=False,
                                       V_non_negative=False, Z_non_negative=False, hessian_pertubation=0.2,
                                       sg_sample_ratio=0.6, random_state=3, max_iter=20, tol=1e-4, device=0)
np.savez(%(out)r + str(rank) + ".npz", U=Ur, V=V, Z=Zr, n_iter=n_iter, r=np.array([r0, r1, c0, c1]))
assert "torch" not in sys.modules
'''


def test_row_sharded_newton_fit_two_processes(tmp_path):
    """fit_newton_sharded on 2 real ranks (host-staged collectives, one GPU): y logit, sg_sample_ratio 0.6, device sampler -- the same
    iterates as the single-process HipNewtonSolver(sg_sampler='device'), including the stopping iteration."""
    from pycmf_amd import _lib
    from pycmf_amd.solver_shell import HipNewtonSolver
    if _lib.device_count() < 1:
        pytest.fail("no GPU visible")
    world = 2
    rng = np.random.RandomState(2)
    m, d, p, k = 230, 170, 110, 10
    X, Y = np.abs(rng.randn(m, d)), rng.rand(d, p)
    U, V, Z = 0.3 * rng.randn(m, k), 0.3 * rng.randn(d, k), 0.3 * rng.randn(p, k)
    data = str(tmp_path / "data.npz")
    np.savez(data, X=X, Y=Y, U=U, V=V, Z=Z)
    script = tmp_path / "worker.py"
    script.write_text(NEWTON_WORKER % {"root": ROOT, "data": data, "out": str(tmp_path / "out")})
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, str(script)],
                              env=_rank_env(r, world, port, tmp_path),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    outs = [q.communicate(timeout=1800)[0].decode() for q in procs]
    for r, (q, o) in enumerate(zip(procs, outs)):
        assert q.returncode == 0, "rank %d failed:\n%s" % (r, o[-3000:])
    ref = HipNewtonSolver(alpha=0.4, l1_reg=0.01, l2_reg=0.05, x_link="linear", y_link="logit", U_non_negative=False,
                          V_non_negative=False, Z_non_negative=False, hessian_pertubation=0.2, sg_sample_ratio=0.6,
                          random_state=3, max_iter=20, tol=1e-4, sg_sampler="device")
    Ur, Vr, Zr = U.copy(), V.copy(), Z.copy()
    _, _, _, n_ref = ref.fit_iterative_update(X, Y, Ur, Vr, Zr)
    ref.release()
    for r in range(world):
        o = np.load(str(tmp_path / "out") + "%d.npz" % r)
        r0, r1, c0, c1 = o["r"]
        assert int(o["n_iter"]) == n_ref
        # the linear sampled X side shares partial sums inside each shard's groups of four rows: same terms, another order
        # (float32 rounding per step, 20 steps); row_classes = 0 is bit-identical (test_gpu_newton.py)
        np.testing.assert_allclose(o["V"], Vr, rtol=1e-5, atol=2e-5 * np.abs(Vr).max())
        np.testing.assert_allclose(o["U"], Ur[r0:r1], rtol=1e-5, atol=2e-5 * np.abs(Ur).max())
        np.testing.assert_allclose(o["Z"], Zr[c0:c1], rtol=1e-5, atol=2e-5 * np.abs(Zr).max())


def _run_bench(args, env_extra, timeout=1800):
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=timeout)
    assert q.returncode == 0, "bench.py %s failed:\n%s" % (" ".join(args), q.stderr.decode()[-3000:])
    import json
    lines = [ln for ln in q.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line expected, got %d" % len(lines)
    return json.loads(lines[0])


@pytest.mark.parametrize("workload", ["tiny", "tiny3", "tiny5", "tiny5l"])
def test_bench_launches_its_own_ranks(workload):
    """`python bench.py --gpus 2` as the driver calls it -- no launcher, no WORLD_SIZE in the environment: bench.py
    starts the two rank processes itself (before any GPU call in the parent) and prints rank 0's JSON line.  The test
    box has one GPU, so both ranks share it and the host-staged test double stands in for RCCL (which refuses two ranks on
    one device)."""
    out = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1", "--workload", workload, "--no-cpu-baseline", "--mu-collective", "rsag"],
                     {"CMF_BENCH_SAME_DEVICE": "1", "CMF_COMM_BACKEND": "host", "CMF_COMM_TIMEOUT": "120"})
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] > 0
    assert len(out["series_ms"]["per_iteration"]) == 3
    coll = out["collective"]
    assert coll["ranks"] == 2 and coll["payload_bytes_per_iteration"] > 0 and coll["ms_per_iteration"] > 0
    assert coll["ranks_seen"] == 2 and coll["rank_seen"] == 0 and "exposed_ms_per_iteration" in coll and coll["per_kind"]
    # MU (--mu-collective rsag): the one sum cut in two (reduce-scatter, all-gather) + the two k^2 Grams, as TWO groups; per-row
    # Newton: 3 all-gathers of factor rows; linear Newton: the k^2 float64 Gram + ONE d x k partial
    # (tiny5l = the shape of --workload c5l: native CSR X with a logit Y goes through the ROW-sharded driver, X held by rows and by columns)
    assert coll["calls_per_iteration"] == {"tiny": 4, "tiny3": 3, "tiny5": 2, "tiny5l": 3}[workload]
    assert coll["launch_points_per_iteration"] == {"tiny": 2, "tiny3": 3, "tiny5": 2, "tiny5l": 3}[workload]
    if workload == "tiny":
        assert coll["protocol"] == "rsag" and set(coll["per_kind"]) == {"all_reduce_f32", "reduce_scatter_f32", "all_gather_f32"}
    if workload not in ("tiny3", "tiny5l"):
        assert coll["replicas"]["identical"], coll["replicas"]     # every rank ends with the same V, bit for bit
    # (--max-warmup 0: the single-GPU per-row Newton line would otherwise warm up to its steady state, and the two runs would no
    # longer be the same iterations)
    one = _run_bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", workload, "--no-cpu-baseline", "--max-warmup", "0"], {})
    assert one["n_gpus"] == 1 and "collective" not in one
    # same synthetic problem, same iteration count: the sharded run ends at the same residuals (rank 0's shard of X / Y
    # for the sharded run, so compare loosely: both are far from the starting residual and close to each other)
    for key in ("x", "y"):
        assert np.isfinite(one["rel_residual"][key]) and abs(out["rel_residual"][key] - one["rel_residual"][key]) < 0.05 * one["rel_residual"][key]


def test_bench_rccl_single_rank():
    """The RCCL branch of bench.py with one rank (all a 1-GPU box can offer RCCL): unique id through the job file,
    ncclCommInitRank inside libcmfhip, the all-reduce of the (d + k) k partial buffer on the context's stream, teardown."""
    out = _run_bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "tiny", "--no-cpu-baseline", "--mu-collective", "allreduce"],
                     {"CMF_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1"})
    assert out["collective"]["backend"] == "rccl" and out["collective"]["calls_per_iteration"] == 1
    assert out["collective"]["ranks_seen"] == 1 and out["collective"]["rank_seen"] == 0     # ncclCommCount / ncclCommUserRank
    assert out["collective"]["payload_bytes_per_iteration"] == (1024 + 64) * 64 * 4
    ref = _run_bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "tiny", "--no-cpu-baseline"], {})
    assert out["rel_residual"] == ref["rel_residual"]      # a 1-rank all-reduce is the identity: bit-identical iterates
    # the row-blocked protocol: ncclReduceScatter and ncclAllGather in place (one rank: both the identity), each in ONE RCCL group
    # (ncclGroupStart / ncclGroupEnd) with its k^2 all-reduce: four calls, two launch points
    out = _run_bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "tiny", "--no-cpu-baseline", "--mu-collective", "rsag"],
                     {"CMF_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1"})
    coll = out["collective"]
    assert coll["protocol"] == "rsag" and coll["calls_per_iteration"] == 4 and coll["replicas"]["identical"]
    assert coll["launch_points_per_iteration"] == 2 and coll["per_kind"]["group"]["calls_per_iteration"] == 2
    assert coll["payload_bytes_per_iteration"] == 2 * 1024 * 64 * 4 + 2 * 64 * 64 * 4
    assert coll["per_kind"]["reduce_scatter_f32"]["calls_per_iteration"] == 1 and coll["per_kind"]["all_gather_f32"]["calls_per_iteration"] == 1
    for key in ("x", "y"):
        assert abs(out["rel_residual"][key] - ref["rel_residual"][key]) <= 1e-5 * ref["rel_residual"][key]
    # the default: both protocols timed on the live communicator before the warm-up, the decision and both timings in the line; the
    # trial restores the factors, so the iterates are those of the run without it
    out = _run_bench(["--gpus", "1", "--steps", "3", "--warmup", "1", "--workload", "tiny", "--no-cpu-baseline"],
                     {"CMF_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1"})
    coll = out["collective"]
    trial = coll["protocol_trial"]
    assert trial["chosen"] == coll["protocol_chosen"] == coll["protocol"] and trial["chosen"] in ("allreduce", "rsag")
    assert set(trial["ms_per_iteration"]) == {"allreduce", "rsag"} and all(v > 0 for v in trial["ms_per_iteration"].values())
    if trial["chosen"] == "rsag":
        assert trial["ms_per_iteration"]["rsag"] < trial["ms_per_iteration"]["allreduce"] * (1 - trial["margin"])
    assert coll["launch_points_per_iteration"] == (2 if trial["chosen"] == "rsag" else 1)
    for key in ("x", "y"):
        assert abs(out["rel_residual"][key] - ref["rel_residual"][key]) <= 1e-5 * ref["rel_residual"][key]
    # linear Newton on native CSR: float64 Gram + float32 partial per iteration
    out = _run_bench(["--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "tiny5", "--no-cpu-baseline"],
                     {"CMF_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1"})
    assert out["collective"]["backend"] == "rccl" and out["collective"]["calls_per_iteration"] == 2


def test_bench_overlapped_all_reduce_in_row_blocks():
    """--overlap-chunks 4 (opt-in): the (d + k) k buffer of the MU step is formed and all-reduced in four row blocks of V, every
    block's collective on the communicator's side stream under the next block's GEMMs (cmf_mu_v_partials_rows,
    cmf_comm_allreduce_f32_bg, cmf_comm_join).  Same payload, four calls, the same iterates up to the split-K partition of
    the smaller GEMMs; exposed + hidden time add up to the collectives' duration.  RCCL with one rank, and two ranks on the
    host-staged double."""
    base = ["--steps", "3", "--warmup", "1", "--workload", "tiny", "--no-cpu-baseline"]
    env = {"CMF_BENCH_FORCE_DIST": "1", "MASTER_ADDR": "127.0.0.1"}
    serial = _run_bench(["--gpus", "1", "--mu-collective", "allreduce"] + base, env)
    over = _run_bench(["--gpus", "1", "--overlap-chunks", "4"] + base, env)
    cs, co = serial["collective"], over["collective"]
    assert cs["calls_per_iteration"] == 1 and co["calls_per_iteration"] == 4 and co["overlap_chunks"] == 4
    assert co["payload_bytes_per_iteration"] == cs["payload_bytes_per_iteration"] == (1024 + 64) * 64 * 4
    # exposed = stream time spent waiting in the joins, hidden = max(0, collectives - exposed).  With one rank a collective
    # lasts ~5 us, the same order as the event pair around a join, so the two clocks agree only up to that granularity:
    # 0.05 ms absolute slack per iteration (four joins)
    assert co["exposed_ms_per_iteration"] >= 0 and co["hidden_ms_per_iteration"] >= 0
    assert abs(co["exposed_ms_per_iteration"] + co["hidden_ms_per_iteration"] - co["ms_per_iteration"]) < 0.05 + 0.5 * co["ms_per_iteration"]
    for key in ("x", "y"):
        assert abs(over["rel_residual"][key] - serial["rel_residual"][key]) <= 1e-5 * serial["rel_residual"][key]
    two = _run_bench(["--gpus", "2", "--overlap-chunks", "2"] + base,
                     {"CMF_BENCH_SAME_DEVICE": "1", "CMF_COMM_BACKEND": "host", "CMF_COMM_TIMEOUT": "120"})
    assert two["collective"]["calls_per_iteration"] == 2
    for key in ("x", "y"):
        assert abs(two["rel_residual"][key] - serial["rel_residual"][key]) < 0.05 * serial["rel_residual"][key]


def test_rccl_abi_single_rank(tmp_path, monkeypatch):
    """The collectives of the C ABI with a one-rank communicator: in-place all-reduce (float32, float64), all-gather, the
    host-scalar reduction, the barrier, accounting -- and the errors of calling them without a communicator."""
    from pycmf_amd import _lib
    from pycmf_amd.comm import RcclCollectives
    monkeypatch.setenv("CMF_COMM_DIR", str(tmp_path))
    monkeypatch.setenv("CMF_COMM_KEY", "abi")
    ctx = _lib.Context(0)
    ctx.set_problem(8, 8, 8, 4)
    a = _lib.DeviceArray(ctx, 6, 5)
    with pytest.raises(ValueError, match="cmf_comm_init"):
        ctx.comm_allreduce(a)
    coll = RcclCollectives(ctx, 0, 1, timed=True)
    assert not os.path.exists(os.path.join(str(tmp_path), "cmf_comm_abi.id"))   # rank 0 removed the id file after the barrier
    ref = np.arange(30, dtype=np.float32).reshape(6, 5)
    ctx.copy_from_host(a, ref)
    coll.all_reduce(a)
    np.testing.assert_array_equal(ctx.copy_to_host(a), ref)
    g = _lib.DeviceArray(ctx, 3, 3, itemsize=8)
    ctx.copy_from_host(g, np.arange(9, dtype=np.float64).reshape(3, 3) / 7.0)
    coll.all_reduce(g)
    np.testing.assert_array_equal(ctx.copy_to_host(g), np.arange(9, dtype=np.float64).reshape(3, 3) / 7.0)
    coll.all_gather(a)
    np.testing.assert_array_equal(ctx.copy_to_host(a), ref)
    coll.reduce_scatter(a)                       # one rank: its chunk is the whole buffer, the sum is the buffer itself
    np.testing.assert_array_equal(ctx.copy_to_host(a), ref)
    assert (coll.ranks_seen, coll.rank_seen) == (1, 0) == ctx.comm_count()
    np.testing.assert_array_equal(coll.all_reduce_host([1.5, -2.0], "max"), [1.5, -2.0])
    coll.barrier()
    kinds = coll.stats_by_kind()
    assert kinds["all_reduce_f32"][:2] == (1, 120) and kinds["all_reduce_f64"][:2] == (1, 72)
    assert kinds["all_gather_f32"][:2] == (1, 120) and kinds["reduce_scatter_f32"][:2] == (1, 120)
    calls, nbytes, ms = coll.stats()
    assert calls == 4 and nbytes == 120 + 72 + 120 + 120 and ms >= 0.0
    assert coll.launch_points_seen() == 4
    # two collectives as ONE RCCL group (ncclGroupStart / ncclGroupEnd): one more launch point, one event pair around the group
    with coll.group():
        coll.all_reduce(a)
        coll.reduce_scatter(a)
    np.testing.assert_array_equal(ctx.copy_to_host(a), ref)
    assert coll.launch_points_seen() == 5 and coll.stats()[0] == 6
    assert coll.stats_by_kind()["group"][0] == 1 and coll.stats_by_kind()["group"][2] >= 0.0
    with pytest.raises(ValueError, match="without cmf_comm_group_start"):
        ctx.comm_group_end()
    assert coll.self_test()                      # known answers of the plain and the grouped forms
    # cmf_scale_f32 (the measurement double of the collectives)
    ctx.scale(a, 3.0)
    np.testing.assert_array_equal(ctx.copy_to_host(a), 3.0 * ref)
    coll.close()
    ctx.close()


@pytest.mark.parametrize("mode", ["allreduce", "rsag"])
def test_dress_rehearsal_eight_ranks_match_one(tmp_path, mode):
    """The N = 8 code path of the headline configuration on ONE GPU: bench.py --gpus 8, eight rank processes
    sharing the device, host-staged collectives, workload c4q (k = 256, 16384^3: the shard shapes of C4 / 8 scaled by four -- 2048-row
    blocks of U, Z and of the row-blocked V update, i.e. the column-tiled fused updates).  After two iterations every rank holds the
    same V, and 16 fixed rows of each of U, V, Z equal the single-GPU run's to 1e-5 of the factor's largest entry."""
    base = ["--steps", "2", "--warmup", "0", "--workload", "c4q", "--no-cpu-baseline"]
    one = _run_bench(["--gpus", "1", "--dump-rows", str(tmp_path / "one")] + base, {})
    eight = _run_bench(["--gpus", "8", "--mu-collective", mode, "--dump-rows", str(tmp_path / "eight")] + base,
                       {"CMF_BENCH_SAME_DEVICE": "1", "CMF_COMM_BACKEND": "host", "CMF_COMM_TIMEOUT": "300"})
    coll = eight["collective"]
    assert coll["ranks"] == 8 and coll["protocol"] == mode and coll["replicas"]["identical"], coll
    assert coll["launch_points_per_iteration"] == (1 if mode == "allreduce" else 2)
    q = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "compare_rows.py"), str(tmp_path / "one"), str(tmp_path / "eight"),
                        "--tol", "1e-5"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert q.returncode == 0, q.stdout.decode() + q.stderr.decode()[-2000:]
    for key in ("x", "y"):
        assert np.isfinite(eight["rel_residual"][key]) and abs(eight["rel_residual"][key] - one["rel_residual"][key]) < 0.05 * one["rel_residual"][key]


@pytest.mark.parametrize("world", [2, 4])
def test_dress_rehearsal_two_and_four_ranks_with_the_default_protocol_trial(tmp_path, world):
    """the rehearsal at the OTHER rank counts the driver's scaling run takes (N = 2, 4), through the DEFAULT
    protocol selection -- the timed trial on the live ranks, whose cost is in the line (`protocol_trial.seconds`) -- on one GPU with the
    host-staged collectives: replicas identical, 16 rows of each factor equal to the single-GPU run's to 1e-5 of the largest entry."""
    base = ["--steps", "2", "--warmup", "0", "--workload", "c4q", "--no-cpu-baseline"]
    one = _run_bench(["--gpus", "1", "--dump-rows", str(tmp_path / "one")] + base, {})
    many = _run_bench(["--gpus", str(world), "--dump-rows", str(tmp_path / "many")] + base,
                      {"CMF_BENCH_SAME_DEVICE": "1", "CMF_COMM_BACKEND": "host", "CMF_COMM_TIMEOUT": "300"})
    coll = many["collective"]
    assert coll["ranks"] == world and coll["replicas"]["identical"], coll
    trial = coll["protocol_trial"]
    assert trial["chosen"] == coll["protocol"] and trial["seconds"] > 0 and set(trial["ms_per_iteration"]) == {"allreduce", "rsag"}, trial
    assert coll["launch_points_per_iteration"] == (1 if coll["protocol"] == "allreduce" else 2)
    q = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "compare_rows.py"), str(tmp_path / "one"), str(tmp_path / "many"),
                        "--tol", "1e-5"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert q.returncode == 0, q.stdout.decode() + q.stderr.decode()[-2000:]
    for key in ("x", "y"):
        assert np.isfinite(many["rel_residual"][key]) and abs(many["rel_residual"][key] - one["rel_residual"][key]) < 0.05 * one["rel_residual"][key]


@pytest.mark.parametrize("solver,kw", [("mu", {}), ("newton", dict(y_link="logit", U_non_negative=False, V_non_negative=False,
                                                                 Z_non_negative=False, l2_reg=0.05, alpha=0.4))])
def test_cmf_n_gpus_through_the_drop_in_api(solver, kw, monkeypatch):
    """``CMF(n_gpus=2).fit`` -- worker processes started by the front end, one rank per GPU (both on GPU 0 here, the
    host-staged test double instead of RCCL) -- against ``CMF(n_gpus=1)`` from the same initial factors: same iteration count, same factors,
    same reconstruction_err_; sklearn's clone carries n_gpus."""
    from sklearn.base import clone
    from pycmf_amd import CMF
    monkeypatch.setenv("PYCMF_AMD_SAME_DEVICE", "1")
    monkeypatch.setenv("CMF_COMM_BACKEND", "host")
    monkeypatch.setenv("CMF_COMM_TIMEOUT", "120")
    rng = np.random.RandomState(3)
    m, d, p, k = 260, 150, 90, 6
    X = np.abs(rng.randn(m, d))
    Y = rng.rand(d, p) if kw.get("y_link") == "logit" else np.abs(rng.randn(d, p))
    est = CMF(n_components=k, solver=solver, x_init="random", y_init="random", random_state=0, max_iter=30, n_gpus=2, **kw)
    U2, V2, Z2 = clone(est).fit_transform(X, Y)
    two = clone(est); two.fit(X, Y)
    one = clone(est).set_params(n_gpus=1); U1, V1, Z1 = one.fit_transform(X, Y)
    assert two.n_iter_ == one.n_iter_
    np.testing.assert_allclose(two.reconstruction_err_, one.reconstruction_err_, rtol=1e-4)
    tol = 2e-4 if solver == "mu" else 2e-3
    for a, b in ((U2, U1), (V2, V1), (Z2, Z1)):
        np.testing.assert_allclose(a, b, rtol=0, atol=tol * np.abs(b).max())


def test_cmf_n_gpus_initialises_on_rank0_for_large_inputs(monkeypatch):
    """Inputs above DEVICE_SVD_MIN_CELLS with a non-custom init: the parent skips the host initialisers and rank 0 computes the
    start with the device-side ones (whole X, Y on its GPU once) -- the same start, hence the same fit, as n_gpus=1."""
    from pycmf_amd import CMF
    monkeypatch.setenv("PYCMF_AMD_SAME_DEVICE", "1")
    monkeypatch.setenv("CMF_COMM_BACKEND", "host")
    monkeypatch.setenv("CMF_COMM_TIMEOUT", "120")
    rng = np.random.RandomState(5)
    U, V, Z = np.abs(rng.randn(2600, 6)), np.abs(rng.randn(1700, 6)), np.abs(rng.randn(60, 6))
    X, Y = U @ V.T + 0.01 * np.abs(rng.randn(2600, 1700)), V @ Z.T           # 4.4e6 cells
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        two = CMF(n_components=6, solver="mu", random_state=0, max_iter=40, n_gpus=2).fit(X, Y)
        one = CMF(n_components=6, solver="mu", random_state=0, max_iter=40, n_gpus=1).fit(X, Y)
    assert two.n_iter_ == one.n_iter_
    np.testing.assert_allclose(two.reconstruction_err_, one.reconstruction_err_, rtol=1e-3)
    np.testing.assert_allclose(two.components, one.components, rtol=0, atol=2e-3 * np.abs(one.components).max())


def test_cmf_n_gpus_sparse_linear_newton_takes_north_stars_partition(monkeypatch):
    """``CMF(solver='newton', n_gpus=2)`` with linear links on CSR X: the front end routes it to north_star's partition
    (nnz-balanced row blocks of X / U, column blocks of Y / Z, V replicated, X resident once per rank) with, per iteration, the
    k^2 float64 Gram all-reduce and ONE all-reduce of the d x k partial -- not to the three all-gathers of the row-sharded
    form -- and reproduces the single-GPU fit."""
    import scipy.sparse as sp
    from pycmf_amd import CMF
    from pycmf_amd import multi_gpu
    monkeypatch.setenv("PYCMF_AMD_SAME_DEVICE", "1")
    monkeypatch.setenv("CMF_COMM_BACKEND", "host")
    monkeypatch.setenv("CMF_COMM_TIMEOUT", "120")
    rng = np.random.RandomState(4)
    m, d, p, k = 400, 220, 60, 5
    dens = rng.rand(m, 1) * 0.3                                   # heavy-tailed rows: the nnz-balanced cut is not the middle row
    X = sp.csr_matrix(np.abs(rng.randn(m, d)) * (rng.rand(m, d) < dens))
    Y = np.abs(rng.randn(d, p))
    off = multi_gpu.partition(X, Y, "newton", 2, dict(x_link="linear", y_link="linear", sg_sample_ratio=1.0))[0]
    nnz = np.diff(X.indptr)
    assert abs(nnz[:off[1]].sum() - nnz[off[1]:].sum()) <= nnz.max() and off[1] != m // 2
    kw = dict(n_components=k, solver="newton", x_init="random", y_init="random", random_state=0, max_iter=20, l2_reg=0.3,
              U_non_negative=False, V_non_negative=False, Z_non_negative=False)
    two = CMF(n_gpus=2, **kw); U2, V2, Z2 = two.fit_transform(X, Y)
    assert multi_gpu.last_collective_calls[0] == 2 * two.n_iter_
    one = CMF(n_gpus=1, **kw); U1, V1, Z1 = one.fit_transform(X, Y)
    assert two.n_iter_ == one.n_iter_
    np.testing.assert_allclose(two.reconstruction_err_, one.reconstruction_err_, rtol=1e-4)
    for a, b in ((U2, U1), (V2, V1), (Z2, Z1)):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-4 * np.abs(b).max())


@pytest.mark.parametrize("workload,extra", [("tiny", []), ("tiny", ["--mu-collective", "allreduce"]), ("tiny", ["--overlap-chunks", "2"]), ("tiny5", [])])
def test_real_rccl_two_gpus(workload, extra):
    """Boxes with at least two GPUs only (the pool's test boxes have one: skipped there): bench.py --gpus 2 over REAL RCCL -- the
    reduce-scatter / all-gather protocol, the single all-reduce, the side-stream overlap and the float64 Gram all-reduce of the
    linear Newton -- every rank ends with the same V and the residuals match the one-GPU run."""
    from pycmf_amd import _lib
    if _lib.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL refuses two ranks on one device)")
    base = ["--steps", "3", "--warmup", "1", "--workload", workload, "--no-cpu-baseline"]
    two = _run_bench(["--gpus", "2"] + base + extra, {"MASTER_ADDR": "127.0.0.1"})
    one = _run_bench(["--gpus", "1"] + base, {})
    coll = two["collective"]
    assert coll["backend"] == "rccl" and coll["ranks_seen"] == 2 and coll["replicas"]["identical"], coll
    for key in ("x", "y"):
        assert abs(two["rel_residual"][key] - one["rel_residual"][key]) < 0.05 * one["rel_residual"][key]


@pytest.mark.parametrize("solver,kw", [("mu", {}), ("newton", dict(l2_reg=0.3, U_non_negative=False, V_non_negative=False, Z_non_negative=False))])
def test_cmf_n_gpus_transform(solver, kw, monkeypatch):
    """``CMF(n_gpus=2).transform`` (the reference has ONE code path for fit and transform, pycmf/cmf.py:726-747): U and Z are fitted
    against the frozen components on two ranks -- both sweeps are local to a rank -- and equal the single-GPU transform; a
    one-sided transform (Y None) needs nothing sharded and runs on one GPU."""
    from pycmf_amd import CMF
    monkeypatch.setenv("PYCMF_AMD_SAME_DEVICE", "1")
    monkeypatch.setenv("CMF_COMM_BACKEND", "host")
    monkeypatch.setenv("CMF_COMM_TIMEOUT", "120")
    rng = np.random.RandomState(8)
    m, d, p, k = 240, 130, 70, 5
    X, Y = np.abs(rng.randn(m, d)), np.abs(rng.randn(d, p))
    X2, Y2 = np.abs(rng.randn(m, d)), np.abs(rng.randn(d, p))
    one = CMF(n_components=k, solver=solver, x_init="random", y_init="random", random_state=0, max_iter=30, n_gpus=1, **kw).fit(X, Y)
    two = CMF(n_components=k, solver=solver, x_init="random", y_init="random", random_state=0, max_iter=30, n_gpus=2, **kw)
    two.components, two.x_weights, two.y_weights, two.n_components_ = one.components.copy(), one.x_weights.copy(), one.y_weights.copy(), k
    U1, V1, Z1 = one.transform(X2, Y2)
    U2, V2, Z2 = two.transform(X2, Y2)
    np.testing.assert_array_equal(V2, one.components)
    for a, b in ((U2, U1), (Z2, Z1)):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-4 * np.abs(b).max())
    Ua, _, _ = two.transform(X2, None)
    Ub, _, _ = one.transform(X2, None)
    np.testing.assert_allclose(Ua, Ub, rtol=0, atol=1e-6 * np.abs(Ub).max())


FORK_SCRIPT = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from pycmf_amd import CMF, multi_gpu, _lib
rng = np.random.RandomState(3)
X, Y = np.abs(rng.randn(300, 170)), np.abs(rng.randn(170, 90))
assert not multi_gpu.can_fork_ranks()             # opt-in only (ADVICE r5)
os.environ["PYCMF_AMD_FORK_RANKS"] = "1"
assert multi_gpu.can_fork_ranks() and not _lib.gpu_touched()
two = CMF(n_components=6, solver="mu", x_init="random", y_init="random", random_state=0, max_iter=30, n_gpus=2).fit(X, Y)
assert multi_gpu.last_fit_info == {"forked": True, "job_bytes": 0}, multi_gpu.last_fit_info   # the ranks read X, Y in place
assert not _lib.gpu_touched()                     # the parent still has not touched a GPU
one = CMF(n_components=6, solver="mu", x_init="random", y_init="random", random_state=0, max_iter=30, n_gpus=1).fit(X, Y)
assert _lib.gpu_touched() and not multi_gpu.can_fork_ranks()
assert two.n_iter_ == one.n_iter_
np.testing.assert_allclose(two.components, one.components, rtol=0, atol=2e-4 * np.abs(one.components).max())
np.testing.assert_allclose(two.x_weights, one.x_weights, rtol=0, atol=2e-4 * np.abs(one.x_weights).max())
# from now on this process holds a GPU runtime: the ranks are fresh processes and the job travels as float32 files, once
again = CMF(n_components=6, solver="mu", x_init="random", y_init="random", random_state=0, max_iter=30, n_gpus=2).fit(X, Y)
info = multi_gpu.last_fit_info
assert info["forked"] is False and 0 < info["job_bytes"] < 0.6 * (X.nbytes + Y.nbytes) + 200000, info
np.testing.assert_allclose(again.components, two.components, rtol=0, atol=1e-6 * np.abs(two.components).max())
print("fork ok")
'''


def test_cmf_n_gpus_hands_the_data_over_without_a_second_host_copy(tmp_path):
    """``CMF(n_gpus=2)`` from a process that has not touched a GPU yet: the ranks are FORKED off the caller and read X, Y, U, V, Z
    in place (no job files: r04 wrote a float64 copy of everything, 68.7 GB at C4); once the process holds a GPU runtime the
    ranks are fresh processes and the data travel as float32 files (half the bytes).  Same fit either way."""
    script = tmp_path / "fork.py"
    script.write_text(FORK_SCRIPT % {"root": ROOT})
    env = dict(os.environ, PYCMF_AMD_SAME_DEVICE="1", CMF_COMM_BACKEND="host", CMF_COMM_TIMEOUT="120")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    q = subprocess.run([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    assert q.returncode == 0 and "fork ok" in q.stdout.decode(), q.stdout.decode()[-3000:]


def test_bench_under_torch_distributed_run():
    """The driver's own launch line for N > 1 -- `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N ...`: the ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT from the launcher's environment, find
    each other through the launcher's pid + start time + port (pycmf_amd/comm.py), and rank 0 prints ONE JSON line.  Two ranks on the
    one GPU of the test box (host-staged double instead of RCCL, which refuses two ranks per device)."""
    import json
    port = _free_port()
    env = dict(os.environ, CMF_BENCH_SAME_DEVICE="1", CMF_COMM_BACKEND="host", CMF_COMM_TIMEOUT="120")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "CMF_COMM_KEY", "CMF_COMM_DIR"):
        env.pop(k, None)
    q = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--workload", "tiny", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert q.returncode == 0, q.stderr.decode()[-3000:]
    lines = [ln for ln in q.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line expected from rank 0, got %d" % len(lines)
    out = json.loads(lines[0])
    coll = out["collective"]
    assert out["n_gpus"] == 2 and coll["ranks"] == 2 and coll["replicas"]["identical"]
    assert coll["protocol_trial"]["chosen"] == coll["protocol"]          # the default: both protocols timed on the two live ranks
    assert coll["launch_points_per_iteration"] == (2 if coll["protocol"] == "rsag" else 1)


def test_bench_under_torch_distributed_run_with_real_rccl_one_rank():
    """The same launch line with REAL RCCL (all a one-GPU box can offer it: one rank, CMF_BENCH_FORCE_DIST=1 takes the sharded driver
    anyway): the unique id travels through the launcher-keyed file under torchrun's environment (its agent is the ranks' parent, it sets
    MASTER_PORT and TORCHELASTIC_RESTART_COUNT), ncclCommInitRank runs inside libcmfhip, the protocol trial times both MU protocols
    -- grouped collectives included -- on the live communicator."""
    import json
    port = _free_port()
    env = dict(os.environ, CMF_BENCH_FORCE_DIST="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "CMF_COMM_KEY", "CMF_COMM_DIR", "CMF_COMM_BACKEND", "CMF_BENCH_SAME_DEVICE"):
        env.pop(k, None)
    q = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
                        "--workload", "tiny", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert q.returncode == 0, q.stderr.decode()[-3000:]
    lines = [ln for ln in q.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    # the JSON line is the LAST thing on stdout: RCCL's version banner (C stdio, buffered on a pipe) is flushed when the communicator
    # is created, not at exit behind the line a driver may read as "the last line"
    nonempty = [ln for ln in q.stdout.decode().splitlines() if ln.strip()]
    assert nonempty[-1] == lines[0], nonempty[-3:]
    coll = json.loads(lines[0])["collective"]
    assert coll["backend"] == "rccl" and coll["ranks_seen"] == 1 and coll["replicas"]["identical"]
    assert set(coll["protocol_trial"]["ms_per_iteration"]) == {"allreduce", "rsag"}
