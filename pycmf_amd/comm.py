"""Process-group plumbing of the sharded solvers WITHOUT PyTorch: one process per GPU, the collectives are RCCL calls inside
libcmfhip.so (csrc/cmf_comm.hip.h) enqueued on the context's stream.

SURVEY.md 8(e): the V update of pycmf/cmf_solvers.py:242-246 sums over the row blocks the ranks own -- one all-reduce of the
(d + k) k partial buffer per iteration (MU), a k^2 float64 Gram + the d k partial (linear Newton, re-associated form), three
in-place all-gathers of factor rows (per-row Newton).

Bootstrap: RCCL needs one 128-byte unique id shared by all ranks.  Rank 0 creates it (``cmf_comm_unique_id``) and publishes it
as a file; the others poll for it.  The file lives in ``CMF_COMM_DIR`` (default: the system temp directory) under a name built
from ``CMF_COMM_KEY`` or, by default, the launcher's pid and ``MASTER_PORT`` -- the ranks of one job share their parent (bench.py's
own launcher, ``python -m torch.distributed.run``, ``multi_gpu.fit_multi_gpu``) and nothing else does.

``HostStagedCollectives`` is a test double for boxes with ONE GPU, where RCCL refuses two ranks on one device: the same interface,
every collective staged through host memory and files in a shared directory.  ``CMF_COMM_BACKEND=host`` selects it; it is never
used when every rank has its own GPU.
"""
import os
import tempfile
import time

import numpy as np

from . import _lib


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def _launcher_start():
    """(clock ticks since boot, seconds since the epoch) at which the parent process -- the launcher all ranks of a job share --
    started; (None, None) where /proc is not available."""
    try:
        with open("/proc/%d/stat" % os.getppid()) as f:
            ticks = int(f.read().rsplit(")", 1)[1].split()[19])      # field 22: starttime
        with open("/proc/stat") as f:
            btime = next(int(ln.split()[1]) for ln in f if ln.startswith("btime"))
        return ticks, btime + ticks / os.sysconf("SC_CLK_TCK")
    except Exception:
        return None, None


def _job_key():
    """What the ranks of ONE job share and no other job does: CMF_COMM_KEY when the launcher sets it, else the launcher's pid AND
    start time (a recycled pid gets a different key), MASTER_PORT and torchrun's restart count (a restarted worker group must not
    meet the previous group's files)."""
    key = os.environ.get("CMF_COMM_KEY")
    if key:
        return key
    ticks, _ = _launcher_start()
    return "%d_%s_%s_%s" % (os.getppid(), ticks if ticks is not None else "x", os.environ.get("MASTER_PORT", "0"),
                            os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))


def _job_dir():
    return os.environ.get("CMF_COMM_DIR") or tempfile.gettempdir()


def fresh_job_env(env=None):
    """For launchers (bench.launch_ranks, multi_gpu.fit_multi_gpu): a private, empty rendezvous directory (mkdtemp, mode 0700) and
    a random key for ONE job, as environment entries for its ranks.  Nothing stale can be in it and no other user can write
    there.  Returns (env, directory); the launcher removes the directory when its ranks have exited."""
    import uuid
    env = dict(os.environ if env is None else env)
    d = tempfile.mkdtemp(prefix="cmf_comm_")
    env["CMF_COMM_DIR"] = d
    env["CMF_COMM_KEY"] = uuid.uuid4().hex
    return env, d


def _timeout():
    return float(os.environ.get("CMF_COMM_TIMEOUT", "600"))


def _wait_for(path, timeout, what):
    t0 = time.time()
    while not os.path.exists(path):
        if time.time() - t0 > timeout:
            raise RuntimeError("pycmf_amd.comm: timed out after %.0f s waiting for %s (%s)" % (timeout, what, path))
        time.sleep(0.01)


def _write_private(path, payload):
    """Create `path` with O_EXCL and mode 0600 and write payload atomically (a reader sees all of it or no file).  A file of the
    same name can only be a leftover of THIS launcher instance (the key holds its pid and start time): it is replaced."""
    tmp = "%s.tmp%d" % (path, os.getpid())
    fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
    try:
        os.write(fd, payload)
    finally:
        os.close(fd)
    try:
        os.unlink(path)
    except OSError:
        pass
    os.rename(tmp, path)


def _check_fresh(path):
    """A rendezvous file must belong to this user, be writable by nobody else, and not predate the launcher (a leftover of an
    earlier job whose key happened to repeat would make ncclCommInitRank wait for ranks that no longer exist)."""
    st = os.stat(path)
    if st.st_uid != os.getuid() or (st.st_mode & 0o022):
        raise RuntimeError("pycmf_amd.comm: %s is not a private file of this user (uid %d, mode %o): refusing it"
                           % (path, st.st_uid, st.st_mode & 0o777))
    _, started = _launcher_start()
    if started is not None and st.st_mtime < started - 2.0 and not os.environ.get("CMF_COMM_KEY"):
        raise RuntimeError("pycmf_amd.comm: %s is older than this job's launcher: a leftover of an earlier job; remove it" % path)


def exchange_unique_id(rank, world, timeout=None):
    """The job's RCCL unique id: created by rank 0, read by everybody else."""
    path = os.path.join(_job_dir(), "cmf_comm_%s.id" % _job_key())
    lib = _lib.load()
    if rank == 0:
        import ctypes as C
        buf = C.create_string_buffer(128)
        _lib._touch()
        _lib.check(lib.cmf_comm_unique_id(buf))
        _write_private(path, buf.raw)
        return buf.raw, path
    _wait_for(path, timeout or _timeout(), "rank 0's RCCL unique id")
    _check_fresh(path)
    with open(path, "rb") as f:
        raw = f.read()
    if len(raw) != 128:
        raise RuntimeError("pycmf_amd.comm: %s holds %d bytes, expected 128" % (path, len(raw)))
    return raw, path


class _Group:
    """``with coll.group():`` -- the collectives enqueued inside go to the backend as one group (RCCL: ncclGroupStart / ncclGroupEnd,
    one launch point on the stream; the doubles run them one after the other)."""

    def __init__(self, coll):
        self.coll = coll

    def __enter__(self):
        self.coll._group_start()
        return self

    def __exit__(self, et, ev, tb):
        self.coll._group_end()
        return False


class _ProtocolCheck:
    def group(self):
        return _Group(self)

    def _group_start(self):
        self.launch_points = getattr(self, "launch_points", 0) + 1
        self._in_group = True

    def _group_end(self):
        self._in_group = False

    def self_test(self):
        """Known-answer run of the in-place reduce-scatter and all-gather on this communicator (every rank must call it): True when
        every rank saw the sums / the gathered chunks it had to see.  `make_sharded_mu` falls back to the single all-reduce
        otherwise -- a wrong result of the row-blocked protocol would otherwise only show as a strange residual."""
        ok = True
        try:
            from . import _lib
            w, r, per = self.world, self.rank, 64
            buf = _lib.DeviceArray(self.ctx, w * per, 1)
            i = np.arange(w * per, dtype=np.float32)
            self.ctx.copy_from_host(buf, (r + 1) + i)
            self.reduce_scatter(buf)
            got = self.ctx.copy_to_host(buf).reshape(-1)[r * per:(r + 1) * per]
            ok = np.array_equal(got, w * (w + 1) / 2 + w * i[r * per:(r + 1) * per])
            mine = np.full(w * per, -1.0, dtype=np.float32)
            mine[r * per:(r + 1) * per] = 100.0 * (r + 1) + np.arange(per)
            self.ctx.copy_from_host(buf, mine)
            self.all_gather(buf)
            want = np.concatenate([100.0 * (q + 1) + np.arange(per) for q in range(w)]).astype(np.float32)
            ok = ok and np.array_equal(self.ctx.copy_to_host(buf).reshape(-1), want)
            # the two grouped pairs of the row-blocked MU iteration: {small all-reduce, reduce-scatter}, {small all-reduce, all-gather}
            small = _lib.DeviceArray(self.ctx, 16, 1)
            for second in (self.reduce_scatter, self.all_gather):
                self.ctx.copy_from_host(small, np.arange(16, dtype=np.float32) + r)
                self.ctx.copy_from_host(buf, mine if second == self.all_gather else (r + 1) + i)
                with self.group():
                    self.all_reduce(small)
                    second(buf)
                ok = ok and np.array_equal(self.ctx.copy_to_host(small).reshape(-1), w * np.arange(16, dtype=np.float32) + w * (w - 1) / 2)
                got = self.ctx.copy_to_host(buf).reshape(-1)
                ok = ok and (np.array_equal(got, want) if second == self.all_gather else
                             np.array_equal(got[r * per:(r + 1) * per], w * (w + 1) / 2 + w * i[r * per:(r + 1) * per]))
            small.release()
            buf.release()
        except Exception as e:   # (a rank that raises must still reach the agreement below, or the others wait for it)
            ok = False
            self.self_test_error = repr(e)
        return bool(self.all_reduce_host([0.0 if ok else 1.0], "max")[0] == 0.0)


class RcclCollectives(_ProtocolCheck):
    """The context's RCCL communicator behind the interface the sharded drivers use.  ``all_reduce`` / ``all_gather`` only enqueue
    on the context's stream; ``all_reduce_host`` and ``barrier`` wait."""

    backend = "rccl"

    def __init__(self, ctx, rank, world, timed=False):
        self.ctx, self.rank, self.world = ctx, rank, world
        uid, path = exchange_unique_id(rank, world)
        ctx.comm_init(rank, world, uid)
        ctx.comm_barrier()              # every rank has read the id file
        self.ranks_seen, self.rank_seen = ctx.comm_count()   # RCCL's own view (bench.py prints it)
        if (self.ranks_seen, self.rank_seen) != (world, rank):
            raise RuntimeError("pycmf_amd.comm: RCCL reports rank %d of %d, the launcher said rank %d of %d"
                               % (self.rank_seen, self.ranks_seen, rank, world))
        if rank == 0:
            try:
                os.remove(path)
            except OSError:
                pass
        if timed:
            ctx.comm_timing(True)

    def all_reduce(self, buf):
        self.ctx.comm_allreduce(buf)

    def all_reduce_bg(self, buf):
        """float32 all-reduce on the communicator's side stream: kernels launched afterwards overlap with it (until join)."""
        self.ctx.comm_allreduce_bg(buf)

    def join(self):
        self.ctx.comm_join()

    def exposed_ms(self):
        return self.ctx.comm_exposed_ms()

    def all_gather(self, full, chunk=None):
        self.ctx.comm_allgather(full, full.numel() // self.world)

    def reduce_scatter(self, full):
        """In place: chunk `rank` of the rank's buffer becomes the sum over the ranks of that chunk."""
        self.ctx.comm_reduce_scatter(full, full.numel() // self.world)

    def _group_start(self):
        self.ctx.comm_group_start()

    def _group_end(self):
        self.ctx.comm_group_end()

    def launch_points_seen(self):
        return self.ctx.comm_launch_points()

    def stats_by_kind(self):
        names = ("all_reduce_f32", "all_reduce_f64", "all_gather_f32", "reduce_scatter_f32", "group")
        return {n: self.ctx.comm_stats_kind(k) for k, n in enumerate(names)}

    def all_reduce_host(self, values, op="sum"):
        return self.ctx.comm_allreduce_host(values, op)

    def barrier(self):
        self.ctx.comm_barrier()

    def reset(self):
        self.ctx.comm_stats(reset=True)
        self.ctx.comm_exposed_ms(reset=True)

    def stats(self):
        return self.ctx.comm_stats(reset=False)

    def close(self):
        self.ctx.comm_destroy()


class HostStagedCollectives(_ProtocolCheck):
    """TEST DOUBLE (one GPU shared by all ranks): the same interface, every collective staged through host memory.  A
    collective number s of rank r is the file ``<dir>/<key>_s_r.npy``; a rank publishes its contribution, waits for all of
    them, combines them in rank order (deterministic) and removes its file of collective s - 1 (every rank has read it by then:
    nobody publishes s before finishing s - 1)."""

    backend = "host-staged (test double)"

    def __init__(self, ctx, rank, world, timed=False):
        self.ctx, self.rank, self.world, self.timed = ctx, rank, world, timed
        self.dir, self.key, self.seq = _job_dir(), _job_key(), 0
        self.calls = self.bytes = 0
        self.ms = 0.0
        self.kinds = {}
        self.ranks_seen, self.rank_seen = world, rank
        self.barrier()

    def _path(self, seq, rank):
        return os.path.join(self.dir, "cmf_host_%s_%d_%d.npy" % (self.key, seq, rank))

    def _exchange(self, mine):
        timeout = _timeout()
        self.seq += 1
        import io
        bio = io.BytesIO()
        np.save(bio, mine)
        _write_private(self._path(self.seq, self.rank), bio.getvalue())
        parts = []
        for r in range(self.world):
            if r == self.rank:
                parts.append(mine)
                continue
            _wait_for(self._path(self.seq, r), timeout, "rank %d in collective %d" % (r, self.seq))
            _check_fresh(self._path(self.seq, r))
            parts.append(np.load(self._path(self.seq, r)))
        if self.seq > 1:
            try:
                os.remove(self._path(self.seq - 1, self.rank))
            except OSError:
                pass
        return parts

    def launch_points_seen(self):
        return getattr(self, "launch_points", 0)

    def _account(self, nbytes, t0, kind="all_reduce_f32"):
        dt = (time.perf_counter() - t0) * 1e3
        if not getattr(self, "_in_group", False):
            self.launch_points = getattr(self, "launch_points", 0) + 1
        self.calls += 1
        self.bytes += nbytes
        self.ms += dt
        c, b, m = self.kinds.get(kind, (0, 0, 0.0))
        self.kinds[kind] = (c + 1, b + nbytes, m + dt)

    def stats_by_kind(self):
        return dict(self.kinds)

    def all_reduce(self, buf):
        t0 = time.perf_counter()
        mine = self.ctx.copy_to_host(buf)
        parts = self._exchange(mine)
        total = parts[0].copy()
        for q in parts[1:]:
            total += q
        self.ctx.copy_from_host(buf, total)
        self._account(mine.nbytes, t0, "all_reduce_f64" if mine.dtype == np.float64 else "all_reduce_f32")

    all_reduce_bg = all_reduce          # the double has no streams: the background form is the blocking one

    def join(self):
        pass

    def exposed_ms(self):
        return self.ms

    def all_gather(self, full, chunk=None):
        t0 = time.perf_counter()
        per = full.numel() // self.world
        whole = self.ctx.copy_to_host(full).reshape(-1)
        parts = self._exchange(whole[self.rank * per:(self.rank + 1) * per].copy())
        self.ctx.copy_from_host(full, np.concatenate(parts))
        self._account(whole.nbytes, t0, "all_gather_f32")

    def reduce_scatter(self, full):
        t0 = time.perf_counter()
        per = full.numel() // self.world
        whole = self.ctx.copy_to_host(full).reshape(-1)
        parts = self._exchange(whole)
        lo, hi = self.rank * per, (self.rank + 1) * per
        total = parts[0][lo:hi].copy()
        for q in parts[1:]:
            total += q[lo:hi]
        whole = whole.copy()
        whole[lo:hi] = total            # the other chunks stay as they were, like ncclReduceScatter in place
        self.ctx.copy_from_host(full, whole)
        self._account(whole.nbytes, t0, "reduce_scatter_f32")

    def all_reduce_host(self, values, op="sum"):
        parts = self._exchange(np.ascontiguousarray(values, dtype=np.float64))
        return np.sum(parts, axis=0) if op == "sum" else np.max(parts, axis=0)

    def barrier(self):
        self.all_reduce_host(np.zeros(1))

    def reset(self):
        self.calls = self.bytes = 0
        self.ms = 0.0
        self.kinds = {}
        self.launch_points = 0

    def stats(self):
        self.ctx.sync()
        return self.calls, self.bytes, self.ms

    def close(self):
        """Collective: a last barrier, then rank 0 -- once every other rank has signalled that it has read everything --
        removes the job's files (nobody removes a file another rank may still have to read)."""
        import glob
        self.barrier()
        done = lambda r: os.path.join(self.dir, "cmf_host_%s_done_%d" % (self.key, r))
        if self.rank != 0:
            open(done(self.rank), "wb").close()
            return
        for r in range(1, self.world):
            _wait_for(done(r), _timeout(), "rank %d to finish" % r)
        for f in glob.glob(os.path.join(self.dir, "cmf_host_%s_*" % self.key)):
            try:
                os.remove(f)
            except OSError:
                pass


class NullCollectives(_ProtocolCheck):
    """MEASUREMENT HOOK (``CMF_COMM_BACKEND=null``): rank r of N with no peers, so a single GPU can time the per-rank COMPUTE of a
    sharded run (the shard's GEMM shapes, split-K choices, row-block launches of --overlap-chunks) with the collective excluded.
    The peers' contributions are stood in for by the rank's own: a sum over the ranks returns the rank's buffer TIMES THE WORLD
    SIZE (one elementwise launch on the stream, counted as compute), an all-gather leaves the other ranks' rows as they are
    (the initial V: finite).  The iterates are then those of N identical shards -- finite, of the right magnitude -- instead of the
    NaN a run with summands missing produces (clocks on NaN operands are not the clocks of a real run)."""

    backend = "null (per-rank compute only, measurement hook; sums = own partial x world)"

    def __init__(self, ctx, rank, world, timed=False):
        self.ctx, self.rank, self.world = ctx, rank, world
        self.ranks_seen, self.rank_seen = 1, rank
        self.launch_points = 0

    def _count(self):
        if not getattr(self, "_in_group", False):
            self.launch_points += 1

    def all_reduce(self, buf):
        self._count()
        if buf.element_size() == 4:
            self.ctx.scale(buf, self.world)

    all_reduce_bg = all_reduce

    def reduce_scatter(self, full):
        self._count()
        rows = full.shape[0] // self.world      # chunk `rank` of the rank's own buffer becomes "the sum"
        self.ctx.scale(full[self.rank * rows:(self.rank + 1) * rows], self.world)

    def launch_points_seen(self):
        return self.launch_points

    def stats_by_kind(self):
        return {}

    def join(self):
        pass

    def all_gather(self, full, chunk=None):
        self._count()

    def all_reduce_host(self, values, op="sum"):
        return np.ascontiguousarray(values, dtype=np.float64)

    def barrier(self):
        pass

    def self_test(self):
        return True

    def reset(self):
        self.launch_points = 0

    def stats(self):
        self.ctx.sync()
        return 0, 0, 0.0

    def exposed_ms(self):
        return 0.0

    def close(self):
        pass


def init_collectives(ctx, rank=None, world=None, timed=False, backend=None):
    """The collectives object of this rank: RCCL unless ``CMF_COMM_BACKEND=host`` (test double).  None for a single rank unless
    ``force`` is requested through world > 1."""
    r, w = env_rank_world()
    rank = r if rank is None else rank
    world = w if world is None else world
    backend = backend or os.environ.get("CMF_COMM_BACKEND", "rccl")
    if backend == "host":
        return HostStagedCollectives(ctx, rank, world, timed)
    if backend == "null":
        return NullCollectives(ctx, rank, world, timed)
    if backend != "rccl":
        raise ValueError("CMF_COMM_BACKEND must be 'rccl', 'host' (test double) or 'null' (measurement hook), got %r" % backend)
    return RcclCollectives(ctx, rank, world, timed)
