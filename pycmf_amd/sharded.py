"""Multi-GPU data-parallel driver: one process per GPU, one all-reduce per iteration.

Partitioning:  rank g owns a row block of X and U and the
matching column block of Y and rows of Z; V is replicated.  The V update needs
sums over all of m and p, so every rank forms its partial

    buf_g = [ X_g^T U_g + Y_g Z_g  (d x k) | U_g^T U_g + Z_g^T Z_g  (k x k) ]

and a single ``all_reduce(sum)`` (RCCL over xGMI, issued from inside libcmfhip on the context's stream:
pycmf_amd/comm.py) turns it into the global numerator/Gram of pycmf/cmf_solvers.py:244-245; every
rank then applies the identical V epilogue and updates its own U_g, Z_g locally.

The local compute is a *backend* object with three methods so that the host
logic (partition arithmetic, collective placement) can be exercised on CPU with
the gloo backend and a test double, while production uses ``HipShardBackend``:

    partials(buf)            fill buf with this shard's partial
    apply_v(buf, l1, l2)     V update from the reduced buffer
    update_uz(l1, l2, mask)  local U / Z update
"""
import contextlib

import numpy as np


def shard_bounds(n, world, rank):
    """Contiguous, balanced partition of range(n): first n % world shards get one extra."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def block_bounds(n, world, rank):
    """Equal blocks of ceil(n / world) rows (the last non-empty one may be shorter, later ones empty): the
    layout an all-gather of equal chunks reassembles in place -- rank r's rows start at r * ceil(n / world)."""
    c = -(-n // world) if world > 0 else n
    lo = min(rank * c, n)
    return lo, min(lo + c, n)


def nnz_balanced_bounds(indptr, world):
    """Contiguous row blocks of a CSR matrix with (nearly) equal numbers of stored values per block
    ("row-shard CSR X by nnz-balanced row blocks"): block g ends at the first row whose
    cumulative count reaches (g + 1) * nnz / world.  Returns world + 1 row offsets."""
    indptr = np.asarray(indptr, dtype=np.int64)
    rows = len(indptr) - 1
    nnz = int(indptr[-1] - indptr[0]) if rows > 0 else 0
    if nnz == 0:
        return np.array([shard_bounds(rows, world, r)[0] for r in range(world)] + [rows], dtype=np.int64)
    targets = indptr[0] + (np.arange(1, world, dtype=np.float64) * nnz / world)
    cuts = np.searchsorted(indptr, targets, side="left")
    # nearest row boundary to the target: the row that straddles it goes to the side that leaves the smaller error
    for i, (c, t) in enumerate(zip(cuts, targets)):
        c = int(min(max(c, 0), rows))
        if c > 0 and abs(indptr[c - 1] - t) <= abs(indptr[c] - t):
            c -= 1
        cuts[i] = c
    off = np.concatenate([[0], np.maximum.accumulate(cuts), [rows]]).astype(np.int64)
    return off


class HipShardBackend:
    """Local shard on one MI355X through libcmfhip (no CPU fallback)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def buf_elems(self):
        return self.ctx.v_buf_elems()

    def partials(self, buf):
        self.ctx.mu_v_partials(buf.data_ptr())

    def apply_v(self, buf, l1, l2):
        self.ctx.mu_v_apply(buf.data_ptr(), l1, l2)

    def update_uz(self, l1, l2, mask):
        self.ctx.mu_uz_update(l1, l2, mask)

    def partials_rows(self, buf, row0, nrows, with_gram):
        self.ctx.mu_v_partials_rows(buf.data_ptr(), row0, nrows, with_gram)

    # row-blocked V update (ShardedMU mode 'rsag')
    def blocked_layout(self, world):
        """(rows per block, d_pad, k_pad); also grows the allocation behind V to world blocks."""
        rows, _ = self.ctx.mu_blocked_layout(world)
        _, dp, _, kp = self.ctx.geometry()
        return rows, dp, kp

    def small_buffers(self):
        from . import _lib
        kp = self.ctx.geometry()[3]
        return _lib.DeviceArray(self.ctx, kp, kp), _lib.DeviceArray(self.ctx, kp, kp)

    def partials_split(self, pbuf, gbuf):
        self.ctx.mu_v_partials_split(pbuf.data_ptr(), gbuf.data_ptr())

    def apply_v_rows(self, p_rows, gbuf, row0, nrows, l1, l2):
        self.ctx.mu_v_apply_rows(p_rows.data_ptr(), gbuf.data_ptr(), row0, nrows, l1, l2)

    def gram_v_rows(self, row0, nrows, g2buf):
        self.ctx.mu_gram_v_rows(row0, nrows, g2buf.data_ptr())

    def update_uz_gram(self, g2buf, l1, l2, mask):
        self.ctx.mu_uz_update_gram(g2buf.data_ptr(), l1, l2, mask)

    def v_full(self, rows):
        """The factor V itself as a (world * block_rows) x k_pad device array: what the all-gather reassembles in place."""
        from . import _lib
        kp = self.ctx.geometry()[3]
        return _lib.DeviceArray(self.ctx, rows, kp, _ptr=self.ctx.factor_dev_ptr(_lib.CMF_V), _owner=None)

    # the three factors saved / restored on the device (the protocol trial of make_sharded_mu runs real iterations)
    def snapshot(self):
        from . import _lib
        m_pad, d_pad, p_pad, kp = self.ctx.geometry()
        saved = []
        for which, rows in ((_lib.CMF_U, m_pad), (_lib.CMF_V, d_pad), (_lib.CMF_Z, p_pad)):
            a = _lib.DeviceArray(self.ctx, max(rows, 1), kp)
            self.ctx.export_factor_rows(which, a.data_ptr())
            saved.append((which, a))
        return saved

    def restore(self, saved):
        for which, a in saved:
            self.ctx.import_factor_rows(which, a.data_ptr())
        self.ctx.sync()

    def drop_snapshot(self, saved):
        for _, a in saved:
            a.release()

    def sync(self):
        self.ctx.sync()

    def row_blocks(self, chunks):
        """256-aligned blocks of the d_pad rows of the partial, at most `chunks` of them."""
        _, dp, _, kp = self.ctx.geometry()
        tiles = dp // 256
        chunks = max(1, min(chunks, tiles))
        cuts = [256 * (tiles * c // chunks) for c in range(chunks + 1)]
        return [(a, b - a) for a, b in zip(cuts, cuts[1:]) if b > a], dp, kp


class ShardedMU:
    """One MU iteration across ``world`` ranks (order V -> U -> Z, cmf_solvers.py:248-263).

    ``mode='allreduce'``: partials -> ONE all-reduce of the (d + k) k buffer -> the identical V epilogue on every rank -> U / Z.
    ``chunks > 1`` (opt-in; needs ``coll`` with background collectives and dense X, Y): the partial is formed in row blocks of
    V, and block c is all-reduced on the communicator's side stream while block c + 1 is computed -- the same single buffer,
    summed in `chunks` pieces.

    ``mode='rsag'``: the same sum cut in two around the epilogue ("Partitioning": reduce-scatter + epilogue +
    all-gather).  The partial P is reduce-scattered in ``world`` equal row blocks, rank r applies ``V_r *= P_r / reg(V_r G)`` to
    ITS block only and forms its share of V^T V, V is all-gathered in place; the two k x k Grams (U^T U + Z^T Z, V^T V) ride in the
    SAME RCCL groups as the two halves (ncclGroupStart / End): two launch points per iteration.  Same bytes on the links as the
    all-reduce; (world - 1) / world of the replicated epilogue and of the d-row Gram disappear from every rank.

    Which of the two runs is MEASURED, not assumed (``make_sharded_mu(mode='auto')``, the default): a few iterations of each on the
    live ranks when the driver is built; north_star's single all-reduce unless the row-blocked form is faster by a clear margin."""

    def __init__(self, backend, buf, world=1, all_reduce=None, chunks=1, coll=None, mode="allreduce", rank=0):
        self.backend = backend
        self.buf = buf
        self.world = world
        self.all_reduce = all_reduce
        self.chunks, self.coll = chunks, coll
        self.mode, self.rank = mode, rank
        self.blocks = None
        if mode not in ("allreduce", "rsag"):
            raise ValueError("mode must be 'allreduce' or 'rsag'")
        if mode == "rsag":
            self.block_rows, self.d_pad, self.kp = backend.blocked_layout(world)   # buf: world * block_rows * k_pad floats
            self.gbuf, self.g2buf = backend.small_buffers()
            lo = min(rank * self.block_rows, self.d_pad)
            self.own = (lo, min(lo + self.block_rows, self.d_pad) - lo)             # (first row, rows) of this rank's block of V
        elif chunks > 1 and coll is not None:
            self.blocks, self.dp, self.kp = backend.row_blocks(chunks)

    def _group(self):
        g = getattr(self.coll, "group", None)
        return g() if g else contextlib.nullcontext()

    def _step_rsag(self, l1, l2, mask):
        """Two launch points on the stream per iteration: {k^2 all-reduce of U^T U + Z^T Z, reduce-scatter of X^T U + Y Z} in front
        of the epilogue, {k^2 all-reduce of the ranks' shares of V^T V, all-gather of V} behind it (RCCL: one group each)."""
        b, coll = self.backend, self.coll
        r0, nr = self.own
        if mask & 2:
            b.partials_split(self.buf, self.gbuf)
            with self._group():
                coll.all_reduce(self.gbuf)                           # k_pad^2 floats
                coll.reduce_scatter(self.buf)                        # first half of the one large sum
            b.apply_v_rows(self.buf[self.rank * self.block_rows:(self.rank + 1) * self.block_rows], self.gbuf, r0, nr, l1, l2)
        if mask & 5:
            b.gram_v_rows(r0, nr, self.g2buf)
        if mask & 7:
            with self._group():
                if mask & 5:
                    coll.all_reduce(self.g2buf)                      # k_pad^2 floats
                if mask & 2:
                    coll.all_gather(b.v_full(self.world * self.block_rows))   # second half, in place on the factor
        if mask & 5:
            b.update_uz_gram(self.g2buf, l1, l2, mask)

    def step(self, l1=0.0, l2=0.0, mask=7):
        if self.mode == "rsag":
            return self._step_rsag(l1, l2, mask)
        if mask & 2:
            if self.blocks and len(self.blocks) > 1:
                last = len(self.blocks) - 1
                for c, (r0, nr) in enumerate(self.blocks):
                    self.backend.partials_rows(self.buf, r0, nr, c == last)
                    # the last block travels with the Gram part, which sits right behind it in the buffer
                    piece = self.buf[r0 * self.kp:(r0 + nr) * self.kp] if c < last else self.buf[r0 * self.kp:self.buf.shape[0]]
                    self.coll.all_reduce_bg(piece)
                self.coll.join()
            else:
                self.backend.partials(self.buf)
                if self.all_reduce is not None:
                    self.all_reduce(self.buf)  # the single collective of the iteration
            self.backend.apply_v(self.buf, l1, l2)
        self.backend.update_uz(l1, l2, mask)


class HipNewtonShardBackend:
    """Linear-link, unsampled Newton step on one shard: cmf_newton_uz_update, then the V sweep either in its re-associated
    three-stage form (cmf_newton_v_gram / _products / _finish: two buffers to sum over the ranks) or in the gradient form
    (cmf_newton_v_partials / _v_apply: one buffer)."""

    def __init__(self, ctx, alpha, nn_mask=0, pert=0.2):
        self.ctx, self.alpha, self.nn_mask, self.pert = ctx, alpha, nn_mask, pert

    def buf_elems(self):
        return self.ctx.v_buf_elems()

    def update_uz(self, l1, l2, mask):
        self.ctx.newton_uz_update(self.alpha, l1, l2, self.nn_mask, mask, self.pert)

    def partials(self, buf):
        self.ctx.newton_v_partials(self.alpha, buf.data_ptr())

    def apply_v(self, buf, l1, l2):
        self.ctx.newton_v_apply(buf.data_ptr(), l1, l2, self.nn_mask, self.pert)

    def gram(self, gbuf):
        self.ctx.newton_v_gram(self.alpha, gbuf.data_ptr())

    def products(self, gbuf, pbuf, l2):
        self.ctx.newton_v_products(self.alpha, l2, self.pert, gbuf.data_ptr(), pbuf.data_ptr())

    def finish(self, pbuf, l1):
        self.ctx.newton_v_finish(pbuf.data_ptr(), l1, self.nn_mask)


class ShardedNewtonLinear:
    """One Newton iteration (order U -> Z -> V, cmf_solvers.py:510-522) across ranks for linear links and
    sg_sample_ratio == 1: U and Z sweeps are local.  The V sweep sums over the ranks

    * with ``gbuf`` (default of the product path, the re-associated form F E + T (O Hinv), csrc/cmf_newton.hip.h): first the
      k_pad^2 float64 Gram  alpha U^T U + (1 - alpha) Z^T Z  (512 KB at k = 256: the inverse must be known BEFORE the data
      pass), then the d x k float32 partial  X^T (alpha U Hinv) + Y ((1 - alpha) Z Hinv)  -- the one large all-reduce;
    * without it (gradient form): one all-reduce of [alpha X^T U + (1 - alpha) Y Z | alpha U^T U + (1 - alpha) Z^T Z]."""

    def __init__(self, backend, buf, world=1, all_reduce=None, gbuf=None):
        self.backend, self.buf, self.world, self.all_reduce, self.gbuf = backend, buf, world, all_reduce, gbuf

    def step(self, l1=0.0, l2=0.0, mask=7):
        self.backend.update_uz(l1, l2, mask)
        if not mask & 2:
            return
        if self.gbuf is not None:
            self.backend.gram(self.gbuf)
            if self.all_reduce is not None:
                self.all_reduce(self.gbuf)
            self.backend.products(self.gbuf, self.buf, l2)
            if self.all_reduce is not None:
                self.all_reduce(self.buf)
            self.backend.finish(self.buf, l1)
        else:
            self.backend.partials(self.buf)
            if self.all_reduce is not None:
                self.all_reduce(self.buf)
            self.backend.apply_v(self.buf, l1, l2)


class HipNewtonRowsBackend:
    """Row-sharded Newton for ANY link / sampling combination ("shard V rows instead for that sweep").

    Every rank drives two contexts on its GPU:
      ctx_uz  problem (m_g, d, p_g):  X rows / Y columns of the rank, V whole      -> sweeps U_g and Z_g
      ctx_v   problem (m, d_g, p):    X columns / Y rows of the rank, U and Z whole -> sweeps V_g
    All three sweeps are row-parallel (pycmf/cmf_solvers.py:394-508), so each context runs the unsharded kernels
    on its rows; what moves between ranks is factor rows only: U, Z before the V sweep, V after it.  The device
    sampler keys by GLOBAL row index (sample_row_offset_*), so the iteration is bit-identical to the unsharded one row by
    row (float32 rounding apart where linear sampled sides share partial sums within each shard's groups of rows).
    """

    def __init__(self, ctx_uz, ctx_v, bounds, shape, alpha, x_link, y_link, nn_mask=0, pert=0.2, ratio=1.0):
        self.ctx_uz, self.ctx_v = ctx_uz, ctx_v
        self.bounds = bounds            # (r0, r1, q0, q1, c0, c1): rows of U, rows of V, rows of Z owned by the rank
        self.shape = shape              # global (m, d, p)
        self.alpha, self.x_link, self.y_link = alpha, x_link, y_link
        self.nn_mask, self.pert, self.ratio = nn_mask, pert, ratio
        r0, _, q0, _, c0, _ = bounds
        ctx_uz.set_option("sample_row_offset_u", r0)
        ctx_uz.set_option("sample_row_offset_z", c0)
        ctx_v.set_option("sample_row_offset_v", q0)
        self.k_pad = ctx_uz.geometry()[3]

    def _step(self, ctx, l1, l2, mask, seed):
        if self.ratio < 1.0:
            ctx.newton_step_device_sampled(self.alpha, l1, l2, self.x_link, self.y_link, self.nn_mask, mask,
                                           self.pert, self.ratio, seed)
        else:
            ctx.newton_step(self.alpha, l1, l2, self.x_link, self.y_link, self.nn_mask, mask, self.pert, 1.0)

    def sweep_uz(self, l1, l2, mask, seed):
        if mask & 5:
            self._step(self.ctx_uz, l1, l2, mask & 5, seed)

    def sweep_v(self, l1, l2, seed):
        self._step(self.ctx_v, l1, l2, 2, seed)

    def rows(self, which):
        """(global rows, first owned row, one-past-last owned row) of factor `which` (0 U, 1 V, 2 Z)."""
        r0, r1, q0, q1, c0, c1 = self.bounds
        return ((self.shape[0], r0, r1), (self.shape[1], q0, q1), (self.shape[2], c0, c1))[which]

    def export_rows(self, which, full):
        """Write the rank's rows of factor `which` into rows [lo, hi) of the staging tensor `full` (a device-to-device
        copy on the contexts' common stream: ordered behind the sweep and in front of the collective by the stream)."""
        _, lo, hi = self.rows(which)
        if hi > lo:
            ctx = self.ctx_v if which == 1 else self.ctx_uz
            ctx.export_factor_rows(which, full[lo:hi].data_ptr())

    def import_rows(self, which, full):
        """Hand the gathered factor (the first n rows of `full`) to the context that holds it whole."""
        ctx = self.ctx_uz if which == 1 else self.ctx_v
        ctx.import_factor_rows(which, full.data_ptr())


class ShardedNewtonRows:
    """One Newton iteration (order U -> Z -> V) with every sweep sharded by rows.  Rows are partitioned by
    ``block_bounds`` (equal blocks of c = ceil(n / world) rows), so a factor is reassembled by ONE in-place
    all-gather of equal chunks: rank r writes its rows at r * c of the staging tensor (world * c rows) and the first
    n rows of the gathered tensor are the factor.  (m + p + d) k_pad floats move per iteration (58 MB at C3) -- half
    of what an all-reduce of zero-padded full tensors moves."""

    def __init__(self, backend, staging, world=1, rank=0, all_gather=None):
        self.backend, self.staging, self.world, self.rank, self.all_gather = backend, staging, world, rank, all_gather

    def _gather(self, which):
        full = self.staging[which]
        self.backend.export_rows(which, full)      # the rank's rows land at rows [lo, hi) = [rank * c, ...)
        if self.all_gather is not None:
            c = full.shape[0] // self.world
            self.all_gather(full, full[self.rank * c:(self.rank + 1) * c])
        self.backend.import_rows(which, full)      # reads the first n rows

    def step(self, l1=0.0, l2=0.0, mask=7, seed=0):
        self.backend.sweep_uz(l1, l2, mask, seed)
        if mask & 2:
            self._gather(0)
            self._gather(2)
            self.backend.sweep_v(l1, l2, seed)
            self._gather(1)


class SingleGpuStep:
    """world == 1: no partial buffer, no collective -- the context's own fused step (what ``CMF.fit`` runs on one GPU:
    ``cmf_mu_step`` / ``cmf_newton_step``)."""

    def __init__(self, fn):
        self._fn = fn

    def step(self, l1=0.0, l2=0.0, mask=7):
        self._fn(l1, l2, mask)


def make_sharded_newton_rows(ctx_uz, ctx_v, bounds, shape, coll, alpha, x_link, y_link, nn_mask=0, pert=0.2, ratio=1.0):
    """bounds = block_bounds of (m, d, p) for this rank.  Both contexts must launch on ONE stream (pass the same handle
    to both constructors): sweeps, row copies and collectives are then ordered by the stream alone, no host syncs.
    ``coll``: the rank's collectives (pycmf_amd/comm.py), created on ``ctx_uz``; None for a single rank."""
    from . import _lib
    world, rank = (coll.world, coll.rank) if coll else (1, 0)
    if ctx_uz.stream_handle() != ctx_v.stream_handle():
        raise ValueError("the U/Z-sweep and V-sweep contexts must share one stream")
    backend = HipNewtonRowsBackend(ctx_uz, ctx_v, bounds, shape, alpha, x_link, y_link, nn_mask, pert, ratio)
    for n, (lo, hi) in zip(shape, (bounds[0:2], bounds[2:4], bounds[4:6])):
        if (lo, hi) != block_bounds(n, world, rank):
            raise ValueError("row-sharded Newton needs block_bounds partitions (got rows [%d, %d) of %d)" % (lo, hi, n))
    staging = [_lib.DeviceArray(ctx_uz, max(world * -(-n // world), 1), backend.k_pad) for n in shape]
    drv = ShardedNewtonRows(backend, staging, world, rank, coll.all_gather if coll else None)
    drv.collectives = coll
    return drv


TRIAL_ITERATIONS = 3      # timed iterations per protocol in the trial of make_sharded_mu(mode='auto') (after one untimed)
TRIAL_MARGIN = 0.02       # the row-blocked protocol must beat north_star's single all-reduce by this fraction to be chosen


def _build_mu_driver(ctx, backend, coll, mode, chunks=1):
    from . import _lib
    if mode == "rsag":
        block_rows, nelem = ctx.mu_blocked_layout(coll.world)
        # (world * block_rows) x k_pad, zero-filled: the rows beyond d_pad of the last blocks stay zero
        buf = _lib.DeviceArray(ctx, coll.world * block_rows, nelem // (coll.world * block_rows))
        drv = ShardedMU(backend, buf, coll.world, coll.all_reduce, coll=coll, mode="rsag", rank=coll.rank)
    else:
        buf = _lib.DeviceArray(ctx, backend.buf_elems(), 1)
        drv = ShardedMU(backend, buf, coll.world, coll.all_reduce, chunks=chunks, coll=coll)
    drv.collectives = coll
    return drv


def _release_mu_driver(drv):
    for name in ("buf", "gbuf", "g2buf"):
        b = getattr(drv, name, None)
        if b is not None and hasattr(b, "release"):
            b.release()


def time_mu_protocols(backend, coll, drivers, iterations=TRIAL_ITERATIONS):
    """Milliseconds per iteration of each driver in `drivers` (name -> ShardedMU) on the LIVE ranks: the factors are saved
    (``backend.snapshot()``), every candidate runs one untimed and `iterations` timed iterations from the same state (stream drained
    and ranks met on both sides of the timed region, the slowest rank's clock), the factors are restored -- also when a candidate
    raises.  Every rank returns the same numbers (they are max-reduced), so every rank takes the same decision."""
    import time
    saved = backend.snapshot()
    out = {}
    try:
        for name, drv in drivers.items():
            try:
                drv.step(0.0, 0.0, 7)
                backend.sync()
                coll.barrier()
                t0 = time.perf_counter()
                for _ in range(iterations):
                    drv.step(0.0, 0.0, 7)
                backend.sync()
                dt = time.perf_counter() - t0
                out[name] = float(coll.all_reduce_host([dt], "max")[0]) / iterations * 1e3
            finally:
                backend.restore(saved)
    finally:
        backend.drop_snapshot(saved)
    return out


def decide_mu_protocol(backend, coll, drivers, iterations=TRIAL_ITERATIONS):
    """The protocol of a sharded MU fit, by the timed trial -- and north_star's single all-reduce WHENEVER THE TRIAL DOES NOT COME
    THROUGH: a candidate that raises (an RCCL error of the grouped forms shows on every rank at the same call),
    a wait that runs into CMF_COMM_TIMEOUT, ranks that cannot agree on the outcome.  Every rank max-reduces a failure flag after
    its trial, so a failure seen by ONE rank decides for all.  Returns (name, record); the record carries the trial's wall-clock
    cost in seconds (it runs 2 x (1 + iterations) iterations and saves / restores the factors: a fit pays it once)."""
    import time
    t0 = time.perf_counter()
    failed, why, ms = 0.0, None, None
    try:
        ms = time_mu_protocols(backend, coll, drivers, iterations)
    except Exception as e:
        failed, why = 1.0, "the timed trial raised %r" % (e,)
    try:
        failed = float(coll.all_reduce_host([failed], "max")[0])
    except Exception as e:
        failed, why = 1.0, why or "the ranks could not agree on the trial's outcome (%r)" % (e,)
    seconds = time.perf_counter() - t0
    if failed:
        return "allreduce", {"chosen": "allreduce", "reason": why or "the timed trial failed on another rank", "seconds": seconds,
                             "timed_iterations": iterations}
    mode = choose_mu_protocol(ms)
    return mode, {"chosen": mode, "ms_per_iteration": ms, "timed_iterations": iterations, "margin": TRIAL_MARGIN, "seconds": seconds,
                  "rule": "rsag only if faster than allreduce by more than the margin (tie-break: north_star's single all-reduce)"}


def choose_mu_protocol(ms, margin=TRIAL_MARGIN):
    """The trial's decision rule: the row-blocked protocol only when it beats north_star's single all-reduce by more than `margin`."""
    return "rsag" if ms["rsag"] < ms["allreduce"] * (1.0 - margin) else "allreduce"


def make_sharded_mu(ctx, coll, chunks=1, mode=None):
    """MU driver of one rank: with ``coll`` None the context's own fused step; else ``mode``

    * ``'allreduce'``: partials -> ONE all-reduce of (d + k) k floats -> replicated epilogue (north_star's protocol; ``chunks`` > 1:
      the buffer reduced in that many row blocks, overlapped with the partials of the next block);
    * ``'rsag'``: reduce-scatter -> epilogue on the rank's row block of V -> all-gather, the two k^2 Grams in the same two groups;
    * ``'auto'`` (default; ``PYCMF_AMD_MU_COLLECTIVE`` overrides): both are built, ``TRIAL_ITERATIONS`` iterations of each are timed
      on the live ranks from the same saved state (``decide_mu_protocol``) and the faster one is kept -- the row-blocked form only
      when it wins by more than ``TRIAL_MARGIN`` and passes the communicator's known-answer test, the single all-reduce whenever
      the trial raises, times out or the ranks disagree; the decision, both timings and the trial's cost in seconds are left in
      ``drv.protocol_trial``.  The choice depends on a wall clock: two fits of the same data may take different protocols, whose
      results differ by float32 summation order (1e-6); ``PYCMF_AMD_MU_COLLECTIVE=allreduce`` (or the recorded choice) pins it --
      ``CMF(random_state=...)`` with ``n_gpus > 1`` does so by itself (pycmf_amd/_worker.py)."""
    import os
    backend = HipShardBackend(ctx)
    if coll is None:
        return SingleGpuStep(lambda l1, l2, mask: ctx.mu_step(l1, l2, mask))
    if mode is None:
        mode = os.environ.get("PYCMF_AMD_MU_COLLECTIVE", "auto") if chunks <= 1 else "allreduce"
    if mode not in ("auto", "rsag", "allreduce"):
        raise ValueError("MU collective mode must be 'auto', 'rsag' or 'allreduce', got %r" % (mode,))
    trial = None
    if mode in ("auto", "rsag") and hasattr(coll, "self_test"):
        try:
            ok, why = bool(coll.self_test()), "known-answer test of the grouped reduce-scatter / all-gather failed"
        except Exception as e:      # an RCCL error inside the grouped forms: every rank sees it at the same call
            ok, why = False, "known-answer test of the grouped reduce-scatter / all-gather raised %r" % (e,)
        if not ok:
            import warnings
            warnings.warn("pycmf_amd: %s on this communicator; falling back to the single all-reduce of the MU iteration" % why, RuntimeWarning)
            trial = {"chosen": "allreduce", "reason": why}
            mode = "allreduce"
    if mode == "auto":
        import time
        t0 = time.perf_counter()
        cands = {"allreduce": _build_mu_driver(ctx, backend, coll, "allreduce"), "rsag": _build_mu_driver(ctx, backend, coll, "rsag")}
        mode, trial = decide_mu_protocol(backend, coll, cands)
        trial["seconds"] = time.perf_counter() - t0          # the candidates' workspaces included
        if "reason" in trial:
            import warnings
            warnings.warn("pycmf_amd: %s; falling back to the single all-reduce of the MU iteration" % trial["reason"], RuntimeWarning)
        drv = cands.pop(mode)
        for other in cands.values():
            _release_mu_driver(other)
    else:
        drv = _build_mu_driver(ctx, backend, coll, mode, chunks)
    drv.protocol_trial = trial
    return drv


def make_sharded_newton(ctx, coll, alpha, nn_mask=0, pert=0.2, single_collective=False):
    """Linear-link Newton driver of one rank.  ``single_collective``: the gradient form with ONE all-reduce of (d + k) k floats
    (round-2 protocol) instead of the re-associated form's k^2 float64 Gram + d k partial."""
    from . import _lib
    backend = HipNewtonShardBackend(ctx, alpha, nn_mask, pert)
    if coll is None:
        return SingleGpuStep(lambda l1, l2, mask: ctx.newton_step(alpha, l1, l2, "linear", "linear", nn_mask, mask, pert, 1.0))
    _, dp, _, kp = ctx.geometry()
    gbuf = None
    if not single_collective:
        # the re-associated three-stage sweep needs the float64 shared Hessian (k_pad <= 1024, options shared_hessian_f64 and
        # newton_reassoc on): where the context reports it unavailable, every rank falls back to the gradient form with its ONE
        # all-reduce -- what the single-GPU step does in the same configuration (every rank holds the same options, so they agree)
        gbuf = _lib.DeviceArray(ctx, kp, kp, itemsize=8)
        try:
            backend.gram(gbuf)
        except NotImplementedError:
            gbuf.release()
            gbuf, single_collective = None, True
    if single_collective:
        drv = ShardedNewtonLinear(backend, _lib.DeviceArray(ctx, backend.buf_elems(), 1), coll.world, coll.all_reduce)
    else:
        drv = ShardedNewtonLinear(backend, _lib.DeviceArray(ctx, dp, kp), coll.world, coll.all_reduce, gbuf=gbuf)
    drv.collectives = coll
    return drv


def _outer_loop(step, global_error, max_iter, tol, verbose):
    """The reference's outer loop (pycmf/cmf_solvers.py:132-195): error at init, one step per iteration, the convergence test
    every 10th iteration on the GLOBAL error."""
    previous = at_init = global_error()
    n_iter = 0
    for n_iter in range(1, max_iter + 1):
        step(n_iter)
        if tol > 0 and n_iter % 10 == 0:
            err = global_error()
            if verbose:
                print("Epoch %02d, error: %f" % (n_iter, err))
            if (previous - err) / at_init < tol:
                break
            previous = err
    return n_iter


def _collectives_for(ctx, rank, world):
    from .comm import env_rank_world, init_collectives
    r, w = env_rank_world()
    rank = r if rank is None else rank
    world = w if world is None else world
    return (init_collectives(ctx, rank, world) if world > 1 else None), rank, world


def fit_mu_sharded(X_rows, Y_cols, U_rows, V, Z_rows, l1_reg=0.0, l2_reg=0.0, max_iter=200, tol=1e-4,
                   device=0, verbose=0, stats=None, rank=None, world=None, update_mask=7, collective=None):
    """Data-parallel MU fit: call from every rank (one process per GPU; RANK / WORLD_SIZE from the environment unless given).

    Rank g passes its row block of X (and the matching rows of U), the matching column block of Y (and rows
    of Z) and the full V (identical on every rank).  Runs the reference's outer loop
    (pycmf/cmf_solvers.py:132-195: step, convergence test every 10th iteration on the *global* error
    0.5||X-UV^T|| + 0.5||Y-VZ^T||) with one all-reduce per iteration for V and one 2-float all-reduce per
    convergence check.  U_rows, V, Z_rows are updated in place; returns (U_rows, V, Z_rows, n_iter).
    """
    from . import _lib
    ctx = _lib.Context(device)
    ctx.set_problem(X_rows.shape[0], X_rows.shape[1], Y_cols.shape[1], V.shape[1])
    ctx.set_data(0, X_rows)
    ctx.set_data(1, Y_cols)
    for which, F in ((_lib.CMF_U, U_rows), (_lib.CMF_V, V), (_lib.CMF_Z, Z_rows)):
        ctx.set_factor(which, F)
    coll, rank, world = _collectives_for(ctx, rank, world)
    # a fit of a few iterations is over before a protocol trial (2 x (1 + TRIAL_ITERATIONS) iterations) could pay for itself:
    # north_star's single all-reduce then, unless the environment pins a protocol
    import os
    short = max_iter < 8 * (1 + TRIAL_ITERATIONS) and "PYCMF_AMD_MU_COLLECTIVE" not in os.environ
    # ``collective``: a protocol pinned by the caller (the front end pins 'allreduce' for a fit with a random_state: the trial's
    # choice hangs on a wall clock, and the two protocols differ in float32 summation order -- a seeded fit must not); the
    # environment variable still wins
    if collective is not None and "PYCMF_AMD_MU_COLLECTIVE" not in os.environ:
        drv = make_sharded_mu(ctx, coll, mode=collective)
    else:
        drv = make_sharded_mu(ctx, coll, mode="allreduce" if short else None)
    if stats is not None:
        stats["mu_protocol"] = getattr(drv, "mode", None)
        stats["mu_protocol_trial"] = getattr(drv, "protocol_trial", None)

    def global_sq():
        sq = np.array(ctx.residual_sq("linear", "linear"))
        return tuple(coll.all_reduce_host(sq)) if coll else tuple(sq)

    def global_error():
        ex2, ey2 = global_sq()
        return 0.5 * np.sqrt(ex2) + 0.5 * np.sqrt(ey2)

    # update_mask: bit 0 U, 1 V, 2 Z (the reference's update_U / update_V / update_Z, pycmf/cmf_solvers.py:252-261; `transform` fits
    # U and Z against a fixed V: both sweeps are local to the rank)
    n_iter = _outer_loop(lambda it: drv.step(l1_reg, l2_reg, update_mask), global_error, max_iter, tol, verbose)
    ctx.sync()
    if stats is not None:   # squared global residuals of the final factors (reconstruction_err_ of the front end)
        stats["ex2"], stats["ey2"] = global_sq()
    for which, F in ((_lib.CMF_U, U_rows), (_lib.CMF_V, V), (_lib.CMF_Z, Z_rows)):
        ctx.get_factor_into(which, F)
    if coll:
        coll.barrier()
        coll.close()
    ctx.close()
    return U_rows, V, Z_rows, n_iter


def fit_newton_linear_sharded(X_rows, Y_cols, U_rows, V, Z_rows, alpha=0.5, l1_reg=0.0, l2_reg=0.0, U_non_negative=True,
                              V_non_negative=True, Z_non_negative=True, hessian_pertubation=0.2, max_iter=200, tol=1e-4,
                              device=0, verbose=0, stats=None, rank=None, world=None, single_collective=False, update_mask=7):
    """Newton fit with linear links and sg_sample_ratio == 1 on north_star's partition -- the same one as MU: rank g holds its row
    block of X / U (CSR X: nnz-balanced blocks) and column block of Y / rows of Z, V replicated, X and Y resident ONCE.  The U
    and Z sweeps are local; the V sweep sums the k^2 Gram and then the d x k partial over the ranks (ShardedNewtonLinear).
    Outer loop and convergence test as in the reference (pycmf/cmf_solvers.py:132-195) on the global error
    alpha ||X - U V^T|| + (1 - alpha) ||Y - V Z^T||."""
    from . import _lib
    ctx = _lib.Context(device)
    ctx.set_problem(X_rows.shape[0], X_rows.shape[1], Y_cols.shape[1], V.shape[1])
    ctx.set_data(0, X_rows)
    ctx.set_data(1, Y_cols)
    for which, F in ((_lib.CMF_U, U_rows), (_lib.CMF_V, V), (_lib.CMF_Z, Z_rows)):
        ctx.set_factor(which, F)
    coll, rank, world = _collectives_for(ctx, rank, world)
    nn_mask = (1 if U_non_negative else 0) | (2 if V_non_negative else 0) | (4 if Z_non_negative else 0)
    drv = make_sharded_newton(ctx, coll, alpha, nn_mask, hessian_pertubation, single_collective)

    def global_sq():
        sq = np.array(ctx.residual_sq("linear", "linear"))
        return tuple(coll.all_reduce_host(sq)) if coll else tuple(sq)

    def global_error():
        ex2, ey2 = global_sq()
        return alpha * np.sqrt(ex2) + (1 - alpha) * np.sqrt(ey2)

    n_iter = _outer_loop(lambda it: drv.step(l1_reg, l2_reg, update_mask), global_error, max_iter, tol, verbose)
    ctx.sync()
    if stats is not None:
        stats["ex2"], stats["ey2"] = global_sq()
        if coll:
            stats["collective_calls"] = coll.stats()[0]
    for which, F in ((_lib.CMF_U, U_rows), (_lib.CMF_V, V), (_lib.CMF_Z, Z_rows)):
        ctx.get_factor_into(which, F)
    if coll:
        coll.barrier()
        coll.close()
    ctx.close()
    return U_rows, V, Z_rows, n_iter


def fit_newton_sharded(X_rows, X_cols, Y_cols, Y_rows, U_rows, V, Z_rows, alpha=0.5, l1_reg=0.0, l2_reg=0.0,
                       x_link="linear", y_link="linear", U_non_negative=True, V_non_negative=True, Z_non_negative=True,
                       hessian_pertubation=0.2, sg_sample_ratio=1.0, random_state=None, max_iter=200, tol=1e-4,
                       device=0, verbose=0, stats=None, rank=None, world=None, update_mask=7):
    """Row-sharded Newton fit for ANY link / sampling combination: call from every rank (one process per GPU).

    Rank g passes its row block of X and the SAME rows of U, the matching column block of Y with its rows of Z, and
    additionally its column block of X and row block of Y (rows ``block_bounds(d, world, rank)`` of V; all three
    partitions are ``block_bounds``, the layout the in-place all-gathers reassemble: the V sweep
    reads whole columns of X and rows of Y, pycmf/cmf_solvers.py:432-486); V is passed whole and identical on every
    rank.  Runs the reference's outer loop (:132-195) with the convergence test on the global error
    alpha ||X - f(UV^T)|| + (1 - alpha) ||Y - f(VZ^T)||.  ``sg_sample_ratio < 1`` uses the device sampler with the
    seed schedule of ``HipNewtonSolver(sg_sampler='device')``, so the iterates equal the single-GPU ones.
    U_rows, V, Z_rows are updated in place; returns (U_rows, V, Z_rows, n_iter).
    """
    from . import _lib
    from .comm import env_rank_world
    r_, w_ = env_rank_world()
    rank = r_ if rank is None else rank
    world = w_ if world is None else world
    m_g, d = X_rows.shape
    m = X_cols.shape[0]
    p_g, p = Y_cols.shape[1], Y_rows.shape[1]
    k = V.shape[1]
    r0, r1 = block_bounds(m, world, rank)
    q0, q1 = block_bounds(d, world, rank)
    c0, c1 = block_bounds(p, world, rank)
    if (m_g, p_g, X_cols.shape[1], Y_rows.shape[0]) != (r1 - r0, c1 - c0, q1 - q0, q1 - q0):
        raise ValueError("blocks do not match block_bounds for rank %d of %d" % (rank, world))
    nn_mask = (1 if U_non_negative else 0) | (2 if V_non_negative else 0) | (4 if Z_non_negative else 0)
    ctx_uz = _lib.Context(device)
    ctx_uz.set_problem(m_g, d, p_g, k)
    ctx_uz.set_data(0, X_rows)
    ctx_uz.set_data(1, Y_cols)
    ctx_uz.set_factor(_lib.CMF_U, U_rows); ctx_uz.set_factor(_lib.CMF_V, V); ctx_uz.set_factor(_lib.CMF_Z, Z_rows)
    ctx_v = _lib.Context(device, ctx_uz.stream_handle())     # both contexts launch on ONE stream
    ctx_v.set_problem(m, q1 - q0, p, k)
    ctx_v.set_data(0, X_cols)
    ctx_v.set_data(1, Y_rows)
    ctx_v.set_factor(_lib.CMF_V, V[q0:q1])
    coll, rank, world = _collectives_for(ctx_uz, rank, world)
    drv = make_sharded_newton_rows(ctx_uz, ctx_v, (r0, r1, q0, q1, c0, c1), (m, d, p), coll, alpha, x_link, y_link, nn_mask,
                                   hessian_pertubation, sg_sample_ratio)
    # the V-sweep context needs U and Z whole before its first sweep: the gathers of the first step provide them

    def global_sq():
        sq = np.array(ctx_uz.residual_sq(x_link, y_link))
        return tuple(coll.all_reduce_host(sq)) if coll else tuple(sq)

    def global_error():
        ex2, ey2 = global_sq()
        return alpha * np.sqrt(ex2) + (1 - alpha) * np.sqrt(ey2)

    seed0 = (int(random_state) if isinstance(random_state, (int, np.integer)) else 0) << 20
    n_iter = _outer_loop(lambda it: drv.step(l1_reg, l2_reg, update_mask, seed0 + it), global_error, max_iter, tol, verbose)
    ctx_uz.sync()
    if stats is not None:
        stats["ex2"], stats["ey2"] = global_sq()
    ctx_uz.get_factor_into(_lib.CMF_U, U_rows)
    ctx_uz.get_factor_into(_lib.CMF_V, V)
    ctx_uz.get_factor_into(_lib.CMF_Z, Z_rows)
    if coll:
        coll.barrier()
        coll.close()
    ctx_v.close()
    ctx_uz.close()
    return U_rows, V, Z_rows, n_iter
