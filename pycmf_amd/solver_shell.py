"""Host-side solver objects: the drop-in seam of the reference.

The reference's driver builds a solver object and calls exactly one method on it,
``fit_iterative_update(X, Y, U, V, Z) -> (U, V, Z, n_iter)``
(pycmf/cmf.py:437-454).  ``HipMUSolver`` and ``HipNewtonSolver`` accept the same
constructor keywords as the reference classes (pycmf/cmf_solvers.py:98-103) and
expose the same three methods (``fit_iterative_update`` :132, ``update_step``
:124, ``compute_error`` :128), but every flop of the update runs in
libcmfhip.so on the GPU:  X and Y are uploaded once, the factors stay resident
for the whole loop, and one scalar comes back per convergence check.

Semantics kept from the reference:
* U, V, Z are mutated in place and also returned (:195, :255, :324).
* the convergence test runs every 10th iteration when tol > 0 (:175-187) with
  the same verbose print format (:178-181, :190-193).
* constructing a solver with ``random_state`` seeds NumPy's *global* RNG
  (:121-122); stochastic Newton draws its per-row samples from that stream in
  the reference's order so that results are reproducible against it.
* MU ignores alpha (its error metric uses the constructor default 0.5, :99).
"""
import os
import time

import numpy as np

from . import _lib


def _as_f64(a):
    return a if (isinstance(a, np.ndarray) and a.dtype == np.float64) else np.asarray(a, dtype=np.float64)


class _HipIterativeSolver:
    def __init__(self, max_iter=200, tol=1e-4, beta_loss="frobenius",
                 l1_reg=0, l2_reg=0, alpha=0.5, verbose=0,
                 U_non_negative=True, V_non_negative=True, Z_non_negative=True,
                 update_U=True, update_V=True, update_Z=True,
                 x_link="linear", y_link="linear", hessian_pertubation=0.2,
                 sg_sample_ratio=1., random_state=None, device=0, stream=None, sg_sampler="numpy",
                 cython_variant=False):
        # like the reference, any beta_loss sklearn can parse is accepted and then ignored: only the
        # Frobenius objective is implemented by either solver (cmf_solvers.py:106, :166)
        if isinstance(beta_loss, str) and beta_loss not in ("frobenius", "kullback-leibler", "itakura-saito"):
            raise ValueError("Invalid beta_loss parameter: got %r" % (beta_loss,))
        self.max_iter = max_iter
        self.tol = tol
        self.beta_loss = 2.0
        self.l1_reg = l1_reg
        self.l2_reg = l2_reg
        self.alpha = alpha
        self.verbose = verbose
        self.U_non_negative = U_non_negative
        self.V_non_negative = V_non_negative
        self.Z_non_negative = Z_non_negative
        self.update_U = update_U
        self.update_V = update_V
        self.update_Z = update_Z
        self.x_link = x_link
        self.y_link = y_link
        self.hessian_pertubation = hessian_pertubation
        self.sg_sample_ratio = sg_sample_ratio
        self.device = device
        self.stream = stream
        if sg_sampler not in ("numpy", "device"):
            raise ValueError("sg_sampler must be 'numpy' (reference RNG stream) or 'device', got %r" % (sg_sampler,))
        self.sg_sampler = sg_sampler
        # True: numerics of the reference's (unused) Cython twin, whose Z sweep shares U's routine and
        # therefore has no l2 term in its logit Hessian (cmf_newton_solver.pyx:287-290; SURVEY N-cy1)
        self.cython_variant = bool(cython_variant)
        self._sample_seed = (int(random_state) if isinstance(random_state, (int, np.integer)) else 0) << 20
        self._ctx = None
        self._bound = None
        if random_state is not None:
            np.random.seed(random_state)

    # ------------------------------------------------------------------ device state
    def _update_mask(self):
        return (_lib.UPD_U if self.update_U else 0) | (_lib.UPD_V if self.update_V else 0) | \
            (_lib.UPD_Z if self.update_Z else 0)

    def _nn_mask(self):
        return (1 if self.U_non_negative else 0) | (2 if self.V_non_negative else 0) | \
            (4 if self.Z_non_negative else 0)

    def bind_data(self, X, Y, k):
        """Upload X and Y ahead of the factors (the initialisers use the device copies)."""
        m = X.shape[0] if X is not None else None
        d = X.shape[1] if X is not None else Y.shape[0]
        p = Y.shape[1] if Y is not None else None
        if m is None or p is None:
            return None
        return self._bind_dims(X, Y, m, d, p, k)

    def _bind(self, X, Y, U, V, Z):
        """Upload X, Y (once per distinct pair) and size the device problem."""
        m, k = U.shape
        return self._bind_dims(X, Y, m, V.shape[0], Z.shape[0], k)

    def rebind(self):
        """Forget the uploaded X / Y: the next call uploads them again.  The device copies are cached by the
        IDENTITY of the arrays (``id(X), id(Y)`` and the shapes), which is what lets a loop of ``update_step`` /
        ``compute_error`` calls on the same matrices (tests/test_cmf.py:126-162 upstream) pay one upload; a caller
        who edits X or Y IN PLACE between calls must call this (the reference re-reads X on every call)."""
        self._bound = None

    def _bind_dims(self, X, Y, m, d, p, k):
        key = (id(X), id(Y), m, d, p, k)
        if self._ctx is None:
            self._ctx = _lib.Context(self.device, self.stream)
            mode = os.environ.get("PYCMF_AMD_SPARSE_MODE")  # "dense" | "native": override the auto choice
            if mode:
                self._ctx.set_option("sparse_mode", {"auto": 0, "dense": 1, "native": 2}[mode])
            if self.cython_variant:
                self._ctx.set_option("z_logit_hessian_l2", 0)
            if os.environ.get("PYCMF_AMD_GEMM_ARITH") == "bf16x6":  # opt-in arithmetic of the k_pad = 256 data passes
                self._ctx.set_option("gemm_arith", 1)
            if os.environ.get("PYCMF_AMD_REFINE_ROWS") == "0":  # A/B: leave ill-conditioned clamped rows in float32 (recorded, warned)
                self._ctx.set_option("refine_rows", 0)
        if self._bound != key:
            self._ctx.set_problem(m, d, p, k)
            if X is not None:
                if X.shape != (m, d):
                    raise ValueError("X has shape %s, factors imply %s" % (X.shape, (m, d)))
                self._ctx.set_data(0, X)
            if Y is not None:
                if Y.shape != (d, p):
                    raise ValueError("Y has shape %s, factors imply %s" % (Y.shape, (d, p)))
                self._ctx.set_data(1, Y)
            self._bound = key
            self._XY = (X, Y)  # keep ids alive
        return self._ctx

    def _push_factors(self, U, V, Z):
        self._ctx.set_factor(_lib.CMF_U, U)
        self._ctx.set_factor(_lib.CMF_V, V)
        self._ctx.set_factor(_lib.CMF_Z, Z)

    def _pull_factors(self, U, V, Z):
        for which, F in ((_lib.CMF_U, U), (_lib.CMF_V, V), (_lib.CMF_Z, Z)):
            if isinstance(F, np.ndarray) and F.dtype == np.float64 and F.flags.writeable:
                self._ctx.get_factor_into(which, F)
            else:
                F[...] = self._ctx.get_factor(which)

    def release(self):
        if self._ctx is not None:
            self._ctx.close()
        self._ctx = None
        self._bound = None

    # ------------------------------------------------------------------ reference API
    def _device_step(self, l1_reg, l2_reg, alpha):
        raise NotImplementedError("Implement in concrete subclass to use")

    def update_step(self, X, Y, U, V, Z, l1_reg, l2_reg, alpha):
        """One sweep over all factors, in place on U, V, Z (cmf_solvers.py:124)."""
        self._bind(X, Y, U, V, Z)
        self._push_factors(U, V, Z)
        self._device_step(l1_reg, l2_reg, alpha)
        self._pull_factors(U, V, Z)

    def _device_step_error(self, l1_reg, l2_reg, alpha):
        """One update step AND (||X - f(UV^T)||_F, ||Y - f(VZ^T)||_F) of its result in one device call, or None where the solver has
        no such call (then the caller runs the step and the error pass separately)."""
        return None

    def _device_error(self):
        ex2, ey2 = self._ctx.residual_sq(self.x_link, self.y_link)
        X, Y = self._XY
        ex = np.sqrt(ex2) if X is not None else 0.0
        ey = np.sqrt(ey2) if Y is not None else 0.0
        return ex, ey

    def compute_error(self, X, Y, U, V, Z):
        """alpha*||X - f(UV^T)||_F + (1-alpha)*||Y - f(VZ^T)||_F (cmf_solvers.py:128-130)."""
        self._bind(X, Y, U, V, Z)
        self._push_factors(U, V, Z)
        ex, ey = self._device_error()
        return self.alpha * ex + (1 - self.alpha) * ey

    def reconstruction_error(self):
        """||X - f(UV^T)||_F + ||Y - f(VZ^T)||_F with the factors currently on the
        device (pycmf/cmf.py:697-698)."""
        ex, ey = self._device_error()
        return ex + ey

    def _run_params(self):
        """Keyword arguments of Context.run for this solver, or None when the loop has to stay on the host (index lists drawn from
        NumPy's stream)."""
        raise NotImplementedError("Implement in concrete subclass to use")

    def fit_iterative_update(self, X, Y, U, V, Z):
        """Alternating minimisation loop (cmf_solvers.py:132-195).  The loop itself runs inside libcmfhip (``cmf_run``: error at
        init, one step per iteration, the check every 10th iteration, early stop) unless the per-row samples come from NumPy's
        global stream, which only the host can draw from; the verbose lines of the reference are printed from the trace the
        C loop returns (same text, same elapsed times, after the loop instead of during it)."""
        start_time = time.time()
        self._bind(X, Y, U, V, Z)
        self._push_factors(U, V, Z)
        self._fit_begin()
        params = self._run_params()
        if params is not None and os.environ.get("PYCMF_AMD_HOST_LOOP") != "1":
            before_run = time.time() - start_time      # upload and binding: part of the reference's clock (it starts at :144)
            n_iter, errs, secs = self._ctx.run(max_iter=self.max_iter, tol=self.tol, **params)
            self._after_run(n_iter)
            if self.verbose:
                every = int(params.get("check_every", 10))
                for i, (e, t) in enumerate(zip(errs[1:], secs[1:]), start=1):
                    print("Epoch %02d reached after %.3f seconds, error: %f" % (every * i, before_run + t, e))
                if self.tol == 0 or n_iter % every != 0:
                    print("Epoch %02d reached after %.3f seconds." % (n_iter, time.time() - start_time))
            self._fit_end()
            self._pull_factors(U, V, Z)
            return U, V, Z, n_iter
        ex, ey = self._device_error()
        previous_error = error_at_init = self.alpha * ex + (1 - self.alpha) * ey

        n_iter = 0
        for n_iter in range(1, self.max_iter + 1):
            check = self.tol > 0 and n_iter % 10 == 0
            # (MU: the check iteration is ONE call -- the step and the error of its result from the step's own products,
            # cmf_mu_step_error -- exactly what the C loop runs, so the two loops stay launch for launch the same)
            fused = self._device_step_error(self.l1_reg, self.l2_reg, self.alpha) if check else None
            if fused is None:
                self._device_step(self.l1_reg, self.l2_reg, self.alpha)
            if check:
                ex, ey = fused if fused is not None else self._device_error()
                error = self.alpha * ex + (1 - self.alpha) * ey
                if self.verbose:
                    print("Epoch %02d reached after %.3f seconds, error: %f" %
                          (n_iter, time.time() - start_time, error))
                if (previous_error - error) / error_at_init < self.tol:
                    break
                previous_error = error

        if self.verbose and (self.tol == 0 or n_iter % 10 != 0):
            self._ctx.sync()
            print("Epoch %02d reached after %.3f seconds." % (n_iter, time.time() - start_time))

        self._fit_end()
        self._pull_factors(U, V, Z)
        return U, V, Z, n_iter

    def _after_run(self, n_iter):
        pass

    def _fit_begin(self):
        pass

    def _fit_end(self):
        pass


class HipMUSolver(_HipIterativeSolver):
    """Multiplicative updates V -> U -> Z (pycmf/cmf_solvers.py:198-263) on the GPU."""

    def _device_step(self, l1_reg, l2_reg, alpha):
        self._ctx.mu_step(l1_reg, l2_reg, self._update_mask())

    def _device_step_error(self, l1_reg, l2_reg, alpha):
        ex2, ey2 = self._ctx.mu_step_error(l1_reg, l2_reg, self._update_mask())
        X, Y = self._XY
        return (np.sqrt(ex2) if X is not None else 0.0), (np.sqrt(ey2) if Y is not None else 0.0)

    def _run_params(self):
        return dict(solver="mu", l1=self.l1_reg, l2=self.l2_reg, alpha_err=self.alpha, update_mask=self._update_mask())


class HipNewtonSolver(_HipIterativeSolver):
    """Row-wise Newton-Raphson sweeps U -> Z -> V (pycmf/cmf_solvers.py:318-522) on the GPU.

    With ``sg_sample_ratio < 1`` the per-row samples are drawn on the host from
    NumPy's global RNG in the reference's order (:328-344: U rows, Z rows, then for
    every V row a U-sample followed by a Z-sample) and handed to the device as
    index lists ("parity mode", ``sg_sampler='numpy'``, the default).  With
    ``sg_sampler='device'`` the same distribution (exactly int(n*ratio) distinct
    candidates per row) is drawn on the GPU from a counter-based generator: no
    host RNG time, no index upload, but not NumPy's stream.
    """

    #: ||H||_F / hessian_pertubation above which the float32 spectral clamp of a row's Hessian leaves the stated tolerance
    #: (tests/tools/fuzz_campaign.py: every case within 3e-3 of the float64 reference below it)
    CLAMP_RATIO_WARN = 1.0e4

    #: index entries per iteration above which drawing the per-row samples from NumPy's stream on the host dominates the iteration
    #: (one np.random.permutation per row plus the upload of the lists: ~16 s per iteration at BASELINE config C3, against 0.24 s
    #: for the whole iteration with sg_sampler='device')
    HOST_SAMPLER_WARN_ENTRIES = 1.0e7

    def _fit_begin(self):
        self._ctx.newton_clamp_stats(reset=True)
        if self.sg_sample_ratio < 1. and self.sg_sampler == "numpy":
            m, d, p, _ = self._ctx.shape
            r = self.sg_sample_ratio
            entries = ((m + p) * int(d * r) if (self.update_U or self.update_Z) else 0) + (d * (int(m * r) + int(p * r)) if self.update_V else 0)
            if entries > self.HOST_SAMPLER_WARN_ENTRIES:
                import warnings
                warnings.warn("pycmf_amd: sg_sampler='numpy' draws every row's sample from NumPy's global stream on the host, like the "
                              "reference (pycmf/cmf_solvers.py:328-344): %.1e list entries per iteration here -- the host RNG and the "
                              "upload of the lists will dominate the iteration.  sg_sampler='device' draws the same distribution on "
                              "the GPU (the benchmarked path; not NumPy's stream)." % entries, RuntimeWarning, stacklevel=3)

    def _fit_end(self):
        self.clamped_rows_, self.clamp_ratio_, self.refined_rows_ = self._ctx.newton_clamp_stats()
        if self.clamp_ratio_ > self.CLAMP_RATIO_WARN:
            import warnings
            warnings.warn("pycmf_amd: %d row Hessians had eigenvalues below hessian_pertubation=%g while ||H||_F / pertubation reached "
                          "%.1e and were NOT redone in float64 (more than refine_rows_max rows in one sweep, n_components > 256, or "
                          "refinement switched off): float32 Hessians resolve the clamped directions only to about 1e-7 * that ratio, "
                          "so the factors may differ from the float64 reference by more than the stated tolerance.  A positive l2_reg "
                          "at least as large as the perturbation, or fewer components than samples per row, keeps the Hessians well "
                          "conditioned." % (self.clamped_rows_, self.hessian_pertubation, self.clamp_ratio_),
                          RuntimeWarning, stacklevel=3)

    def _run_params(self):
        if self.sg_sample_ratio < 1. and self.sg_sampler != "device":
            return None                 # NumPy's stream: the lists are drawn on the host, iteration by iteration
        return dict(solver="newton", l1=self.l1_reg, l2=self.l2_reg, alpha=self.alpha, alpha_err=self.alpha, x_link=self.x_link,
                    y_link=self.y_link, nn_mask=self._nn_mask(), update_mask=self._update_mask(), pert=self.hessian_pertubation,
                    ratio=min(float(self.sg_sample_ratio), 1.0), seed=self._sample_seed)

    def _after_run(self, n_iter):
        if self.sg_sample_ratio < 1.:
            self._sample_seed += n_iter     # the C loop drew iteration i under seed + i, like _device_step would have

    def _draw(self, rows, n, ratio):
        size = int(n * ratio)
        out = np.empty((rows, size), dtype=np.int32)
        ar = np.arange(n)
        for i in range(rows):
            out[i] = np.random.permutation(ar)[:size]
        return out

    def _device_step(self, l1_reg, l2_reg, alpha):
        m, d, p, _ = self._ctx.shape
        ratio = self.sg_sample_ratio
        u_idx = z_idx = vx_idx = vy_idx = None
        if ratio < 1. and self.sg_sampler == "device":
            self._sample_seed += 1
            self._ctx.newton_step_device_sampled(alpha, l1_reg, l2_reg, self.x_link, self.y_link, self._nn_mask(),
                                                 self._update_mask(), self.hessian_pertubation, ratio,
                                                 self._sample_seed)
            return
        if ratio < 1.:
            if self.update_U:
                u_idx = self._draw(m, d, ratio)
            if self.update_Z:
                z_idx = self._draw(p, d, ratio)
            if self.update_V:
                sm, sp_ = int(m * ratio), int(p * ratio)
                vx_idx = np.empty((d, sm), dtype=np.int32)
                vy_idx = np.empty((d, sp_), dtype=np.int32)
                am, ap = np.arange(m), np.arange(p)
                for i in range(d):
                    vx_idx[i] = np.random.permutation(am)[:sm]
                    vy_idx[i] = np.random.permutation(ap)[:sp_]
        self._ctx.newton_step(alpha, l1_reg, l2_reg, self.x_link, self.y_link,
                              self._nn_mask(), self._update_mask(),
                              self.hessian_pertubation, ratio, u_idx, z_idx, vx_idx, vy_idx)
