"""CPU oracle for the CMF factor-update hot path  --  TEST INFRASTRUCTURE ONLY.

This module is a float64 NumPy restatement of the reference solver's
arithmetic (smn-ailab/PyCMF, pycmf/cmf_solvers.py).  It exists so that the HIP
kernels can be checked on a box where the reference itself is absent.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import it; nothing under ``pycmf_amd/`` does, and the product path raises
when the HIP library is missing instead of falling back to this file.

Pinning: the reference holds no golden vectors of its own, so
the oracle is pinned against outputs of the reference *run in the build
container*: ``tests/golden/make_golden.py`` imports /root/reference, runs its
``update_step`` / ``fit_transform`` on seeded inputs and stores inputs+outputs
in ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` replays them through
this file (agreement ~1e-12).

Every function cites the reference lines it restates (paths relative to
/root/reference).  The third-party arithmetic underneath the reference
(numpy.dot -> OpenBLAS dgemm, scipy.linalg.eigh -> LAPACK syevr/syevd,
scipy.special.expit, sklearn _beta_divergence) is un-vendored and unpinned
there (requirements.txt:1-7); fixtures were minted with numpy 2.2.6 /
scipy 1.15.3 / scikit-learn 1.7.2.
"""
import time

import numpy as np
import scipy.linalg
import scipy.sparse as sp
from scipy.special import expit

# pycmf/cmf_solvers.py:12 -- the "zero denominator" replacement is float32 eps
# even though all arithmetic is float64.
MU_EPS = float(np.finfo(np.float32).eps)


# --------------------------------------------------------------------------
# link helpers                                   pycmf/cmf_solvers.py:18-33
# --------------------------------------------------------------------------
def link_apply(t, link):
    """Inverse link f(.) applied to the linear predictor (cmf_solvers.py:27-33)."""
    if link == "linear":
        return t
    if link == "logit":
        return expit(t)
    raise ValueError("Invalid link function {}".format(link))


def link_slope(t):
    """sigma'(t) = s(1-s)                         (cmf_solvers.py:22-24)."""
    s = expit(t)
    return s * (1.0 - s)


def _dense(a):
    """ndarray view of whatever a sparse/np.matrix expression produced."""
    if sp.issparse(a):
        return a.toarray()
    return np.asarray(a)


def _matmul(a, b):
    """safe_sparse_dot equivalent (cmf_solvers.py:232,238,244): ``a @ b`` with
    a dense ndarray result."""
    return _dense(a @ b)


# --------------------------------------------------------------------------
# error metric                                   pycmf/cmf_solvers.py:36-42
# --------------------------------------------------------------------------
def factorization_error(target, left, right_t, link):
    """||target - f(left @ right_t)||_F.

    linear: sklearn ``_beta_divergence(target, left, right_t, 2,
    square_root=True)`` = sqrt(2 * 0.5*||T - LR||^2).  For sparse targets
    sklearn expands the square instead of forming L@R
    (``||T||^2 + tr((L^T L)(R R^T)) - 2 sum(T * LR)``); restated here so the
    sparse path carries the same rounding behaviour.
    logit: ``np.linalg.norm(target - sigmoid(left @ right_t))`` (dense only in
    the reference; a sparse target is densified by the subtraction).
    """
    if target is None:
        return 0
    if link == "linear":
        if sp.issparse(target):
            t2 = float(target.data @ target.data)
            # sum over nnz of T_ij * (L R)_ij
            coo = target.tocoo()
            lr_at_nnz = np.einsum("ij,ij->i", left[coo.row, :], right_t[:, coo.col].T)
            cross = float(coo.data @ lr_at_nnz)
            lr2 = float(np.sum((left.T @ left) * (right_t @ right_t.T)))
            res = (t2 + lr2 - 2.0 * cross) / 2.0
        else:
            diff = target - left @ right_t
            res = float(np.sum(diff * diff)) / 2.0
        return np.sqrt(2.0 * res)
    if link == "logit":
        est = expit(left @ right_t)
        return float(np.linalg.norm(_dense(target - est)))
    raise ValueError("Invalid link function {}".format(link))


def weighted_error(X, Y, U, V, Z, alpha, x_link, y_link):
    """alpha*e(X,U,V^T) + (1-alpha)*e(Y,V,Z^T)      (cmf_solvers.py:128-130)."""
    return alpha * factorization_error(X, U, V.T, x_link) + \
        (1 - alpha) * factorization_error(Y, V, Z.T, y_link)


# --------------------------------------------------------------------------
# MU solver                                     pycmf/cmf_solvers.py:198-263
# --------------------------------------------------------------------------
def mu_ratio(num, den, l1, l2, F):
    """num / reg(den) (cmf_solvers.py:212-228): ``den += l1`` if l1>0;
    ``den += l2*F`` if l2>0 (F = factor before its update); exact zeros of den
    become float32-eps; gamma is hard-wired to 1."""
    den = np.array(den, dtype=np.float64, copy=True)
    if l1 > 0:
        den += l1
    if l2 > 0:
        den = den + l2 * F
    den[den == 0] = MU_EPS
    return num / den


def mu_update_step(X, Y, U, V, Z, l1=0.0, l2=0.0,
                   update_U=True, update_V=True, update_Z=True):
    """One multiplicative sweep V -> U -> Z, in place (cmf_solvers.py:248-263).

    Association order is the reference's: the U and Z denominators form the
    full (U V^T) / (Z V^T) product before multiplying by V (:233, :239).  alpha
    does not appear anywhere in the MU path.
    """
    if update_V:
        num = _matmul(X.T, U) + _matmul(Y, Z)                 # :244
        den = V @ (U.T @ U + Z.T @ Z)                         # :245
        V *= mu_ratio(num, den, l1, l2, V)                    # :253-255
    if update_U:
        num = _matmul(X, V)                                   # :232
        den = (U @ V.T) @ V                                   # :233
        U *= mu_ratio(num, den, l1, l2, U)                    # :257-259
    if update_Z:
        num = _matmul(Y.T, V)                                 # :238
        den = (Z @ V.T) @ V                                   # :239
        Z *= mu_ratio(num, den, l1, l2, Z)                    # :261-263


# --------------------------------------------------------------------------
# Newton solver (live pure-Python path)         pycmf/cmf_solvers.py:318-522
# --------------------------------------------------------------------------
_BLAS_CONTROLLER = None


def _one_blas_thread():
    """Context in which BLAS / LAPACK run on ONE thread: the k x k ``eigh`` of ``safe_invert`` (k <= a few hundred) is an order of
    magnitude SLOWER on a many-threaded OpenBLAS (measured on the 256-CPU test host: 10-20 s of a test were three 200 x 200
    eigensolves).  Same algorithm, same result to rounding; a no-op without threadpoolctl."""
    global _BLAS_CONTROLLER
    if _BLAS_CONTROLLER is None:
        try:
            from threadpoolctl import ThreadpoolController
            _BLAS_CONTROLLER = ThreadpoolController()
        except Exception:
            _BLAS_CONTROLLER = False
    if not _BLAS_CONTROLLER:
        import contextlib
        return contextlib.nullcontext()
    return _BLAS_CONTROLLER.limit(limits=1, user_api="blas")


def safe_invert(H, pert):
    """Q diag(1/max(|lam|, pert)) Q^T            (cmf_solvers.py:346-356)."""
    with _one_blas_thread():
        lam, Q = scipy.linalg.eigh(H)
        lam = np.abs(lam)
        lam[lam < pert] = pert
        return (Q @ np.diag(1.0 / lam)) @ Q.T


def draw_sample(n, ratio):
    """Index sample of the reference's ``_stochastic_sample``
    (cmf_solvers.py:328-344): None when ratio == 1 (no RNG call at all),
    else the first int(n*ratio) entries of ``np.random.permutation(arange(n))``
    drawn from the *global* legacy RNG."""
    if ratio < 1.0:
        size = int(n * ratio)
        return np.random.permutation(np.arange(n))[:size]
    return None


def _row_step(F, i, grad, Hinv, non_negative):
    """F[i] <- F[i] - grad @ Hinv, then clamp      (cmf_solvers.py:321-326)."""
    F[i, :] = F[i, :] - 1.0 * (grad @ Hinv)
    if non_negative:
        row = F[i, :]
        row[row < 0] = 0.0


def _target_row(T, i, cols):
    """Dense 1-D copy of T[i, cols] for ndarray or sparse T."""
    if sp.issparse(T):
        r = T[i, :].toarray().ravel()
        return r if cols is None else r[cols]
    return T[i, :] if cols is None else T[i, cols]


def _target_col(T, rows, j):
    """Dense 1-D copy of T[rows, j] for ndarray or sparse T."""
    if sp.issparse(T):
        c = T[:, j].toarray().ravel()
        return c if rows is None else c[rows]
    return T[:, j] if rows is None else T[rows, j]


def newton_sweep_U(U, V, X, alpha, l1, l2, link, non_negative, ratio, pert,
                   masks=None):
    """Row-wise Newton sweep over U            (cmf_solvers.py:394-430).

    ``masks`` (optional list) records the sample drawn for every row, in draw
    order, so that the HIP path can be replayed with identical samples.
    """
    k = U.shape[1]
    full = (ratio == 1.0)
    if full:
        R = _dense(link_apply(U @ V.T, link) - X)                         # :399
        G = alpha * (R @ V) + l1 * np.sign(U) + l2 * U                    # :400
    shared = (link == "linear" and full)
    if shared:
        Hinv = safe_invert(alpha * (V.T @ V) + l2 * np.eye(k), pert)      # :410
    for i in range(U.shape[0]):
        u = U[i, :]
        s = draw_sample(V.shape[0], ratio)                                # :414
        if masks is not None:
            masks.append(s)
        Vs = V if s is None else V[s, :]
        if full:
            g = G[i, :]
        else:
            r = link_apply(u @ Vs.T, link) - _target_row(X, i, s)         # :419
            g = alpha * (r @ Vs) + l1 * np.sign(u) + l2 * u               # :420
        if not shared:
            if link == "linear":
                Hinv = safe_invert(alpha * (Vs.T @ Vs) + l2 * np.eye(k), pert)   # :424
            else:
                w = link_slope(u @ Vs.T)
                # logit Hessian of U carries NO l2 term (:427-428)
                Hinv = safe_invert(alpha * ((Vs.T * w) @ Vs), pert)
        _row_step(U, i, g, Hinv, non_negative)


def newton_sweep_Z(Z, V, Y, alpha, l1, l2, link, non_negative, ratio, pert,
                   masks=None, cython_variant=False):
    """Row-wise Newton sweep over Z            (cmf_solvers.py:488-508).

    Never uses a precomputed gradient or shared inverse; weight is (1-alpha);
    both link branches add l2*I (:501-506).

    ``cython_variant``: the (dead) Cython path updates Z with the same routine as U,
    ``_newton_update_left(Z, V, Y^T, 1-alpha, ...)`` (cmf_solvers.py:300-304,
    cmf_newton_solver.pyx:240-292), whose logit Hessian carries NO l2 term
    (pyx:287-290); everything else is identical in value.
    """
    k = Z.shape[1]
    for i in range(Z.shape[0]):
        z = Z[i, :]
        s = draw_sample(V.shape[0], ratio)                                # :494
        if masks is not None:
            masks.append(s)
        Vs = V if s is None else V[s, :]
        r = link_apply(Vs @ z, link) - _target_col(Y, s, i)               # :495
        g = (1 - alpha) * (r @ Vs) + l1 * np.sign(z) + l2 * z             # :497
        if link == "linear":
            H = (1 - alpha) * (Vs.T @ Vs) + l2 * np.eye(k)                # :501
        else:
            w = link_slope(Vs @ z)
            H = (1 - alpha) * ((Vs.T * w) @ Vs)                           # :505 / pyx:289
            if not cython_variant:
                H = H + l2 * np.eye(k)
        _row_step(Z, i, g, safe_invert(H, pert), non_negative)


def newton_sweep_V(V, U, Z, X, Y, alpha, l1, l2, x_link, y_link, non_negative,
                   ratio, pert, masks=None):
    """Row-wise Newton sweep over the shared factor V (cmf_solvers.py:432-486).

    Per row two independent samples are drawn, first over the rows of U/X then
    over the rows of Z (= columns of Y) (:455-456).
    """
    k = V.shape[1]
    full = (ratio == 1.0)
    if full:
        RX = _dense(link_apply(U @ V.T, x_link) - X)                      # :436
        RY = _dense(link_apply(Z @ V.T, y_link) - Y.T)                    # :437
        G = alpha * (RX.T @ U) + (1 - alpha) * (RY.T @ Z) + \
            l1 * np.sign(V) + l2 * V                                      # :438
    shared = (x_link == "linear" and y_link == "linear" and full)
    if shared:
        Hinv = safe_invert(alpha * (U.T @ U) + (1 - alpha) * (Z.T @ Z) +
                           l2 * np.eye(k), pert)                          # :448
    for i in range(V.shape[0]):
        v = V[i, :]
        su = draw_sample(U.shape[0], ratio)                               # :455
        sz = draw_sample(Z.shape[0], ratio)                               # :456
        if masks is not None:
            masks.append((su, sz))
        Us = U if su is None else U[su, :]
        Zs = Z if sz is None else Z[sz, :]
        if full:
            g = G[i, :]
        else:
            rx = link_apply(Us @ v, x_link) - _target_col(X, su, i)       # :459
            ry = link_apply(v @ Zs.T, y_link) - _target_row(Y, i, sz)     # :460
            g = alpha * (rx @ Us) + (1 - alpha) * (ry @ Zs) + \
                l1 * np.sign(v) + l2 * v                                  # :461
        if not shared:
            if x_link == "logit":
                HU = (Us.T * link_slope(Us @ v)) @ Us                     # :468
            else:
                HU = Us.T @ Us                                            # :471
            if y_link == "logit":
                HZ = (Zs.T * link_slope(v @ Zs.T)) @ Zs                   # :476
            else:
                HZ = Zs.T @ Zs                                            # :479
            Hinv = safe_invert(alpha * HU + (1 - alpha) * HZ + l2 * np.eye(k), pert)
        _row_step(V, i, g, Hinv, non_negative)


def newton_update_step(X, Y, U, V, Z, alpha, l1=0.0, l2=0.0,
                       x_link="linear", y_link="linear",
                       U_non_negative=True, V_non_negative=True, Z_non_negative=True,
                       ratio=1.0, pert=0.2,
                       update_U=True, update_V=True, update_Z=True, masks=None, cython_variant=False):
    """One Newton sweep U -> Z -> V, in place     (cmf_solvers.py:510-522).

    ``masks``: optional dict {"U": [], "Z": [], "V": []} that receives the
    drawn samples.
    """
    mU = masks["U"] if masks is not None else None
    mZ = masks["Z"] if masks is not None else None
    mV = masks["V"] if masks is not None else None
    if update_U:
        newton_sweep_U(U, V, X, alpha, l1, l2, x_link, U_non_negative, ratio, pert, mU)
    if update_Z:
        newton_sweep_Z(Z, V, Y, alpha, l1, l2, y_link, Z_non_negative, ratio, pert, mZ,
                       cython_variant=cython_variant)
    if update_V:
        newton_sweep_V(V, U, Z, X, Y, alpha, l1, l2, x_link, y_link,
                       V_non_negative, ratio, pert, mV)


# --------------------------------------------------------------------------
# solver shell                                  pycmf/cmf_solvers.py:45-195
# --------------------------------------------------------------------------
class OracleSolver:
    """Shell around the two update steps with the reference's outer loop:
    error at init, one ``update_step`` per iteration, convergence test every
    10th iteration when tol > 0 (cmf_solvers.py:132-195).  The constructor
    seeds the global RNG like the reference (:121-122)."""

    def __init__(self, solver="mu", max_iter=200, tol=1e-4, l1_reg=0, l2_reg=0,
                 alpha=0.5, verbose=0,
                 U_non_negative=True, V_non_negative=True, Z_non_negative=True,
                 update_U=True, update_V=True, update_Z=True,
                 x_link="linear", y_link="linear", hessian_pertubation=0.2,
                 sg_sample_ratio=1., random_state=None):
        self.solver = solver
        self.max_iter = max_iter
        self.tol = tol
        self.l1_reg = l1_reg
        self.l2_reg = l2_reg
        self.alpha = alpha
        self.verbose = verbose
        self.nn = (U_non_negative, V_non_negative, Z_non_negative)
        self.upd = (update_U, update_V, update_Z)
        self.x_link = x_link
        self.y_link = y_link
        self.pert = hessian_pertubation
        self.ratio = sg_sample_ratio
        if random_state is not None:
            np.random.seed(random_state)

    def compute_error(self, X, Y, U, V, Z):
        return weighted_error(X, Y, U, V, Z, self.alpha, self.x_link, self.y_link)

    def update_step(self, X, Y, U, V, Z):
        if self.solver == "mu":
            mu_update_step(X, Y, U, V, Z, self.l1_reg, self.l2_reg,
                           update_U=self.upd[0], update_V=self.upd[1], update_Z=self.upd[2])
        else:
            newton_update_step(X, Y, U, V, Z, self.alpha, self.l1_reg, self.l2_reg,
                               self.x_link, self.y_link, *self.nn,
                               ratio=self.ratio, pert=self.pert,
                               update_U=self.upd[0], update_V=self.upd[1],
                               update_Z=self.upd[2])

    def fit_iterative_update(self, X, Y, U, V, Z):
        t0 = time.time()
        prev = err0 = self.compute_error(X, Y, U, V, Z)                   # :168
        n_iter = 0
        for n_iter in range(1, self.max_iter + 1):                        # :170
            self.update_step(X, Y, U, V, Z)
            if self.tol > 0 and n_iter % 10 == 0:                         # :175
                err = self.compute_error(X, Y, U, V, Z)
                if self.verbose:
                    print("Epoch %02d reached after %.3f seconds, error: %f" %
                          (n_iter, time.time() - t0, err))
                if (prev - err) / err0 < self.tol:                        # :183
                    break
                prev = err
        if self.verbose and (self.tol == 0 or n_iter % 10 != 0):
            print("Epoch %02d reached after %.3f seconds." % (n_iter, time.time() - t0))
        return U, V, Z, n_iter
