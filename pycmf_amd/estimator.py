"""scikit-learn style front end: ``CMF`` and ``collective_matrix_factorization``.

API-compatible counterpart of pycmf/cmf.py:215-776 (same keywords, defaults,
attributes, validation messages and init routing); the solve itself is delegated
to the HIP solver objects in :mod:`pycmf_amd.solver_shell`.
"""
import numbers
import warnings

import numpy as np
from sklearn.base import BaseEstimator, TransformerMixin
from sklearn.utils import check_array

from .factor_init import initialize_mf, init_custom, DeviceOperand, DEVICE_SVD_MIN_CELLS
from .solver_shell import HipMUSolver, HipNewtonSolver
from .topic_terms import print_topic_terms_from_matrix, print_topic_terms_with_importances

_BETA_NAMES = {'frobenius': 2, 'kullback-leibler': 1, 'itakura-saito': 0}


def _check_beta_loss(beta_loss):
    # sklearn's _beta_loss_to_float contract (used at pycmf/cmf_solvers.py:106); only
    # the Frobenius objective is implemented by either solver (:166)
    if isinstance(beta_loss, str):
        if beta_loss not in _BETA_NAMES:
            raise ValueError('Invalid beta_loss parameter: got %r instead of one of %r, or a float.'
                             % (beta_loss, list(_BETA_NAMES)))
    elif not isinstance(beta_loss, numbers.Number):
        raise ValueError('Invalid beta_loss parameter: got %r instead of one of %r, or a float.'
                         % (beta_loss, list(_BETA_NAMES)))


def collective_matrix_factorization(X, Y, U=None, V=None, Z=None,
                                    x_init=None, y_init=None, n_components=None,
                                    solver='mu', alpha=0.5, beta_loss='frobenius',
                                    tol=1e-4, max_iter=200, l1_reg=0., l2_reg=0.,
                                    random_state=None, verbose=0,
                                    U_non_negative=True, V_non_negative=True, Z_non_negative=True,
                                    update_U=True, update_V=True, update_Z=True,
                                    x_link="linear", y_link="linear",
                                    hessian_pertubation=0.2, sg_sample_ratio=1.,
                                    device=0, sg_sampler="numpy", n_gpus=1, _return_solver=False):
    """Factorise X ~ f(U V^T) and Y ~ f(V Z^T) with a shared V on an MI355X.

    Same contract as the reference function (pycmf/cmf.py:215-456): returns
    ``(U, V, Z, n_iter)``; with ``x_init='custom'`` / ``y_init='custom'`` the given
    U/V/Z are the starting point (and U, Z are updated in place).

    ``n_gpus > 1``: data-parallel fit on that many GPUs of the node: one worker process per GPU takes a
    row block of X / U and a column block of Y / Z, V is reassembled by RCCL once per iteration
    (pycmf_amd/multi_gpu.py).  This process touches no GPU; the initial factors come from the host initialisers.  With
    ``sg_sample_ratio < 1`` the workers draw their samples with the device sampler.
    """
    if n_components is None:
        n_components = max(X.shape[1], Y.shape[1])
    _check_beta_loss(beta_loss)

    if update_U or update_V:
        X = check_array(X, accept_sparse=('csr', 'csc'), dtype=float)
    if update_Z or update_V:
        Y = check_array(Y, accept_sparse=('csr', 'csc'), dtype=float)

    if update_V and X.shape[1] != Y.shape[0]:
        raise ValueError("Expected X.shape[1] == Y.shape[0], " +
                         "found X.shape = {}, Y.shape = {}".format(X.shape[1], Y.shape[0]))
    if x_link not in ("linear", "logit"):
        raise ValueError("No such link %s for x_link" % x_link)
    if y_link not in ("linear", "logit"):
        raise ValueError("No such link %s for y_link" % y_link)

    # ---- solver dispatch (cmf.py:433-453)
    common = dict(max_iter=max_iter, tol=tol, verbose=verbose, update_U=update_U, update_V=update_V,
                  update_Z=update_Z, l1_reg=l1_reg, l2_reg=l2_reg, random_state=random_state, device=device)
    if solver == "mu":
        if x_link != "linear" or y_link != "linear":
            warnings.warn("mu solver does not accept link functions other than linear, "
                          "link arguments will be ignored")
        solver_object = HipMUSolver(beta_loss=beta_loss, **common)
    elif solver == "newton":
        if alpha == "auto":
            alpha = Y.shape[1] / (X.shape[0] + Y.shape[1])
        solver_object = HipNewtonSolver(alpha=alpha, U_non_negative=U_non_negative,
                                        V_non_negative=V_non_negative, Z_non_negative=Z_non_negative,
                                        x_link=x_link, y_link=y_link,
                                        hessian_pertubation=hessian_pertubation,
                                        sg_sample_ratio=sg_sample_ratio, sg_sampler=sg_sampler, **common)
    else:
        raise ValueError("No such solver: %s" % solver)

    if n_gpus > 1 and (X is None or Y is None):
        n_gpus = 1     # a transform of one side only (cmf.py:726-747 with X or Y None): nothing to shard against, one GPU does it

    # Large inputs go to the GPU before the initialisers run, so that the randomized SVD behind
    # 'svd' / 'nndsvd*' can use the device copy for its products (the solver later reuses the upload).
    op_x = op_y = None
    big = [M is not None and M.shape[0] * M.shape[1] >= DEVICE_SVD_MIN_CELLS for M in (X, Y)]
    needs_dev = [i != 'custom' for i in (x_init, y_init)]   # every non-custom rule reads the data: mean and / or SVD
    if n_gpus <= 1 and X is not None and Y is not None and any(b and n for b, n in zip(big, needs_dev)):
        ctx = solver_object.bind_data(X, Y, n_components)
        if ctx is not None:
            op_x = DeviceOperand(ctx, 0, X.shape)
            op_y = DeviceOperand(ctx, 1, Y.shape)

    # ---- initial factors (cmf.py:402-430)
    init_spec = dict(x_init=x_init, y_init=y_init, n_components=n_components, random_state=random_state, x_link=x_link,
                     y_link=y_link, U_non_negative=U_non_negative, V_non_negative=V_non_negative, Z_non_negative=Z_non_negative)
    # n_gpus > 1 and inputs large enough for the device-side initialisers: rank 0 of the workers computes them on ITS GPU
    # (this process touches none); custom starts are validated here as always
    defer_init = (n_gpus > 1 and X is not None and Y is not None and any(b and n for b, n in zip(big, needs_dev))
                  and isinstance(random_state, (int, np.integer, type(None))))
    if defer_init and (x_init == 'custom' or y_init == 'custom'):
        defer_init = False
    if defer_init:
        U, V, Z = (np.zeros((n, n_components)) for n in (X.shape[0], X.shape[1], Y.shape[1]))
    else:
        U, V, Z = initial_factors(X, Y, U, V, Z, op_x=op_x, op_y=op_y, **init_spec)

    U, V, Z = _writable_f64(U), _writable_f64(V), _writable_f64(Z)
    if n_gpus > 1:
        from .multi_gpu import fit_multi_gpu
        U, V, Z = (np.ascontiguousarray(F) for F in (U, V, Z))
        params = dict(l1_reg=l1_reg, l2_reg=l2_reg, max_iter=max_iter, tol=tol, verbose=verbose,
                      alpha=(0.5 if solver == "mu" else float(alpha)), x_link=x_link, y_link=y_link,
                      U_non_negative=bool(U_non_negative), V_non_negative=bool(V_non_negative),
                      Z_non_negative=bool(Z_non_negative), hessian_pertubation=hessian_pertubation,
                      sg_sample_ratio=sg_sample_ratio,
                      random_state=(int(random_state) if isinstance(random_state, (int, np.integer)) else None),
                      # transform / partial updates (cmf.py:726-747: one code path for fit and transform): the U and Z sweeps are
                      # local to a rank, a fixed V simply skips the sum over the ranks
                      update_mask=(1 if update_U else 0) | (2 if update_V else 0) | (4 if update_Z else 0))
        if defer_init:
            params["init"] = dict(init_spec, random_state=(None if random_state is None else int(random_state)))
        n_iter, result = fit_multi_gpu(X, Y, U, V, Z, solver, int(n_gpus), params)
        solver_object.release()
        if _return_solver:
            return U, V, Z, n_iter, result
        return U, V, Z, n_iter
    U, V, Z, n_iter = solver_object.fit_iterative_update(X, Y, U, V, Z)
    if _return_solver:
        return U, V, Z, n_iter, solver_object
    solver_object.release()
    return U, V, Z, n_iter


def initial_factors(X, Y, U, V, Z, x_init, y_init, n_components, random_state, x_link, y_link,
                    U_non_negative, V_non_negative, Z_non_negative, op_x=None, op_y=None):
    """Initial U, V, Z as the reference's driver forms them (pycmf/cmf.py:402-430): per-side rule ('custom' validates the
    given arrays, a logit link forces 'random'), then the V merge.  ``op_x`` / ``op_y``: device copies of X / Y for the
    initialisers' passes over the data."""
    if x_init == 'custom':
        if X is not None:
            U = init_custom(U, X, n_components, 0, non_negative=U_non_negative, random_state=random_state)
            V = init_custom(V, X, n_components, 1, non_negative=V_non_negative, random_state=random_state)
    else:
        x_init = "random" if x_link == "logit" else x_init
        U, V = initialize_mf(X, n_components, init=x_init, random_state=random_state,
                             non_negative=(U_non_negative or V_non_negative), operand=op_x)
    if y_init == 'custom':
        if Y is not None:
            V = init_custom(V, Y, n_components, 0, non_negative=V_non_negative, random_state=random_state)
            Z = init_custom(Z, Y, n_components, 1, non_negative=Z_non_negative, random_state=random_state)
        V_from_y = V
    else:
        y_init = "random" if y_link == "logit" else y_init
        V_from_y, Z = initialize_mf(Y, n_components, init=y_init, random_state=random_state,
                                    non_negative=(Z_non_negative or V_non_negative), operand=op_y)
    if U_non_negative == Z_non_negative:
        V = (V + V_from_y) / 2
    elif Z_non_negative and not U_non_negative:
        V = V_from_y
    return U, V, Z


def _writable_f64(F):
    if isinstance(F, np.ndarray) and F.dtype == np.float64 and F.flags.writeable:
        return F
    return np.array(F, dtype=np.float64)


class CMF(BaseEstimator, TransformerMixin):
    """Collective Matrix Factorization on MI355X; drop-in for ``pycmf.CMF``
    (pycmf/cmf.py:459-776): same constructor keywords and defaults (:620-624), same
    ``fit`` / ``fit_transform`` / ``transform`` / ``print_topic_terms`` methods and the
    attributes ``reconstruction_err_``, ``n_components_``, ``x_weights``,
    ``components``, ``y_weights``, ``n_iter_`` (:697-704).

    Extra keywords: ``device`` (GPU ordinal, default 0), ``sg_sampler`` and ``n_gpus`` (default 1; N > 1 = ``fit`` and
    ``transform`` run data-parallel on N GPUs of the node, one worker process per GPU; MU: V reassembled by ONE sum over the
    ranks per iteration -- a single RCCL all-reduce, or reduce-scatter + all-gather around a row-blocked V update when a timed
    trial on the live ranks finds that faster; a fit with a ``random_state`` pins the single all-reduce, so that it is
    reproducible run to run).

    ``n_gpus > 1`` with ``sg_sample_ratio < 1``: the samples come from the DEVICE sampler whatever ``sg_sampler`` says -- NumPy's
    stream (pycmf/cmf_solvers.py:328-344) is global and sequential, one draw per row in sweep order, and cannot be cut across
    ranks that sweep their rows concurrently.  Such a fit reproduces the reference's sampling STATISTICS (exactly
    ``int(n * ratio)`` distinct uniform indices per row), not its iterates; ``n_gpus=1, sg_sampler='numpy'`` does.

    ``sg_sampler`` (only read when ``sg_sample_ratio < 1``): ``'numpy'`` (default) draws every row's sample from NumPy's global
    stream on the host, in the reference's order -- results reproduce the reference's for the same ``random_state``, but the
    host RNG costs one ``np.random.permutation`` per row per sweep: fine at the reference's own sizes, seconds per iteration
    beyond ~1e7 drawn indices (a ``RuntimeWarning`` says so; BASELINE config C3 would spend ~16 s per iteration there).
    ``'device'`` draws the same distribution (exactly ``int(n * ratio)`` distinct indices per row, uniform) on the GPU from a
    counter-based generator: the benchmarked path (C3: 0.24 s per iteration), statistically but not numerically NumPy's stream.
    """

    def __init__(self, n_components=None, x_init=None, y_init=None, solver='mu', alpha='auto',
                 beta_loss='frobenius', tol=1e-4, max_iter=600,
                 random_state=None, l1_reg=0., l2_reg=0., verbose=0,
                 U_non_negative=True, V_non_negative=True, Z_non_negative=True,
                 x_link="linear", y_link="linear", hessian_pertubation=0.2, sg_sample_ratio=1.,
                 device=0, sg_sampler="numpy", n_gpus=1):
        self.n_components = n_components
        self.x_init = x_init
        self.y_init = y_init
        self.solver = solver
        self.alpha = alpha
        self.beta_loss = beta_loss
        self.tol = tol
        self.max_iter = max_iter
        self.random_state = random_state
        self.l1_reg = l1_reg
        self.l2_reg = l2_reg
        self.verbose = verbose
        self.U_non_negative = U_non_negative
        self.V_non_negative = V_non_negative
        self.Z_non_negative = Z_non_negative
        self.x_link = x_link
        self.y_link = y_link
        self.hessian_pertubation = hessian_pertubation
        self.sg_sample_ratio = sg_sample_ratio
        self.device = device
        self.sg_sampler = sg_sampler
        self.n_gpus = n_gpus

    def _kwargs(self):
        return dict(solver=self.solver, beta_loss=self.beta_loss, tol=self.tol, max_iter=self.max_iter,
                    l1_reg=self.l1_reg, l2_reg=self.l2_reg, random_state=self.random_state,
                    verbose=self.verbose, U_non_negative=self.U_non_negative,
                    V_non_negative=self.V_non_negative, Z_non_negative=self.Z_non_negative,
                    x_link=self.x_link, y_link=self.y_link,
                    hessian_pertubation=self.hessian_pertubation,
                    sg_sample_ratio=self.sg_sample_ratio, device=self.device, sg_sampler=self.sg_sampler)

    def fit_transform(self, X, Y, U=None, V=None, Z=None):
        X = check_array(X, accept_sparse=('csr', 'csc'), dtype=float)
        Y = check_array(Y, accept_sparse=('csr', 'csc'), dtype=float)
        if X.shape[1] != Y.shape[0]:
            raise ValueError("Expected X.shape[1] == Y.shape[0], " +
                             "found X.shape = {}, Y.shape = {}".format(X.shape, Y.shape))
        U, V, Z, n_iter_, solver_object = collective_matrix_factorization(
            X=X, Y=Y, U=U, V=V, Z=Z, n_components=self.n_components,
            x_init=self.x_init, y_init=self.y_init, alpha=self.alpha, n_gpus=self.n_gpus,
            _return_solver=True, **self._kwargs())
        # unweighted sum of the two residual norms, evaluated on the device where the
        # data and the final factors still live (cmf.py:697-698)
        self.reconstruction_err_ = solver_object.reconstruction_error()
        solver_object.release()
        self.n_components_ = U.shape[1]
        self.x_weights = U
        self.components = V
        self.y_weights = Z
        self.n_iter_ = n_iter_
        return U, V, Z

    def fit(self, X, Y, **params):
        self.fit_transform(X, Y, **params)
        return self

    def transform(self, X, Y):
        """Re-fit U and/or Z with the learnt components V held fixed; pass ``None`` for
        the side that should be left alone (cmf.py:726-747)."""
        assert hasattr(self, "components")
        update_U = X is not None
        update_Z = Y is not None
        alpha = 1 if Y is None else 0 if X is None else "auto"
        U = None if update_U else self.x_weights
        Z = None if update_Z else self.y_weights
        U, V, Z, _ = collective_matrix_factorization(
            X=X, Y=Y, U=U, V=self.components, Z=Z, n_components=self.n_components,
            x_init="custom", y_init="custom", alpha=alpha,
            update_U=update_U, update_V=False, update_Z=update_Z, **self._kwargs())
        return U, V, Z

    def print_topic_terms(self, vectorizer, topn_words=10, importances=True):
        """Print the top terms per topic (cmf.py:749-776); works with both the old
        ``get_feature_names`` and the current ``get_feature_names_out`` vectorizer API."""
        getter = getattr(vectorizer, "get_feature_names_out", None) or vectorizer.get_feature_names
        idx_to_word = np.array(getter())
        if importances:
            print_topic_terms_with_importances(self.x_weights, self.y_weights, idx_to_word,
                                               topn_words=topn_words)
        else:
            print_topic_terms_from_matrix(self.x_weights, idx_to_word, topn_words=topn_words)
