"""Factor initialisation.  Small inputs follow the reference on the host; for large ones the passes over the data (the
randomized SVD behind 'svd' / 'nndsvd*', the mean behind 'random' / 'nndsvda' / 'nndsvdar') run on the GPU copy of X / Y
(SURVEY 8(f) F2); the O((m + d) k) bookkeeping (NNDSVD's sign split, NumPy's random draws) stays on the host.

Behavioural counterpart of ``_initialize_mf`` / ``_init_custom`` / ``_check_init``
(pycmf/cmf.py:30-212); pinned by tests/golden/g5_init.npz.  Returns ``(A, B.T)``
for ``M ~ A @ B`` exactly like the reference (:202), i.e. the second factor is a
transposed *view* (F-ordered).
"""
import warnings
from math import sqrt

import numpy as np
from sklearn.utils import check_array, check_random_state
from sklearn.utils.extmath import randomized_svd
from sklearn.utils.validation import check_non_negative


# matrices with at least this many cells route the randomized SVD's big products through the GPU
DEVICE_SVD_MIN_CELLS = 4_000_000


class DeviceOperand:
    """X or Y as it already sits on the GPU (libcmfhip): the initialisers' passes over the data run there."""

    def __init__(self, ctx, which, shape):
        self.ctx, self.which, self.shape = ctx, which, shape

    def dot(self, B):
        return self.ctx.data_matmul(self.which, False, B)

    def tdot(self, B):
        return self.ctx.data_matmul(self.which, True, B)

    def mean(self):
        return self.ctx.data_sum()[self.which] / (float(self.shape[0]) * float(self.shape[1]))

    def rsvd(self, transpose, k, size, n_iter, omega):
        return self.ctx.rsvd(self.which, transpose, k, size, n_iter, omega)


def randomized_svd_device(op, n_components, random_state=None, n_oversamples=10):
    """Randomized truncated SVD (Halko, Martinsson & Tropp 2011, Alg. 4.3/4.4 + 5.1) with the same
    defaults as ``sklearn.utils.extmath.randomized_svd`` -- n_iter 'auto' (7 when k < 0.1 min(shape),
    else 4), normalised power iterations, 'auto' transposition, u-based sign flip -- and the same Gaussian test
    matrix for a given ``random_state``.  Everything but the draw of that matrix and the eigen-decomposition of
    one (k + 10)^2 matrix runs on the GPU copy of the data (``cmf_rsvd``: MFMA / SpMM passes, CholeskyQR2 in float64
    where sklearn normalises with a pivoted LU -- the same subspace)."""
    n_samples, n_features = op.shape
    size = min(n_components + n_oversamples, min(op.shape))
    n_components = min(n_components, size)
    n_iter = 7 if n_components < 0.1 * min(op.shape) else 4
    transpose = n_samples < n_features
    cols = n_samples if transpose else n_features
    rng = check_random_state(random_state)
    omega = rng.normal(size=(cols, n_components + n_oversamples))[:, :size]
    U, s, Vt = op.rsvd(transpose, n_components, size, n_iter, omega)
    # svd_flip as sklearn does it: u-based when not transposed; when transposed sklearn flips the TRANSPOSED problem v-based
    # ("to actually flip based on u and not v"), and that problem's V is M's U.  cmf_rsvd hands U and Vt back in M's
    # orientation either way, so the decision always reads M's U: the largest |entry| of every column becomes positive
    signs = np.sign(U[np.argmax(np.abs(U), axis=0), range(U.shape[1])])
    signs[signs == 0] = 1.0
    U *= signs
    Vt *= signs[:, None]
    return U, s, Vt


def _rsvd(M, k, random_state, operand):
    if operand is not None and M.shape[0] * M.shape[1] >= DEVICE_SVD_MIN_CELLS:
        return randomized_svd_device(operand, k, random_state=random_state)
    return randomized_svd(M, k, random_state=random_state)


def _l2(x):
    return sqrt(float(np.dot(x.ravel(), x.ravel())))


def validate_custom(A, shape, whom, non_negative):
    """pycmf/cmf.py:30-38."""
    A = check_array(A)
    if np.shape(A) != shape:
        raise ValueError('Array with wrong shape passed to %s. Expected %s, '
                         'but got %s ' % (whom, shape, np.shape(A)))
    if non_negative:
        check_non_negative(A, whom)
        if np.max(A) == 0:
            raise ValueError('Array passed to %s is full of zeros.' % whom)


def _mean(M, operand):
    """M.mean() -- from the device copy when there is one (a pass over a 17 GB matrix on the host costs seconds)."""
    if operand is not None and M.shape[0] * M.shape[1] >= DEVICE_SVD_MIN_CELLS:
        return operand.mean()
    return M.mean()


def _random_pair(M, k, random_state, non_negative, operand=None):
    # scale so that A @ B has roughly the mean of M (cmf.py:110-117)
    scale = np.sqrt(np.abs(_mean(M, operand)) / k)
    rng = check_random_state(random_state)
    A = scale * rng.randn(M.shape[0], k)
    B = scale * rng.randn(k, M.shape[1])
    if non_negative:
        np.abs(A, out=A)
        np.abs(B, out=B)
    return A, B


def _svd_pair(M, k, random_state, operand=None):
    # cmf.py:119-142; randomized_svd yields at most min(M.shape) triplets -> zero-pad
    n, f = M.shape
    if min(n, f) < k:
        warnings.warn('The number of components is smaller than the rank in svd initialization.' +
                      'The input will be padded with zeros to compensate for the lack of singular values.')
    Us, S, Vt = _rsvd(M, k, random_state, operand)
    if k > f:
        r = Us.shape[1]
        Up = np.zeros((n, k)); Up[:, :r] = Us
        Vp = np.zeros((k, Vt.shape[1])); Vp[:Vt.shape[0], :] = Vt
        Sp = np.zeros(k); Sp[:S.shape[0]] = S
        Us, S, Vt = Up, Sp, Vp
    root = np.diag(np.sqrt(S))
    return np.dot(Us, root), np.dot(root, Vt)


def _nndsvd_pair(M, k, variant, random_state, eps, operand=None):
    # Boutsidis & Gallopoulos NNDSVD, as in cmf.py:144-197
    Us, S, Vt = _rsvd(M, k, random_state, operand)
    A = np.zeros(Us.shape)
    B = np.zeros(Vt.shape)
    A[:, 0] = np.sqrt(S[0]) * np.abs(Us[:, 0])
    B[0, :] = np.sqrt(S[0]) * np.abs(Vt[0, :])
    for j in range(1, k):
        x, y = Us[:, j], Vt[j, :]
        xp, yp = np.maximum(x, 0), np.maximum(y, 0)
        xn, yn = np.abs(np.minimum(x, 0)), np.abs(np.minimum(y, 0))
        xp_n, yp_n, xn_n, yn_n = _l2(xp), _l2(yp), _l2(xn), _l2(yn)
        pos, neg = xp_n * yp_n, xn_n * yn_n
        if pos > neg:
            u, v, sigma = xp / xp_n, yp / yp_n, pos
        else:
            u, v, sigma = xn / xn_n, yn / yn_n, neg
        lbd = np.sqrt(S[j] * sigma)
        A[:, j] = lbd * u
        B[j, :] = lbd * v
    A[A < eps] = 0
    B[B < eps] = 0
    if variant == "nndsvda":
        avg = _mean(M, operand)
        A[A == 0] = avg
        B[B == 0] = avg
    elif variant == "nndsvdar":
        rng = check_random_state(random_state)
        avg = _mean(M, operand)
        A[A == 0] = abs(avg * rng.randn(len(A[A == 0])) / 100)
        B[B == 0] = abs(avg * rng.randn(len(B[B == 0])) / 100)
    return A, B


def initialize_mf(M, n_components, init=None, eps=1e-6, random_state=None, non_negative=False, operand=None):
    """Initial guess M ~ A @ B;  returns (A, B.T).  pycmf/cmf.py:41-202.

    ``operand`` (optional :class:`DeviceOperand` for M): large matrices run the randomized SVD's
    products on the GPU where M already resides; small ones use sklearn on the host."""
    if non_negative:
        check_non_negative(M, "MF initialization")
    n_features = M.shape[1]
    if init is None:
        if n_components < n_features:
            init = 'nndsvdar' if non_negative else 'svd'
        else:
            init = 'random'

    if init == 'random':
        A, B = _random_pair(M, n_components, random_state, non_negative, operand)
    elif init == 'svd':
        if non_negative:
            raise ValueError('SVD initialization incompatible with NMF (use nndsvd instead)')
        A, B = _svd_pair(M, n_components, random_state, operand)
    elif init in ('nndsvd', 'nndsvda', 'nndsvdar'):
        if not non_negative:
            warnings.warn('%s results in non-negative constrained factors,' % init +
                          'so SVD initialization should provide better initial estimate')
        A, B = _nndsvd_pair(M, n_components, init, random_state, eps, operand)
    else:
        raise ValueError("Invalid init argument")
    return A, B.T


def init_custom(A, M, n_components, idx, non_negative=False, random_state=None):
    """User-supplied factor (validated) or the idx-th half of a 'random' init
    (pycmf/cmf.py:205-212)."""
    if A is not None:
        validate_custom(A, (M.shape[idx], n_components), "CMF (input {})".format(idx), non_negative)
        return A
    return initialize_mf(M, n_components, init="random", random_state=random_state,
                         non_negative=non_negative)[idx]
