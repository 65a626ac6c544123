"""pycmf_amd -- MI355X-native collective matrix factorisation (drop-in for pycmf).

The public names are resolved on first use (PEP 562): ``from pycmf_amd import _lib`` -- what bench.py and the sharded
drivers need -- then imports NumPy only, not scikit-learn / SciPy / pandas behind the estimator."""

__all__ = ["CMF", "collective_matrix_factorization", "HipMUSolver", "HipNewtonSolver"]


def __getattr__(name):
    if name in ("CMF", "collective_matrix_factorization"):
        from . import estimator
        return getattr(estimator, name)
    if name in ("HipMUSolver", "HipNewtonSolver"):
        from . import solver_shell
        return getattr(solver_shell, name)
    raise AttributeError("module 'pycmf_amd' has no attribute %r" % name)


def __dir__():
    return sorted(list(globals()) + __all__)
