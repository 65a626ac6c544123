"""pycmf_amd -- MI355X-native collective matrix factorisation (drop-in for pycmf)."""
from .estimator import CMF, collective_matrix_factorization  # noqa: F401
from .solver_shell import HipMUSolver, HipNewtonSolver  # noqa: F401

__all__ = ["CMF", "collective_matrix_factorization", "HipMUSolver", "HipNewtonSolver"]
