"""scilmm_amd -- MI355X-native sparse-Cholesky REML engine behind the SciLMM surface.

Mirrors ``scilmm/__init__.py:1-2`` (star-exports of the estimator module): ``SparseCholesky`` (the class),
``REML``, ``HE``, ``run_estimates``, ``run_estimates_from_paths``, ``bolt_gradient_estimation``, ...
Importing the package does not need a GPU; constructing ``SparseCholesky()`` needs the built library,
and any numeric call needs a device (no CPU fallback).
"""
from .SparseCholesky import (SparseCholesky, REML, HE, run_estimates, run_estimates_from_paths,  # noqa: F401
                             bolt_gradient_estimation, estimate_var_comps, estimate_fixed_effects,
                             negative_log_likelihood, simulate_vector, matrices_weighted_sum, compute_gradients,
                             compute_hess, compute_varcomp_stderr)
from .factor import Symbolic, Factor  # noqa: F401
from ._lib import ScilmmError, NotPositiveDefiniteError  # noqa: F401
