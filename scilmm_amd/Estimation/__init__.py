from .LMM import LMM, SparseCholesky  # noqa: F401
