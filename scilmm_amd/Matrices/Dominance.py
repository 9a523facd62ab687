"""``dominance(rel, ibd)`` with the reference's signature (scilmm/Matrices/Dominance.py:12-43), computed on the MI355X.

``rel``: sparse n x n parent matrix (row i holds a nonzero in the columns of i's recorded parents, at most two -- the
reference's ``Relationship`` format), ``ibd``: the IBD matrix (CSR, both halves).  Returns the dominance matrix with a
unit diagonal; entries of ``ibd``'s pattern whose dominance coefficient is 0 are not stored (as in the reference).  The arithmetic runs in ``scilmm_dominance`` (csrc/dominance.hip); there is no
CPU fallback.
"""
import numpy as np


def parents_of(rel):
    """(n, 2) int32 parent table of a sparse parent matrix: the columns stored in each row, in stored order (the order
    the reference reads them in, Dominance.py:17-20), -1 where fewer than two are recorded."""
    rel = rel.tocsr()
    n = rel.shape[0]
    cnt = np.diff(rel.indptr)
    if cnt.max(initial=0) > 2:
        raise ValueError("a row of the parent matrix holds more than two parents")
    par = np.full((n, 2), -1, dtype=np.int32)
    has1, has2 = cnt >= 1, cnt >= 2
    par[has1, 0] = rel.indices[rel.indptr[:-1][has1]]
    par[has2, 1] = rel.indices[rel.indptr[:-1][has2] + 1]
    return par


def dominance(rel, ibd):
    from .. import _lib
    ibd = ibd.tocsr()
    D = _lib.dominance(ibd, parents_of(rel))
    D.eliminate_zeros()  # like the reference's closing sparse arithmetic (Dominance.py:41-43): zeros are not stored
    if D.shape[0] > 0 and not np.all(D.diagonal() == 1.0):
        # the device kernel writes the unit diagonal where the IBD pattern STORES one (always, for a relationship matrix);
        # the reference sets D_ii = 1 for every i whatever ibd holds there (Dominance.py:41-42)
        D = D.tolil()
        D.setdiag(1.0)
        D = D.tocsr()
    return D
