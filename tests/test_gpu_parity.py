"""GPU parity: the HIP engine (through the C-ABI) against the CPU oracle on the same seeded inputs.

Tolerances (fp64, same permutation): logdet / solves / L / L*R relative 1e-10.
"""
import numpy as np
import pytest
import scipy.sparse as sp

from tests.helpers import random_spd, rel_err, small_pedigree

pytestmark = pytest.mark.gpu

TOL = 1e-10


def _engine(mats, **kw):
    from scilmm_amd.factor import Symbolic
    return Symbolic(mats, **kw)


def _check_factor(mats, sigma2, sym, rs=(1, 5, 103)):
    from oracle import oracle as O
    V = sum(s * m for s, m in zip(sigma2, mats)).tocsr()
    n = V.shape[0]
    f = sym.factorize(sigma2)
    perm = f.P()
    assert sorted(perm.tolist()) == list(range(n))
    o = O.OracleFactor(V, perm)
    assert abs(f.logdet() - o.logdet()) <= TOL * max(1.0, abs(o.logdet()))
    rng = np.random.default_rng(n)
    for r in rs:
        B = rng.standard_normal((n, r))
        assert rel_err(f(B), o(B)) < TOL, ("solve", n, r)
        assert rel_err(f.lmul(B), o.lmul(B)) < TOL, ("lmul", n, r)
    b = rng.standard_normal(n)
    x = f(b)
    assert x.shape == (n,)
    assert rel_err(V @ x, b) < 1e-9
    # Factor.L(): the Cholesky factor of V[P][:,P] is unique, so the exported CSC must equal the oracle's factor
    # entry by entry at every size (the engine stores relaxation zeros explicitly: compare as matrices)
    Lg = f.L()
    Lo = o.L()
    assert Lg.shape == Lo.shape == (n, n)
    diff = (Lg - Lo).tocsr()
    scale = np.abs(Lo.data).max()
    assert (np.abs(diff.data).max() if diff.nnz else 0.0) < TOL * scale, ("L", n)
    assert sp.triu(Lg, 1).nnz == 0
    return f


@pytest.mark.parametrize("mfma", ["1", "0"])
@pytest.mark.parametrize("n,density,seed", [(1, 1.0, 0), (2, 1.0, 1), (7, 0.5, 2), (33, 0.2, 3), (64, 0.9, 4),
                                            (65, 0.9, 5), (130, 0.5, 6), (200, 0.05, 7), (300, 0.02, 8)])
def test_random_spd_single_matrix(n, density, seed, mfma, monkeypatch):
    monkeypatch.setenv("SCILMM_TUNING", "1")  # schedule switches are only honoured with this set
    monkeypatch.setenv("SCILMM_NO_MFMA", "0" if mfma == "1" else "1")
    A = random_spd(n, density, seed)
    for ordering in ("amd", "natural"):
        sym = _engine([A], ordering=ordering)
        _check_factor([A], [1.0], sym, rs=(1, 5, 103, 130))


def test_dense_block_chain():
    """A fully dense SPD matrix: exercises the split-supernode chain (the dominant case at scale)."""
    rng = np.random.default_rng(0)
    n = 300
    G = rng.standard_normal((n, n))
    A = sp.csr_matrix(G @ G.T + n * np.eye(n))
    sym = _engine([A], ordering="natural")
    _check_factor([A], [1.0], sym)


def test_two_components_identity():
    A = random_spd(150, 0.1, 11)
    I = sp.identity(150, format="csr")
    sym = _engine([A, I])
    for s2 in ([0.4, 0.6], [1.3, 0.01], [1e-3, 2.0]):
        _check_factor([A, I], s2, sym, rs=(3,))


def test_refactorize_reuses_symbolic():
    from oracle import oracle as O
    A = random_spd(120, 0.1, 12)
    I = sp.identity(120, format="csr")
    sym = _engine([A, I])
    f = sym.factorize([0.5, 0.5])
    ld1 = f.logdet()
    f.refactorize([0.2, 0.9])
    V = (0.2 * A + 0.9 * I).tocsr()
    o = O.OracleFactor(V, f.P())
    assert abs(f.logdet() - o.logdet()) < TOL * abs(o.logdet())
    assert abs(ld1 - f.logdet()) > 1e-3


def test_released_host_maps_keep_the_resident_values_usable():
    """scilmm_symbolic_release_host_maps: evaluations go on from the values in HBM; a further upload is refused."""
    from oracle import oracle as O
    from scilmm_amd._lib import ScilmmError
    A = random_spd(300, 0.05, 21)
    I = sp.identity(300, format="csr")
    sym = _engine([A, I])
    f = sym.factorize([0.5, 0.5])
    sym.release_host_maps()
    f.refactorize([0.3, 0.8])
    o = O.OracleFactor((0.3 * A + 0.8 * I).tocsr(), f.P())
    assert abs(f.logdet() - o.logdet()) < TOL * abs(o.logdet())
    with pytest.raises(ScilmmError):
        sym.set_values(0, A.data)


def test_factor_knows_what_it_holds():
    """Factor.holds: the drop-in skips the reference's extra factorization at the optimum (SparseCholesky.py:182) only when
    the resident factor IS the one asked for -- same sigma2, same values, not consumed by the selected inverse."""
    A = random_spd(200, 0.05, 3)
    I = sp.identity(200, format="csr")
    sym = _engine([A, I])
    f = sym.factorize([0.5, 0.5])
    assert f.holds([0.5, 0.5]) and not f.holds([0.5, 0.6])
    f.refactorize([0.3, 0.8])
    assert f.holds([0.3, 0.8]) and not f.holds([0.5, 0.5])
    f.refactorize_async([0.2, 0.9])
    assert not f.holds([0.2, 0.9])
    f.wait()
    assert f.holds([0.2, 0.9])
    sym.set_values(0, 2.0 * A.data)
    assert not f.holds([0.2, 0.9])
    f.refactorize([0.2, 0.9])
    assert f.holds([0.2, 0.9])
    f.inverse_traces()
    assert not f.holds([0.2, 0.9])


def test_user_permutation_and_L_uniqueness():
    from oracle import oracle as O
    A = random_spd(90, 0.15, 13)
    perm = np.random.default_rng(1).permutation(90)
    sym = _engine([A], perm=perm)
    f = sym.factorize([1.0])
    assert np.array_equal(f.P(), perm)
    d = O.DenseFactor(A, perm)
    assert rel_err(f.L().toarray(), d.L().toarray()) < TOL


def test_not_positive_definite_reports_column():
    from scilmm_amd._lib import NotPositiveDefiniteError
    A = sp.csr_matrix(np.array([[1.0, 2.0, 0.0], [2.0, 1.0, 0.0], [0.0, 0.0, 1.0]]))
    sym = _engine([A], ordering="natural")
    with pytest.raises(NotPositiveDefiniteError) as ei:
        sym.factorize([1.0])
    assert ei.value.column == 1


def test_quadforms_and_spmm():
    from oracle import oracle as O
    A = random_spd(257, 0.05, 14)
    I = sp.identity(257, format="csr")
    sym = _engine([A, I])
    rng = np.random.default_rng(3)
    for r in (1, 2, 100, 140):
        U = rng.standard_normal((257, r))
        assert rel_err(sym.quadforms(0, U), O.quadforms(A, U)) < 1e-12
        assert rel_err(sym.quadforms(1, U), (U * U).sum(axis=0)) < 1e-12
    X = rng.standard_normal((257, 3))
    assert rel_err(sym.spmm(0, X), A @ X) < 1e-12
    assert rel_err(sym.spmm(1, X), X) < 1e-12
    x = rng.standard_normal(257)
    assert rel_err(sym.spmm(0, x), A @ x) < 1e-12


def test_pedigree_10k_config():
    """BASELINE config 1 shape (10k simulated pedigree, sf 0.001), V = 0.4 A + 0.6 I."""
    from oracle import oracle as O
    A, _ = small_pedigree(10000, 0.001, 0)
    n = A.shape[0]
    I = sp.identity(n, format="csr")
    sym = _engine([A, I])
    f = sym.factorize([0.4, 0.6])
    V = (0.4 * A + 0.6 * I).tocsr()
    o = O.OracleFactor(V, f.P())
    assert abs(f.logdet() - o.logdet()) < TOL * abs(o.logdet())
    rng = np.random.default_rng(5)
    B = rng.standard_normal((n, 103))
    assert rel_err(f(B), o(B)) < TOL
    assert rel_err(f.lmul(B), o.lmul(B)) < TOL
    s = O.SuperLUFactor(V)
    assert abs(f.logdet() - s.logdet()) < 1e-9 * abs(s.logdet())
    assert rel_err(f(B[:, :4]), s(B[:, :4])) < 1e-9
    U = f(B)
    assert rel_err(sym.quadforms(0, U), O.quadforms(A, U)) < 1e-11


def test_pedigree_denser_10k():
    """sf 0.01: has ~900-wide dense fronts -> multi-block chains with real descendants."""
    from oracle import oracle as O
    A, _ = small_pedigree(10000, 0.01, 0)
    n = A.shape[0]
    I = sp.identity(n, format="csr")
    sym = _engine([A, I])
    f = sym.factorize([0.4, 0.6])
    V = (0.4 * A + 0.6 * I).tocsr()
    o = O.OracleFactor(V, f.P())
    assert abs(f.logdet() - o.logdet()) < TOL * abs(o.logdet())
    rng = np.random.default_rng(6)
    B = rng.standard_normal((n, 103))
    assert rel_err(f(B), o(B)) < TOL
    assert rel_err(f.lmul(B), o.lmul(B)) < TOL


def test_factorization_is_bitwise_reproducible(monkeypatch):
    """SCILMM_DETERMINISTIC=1: no float atomics anywhere in the factorization (the prelude -> tail contributions stay on
    the fixed-order target-coordinate path instead of k_outside): the same inputs give the same bits."""
    monkeypatch.setenv("SCILMM_DETERMINISTIC", "1")
    A, _ = small_pedigree(10000, 0.01, 0)
    n = A.shape[0]
    sym = _engine([A, sp.identity(n, format="csr")])
    f = sym.factorize([0.4, 0.6])
    ld1, L1 = f.logdet(), f.L()
    f.refactorize([0.7, 0.2])
    f.refactorize([0.4, 0.6])
    ld2, L2 = f.logdet(), f.L()
    assert ld1 == ld2
    assert np.array_equal(L1.data, L2.data) and np.array_equal(L1.indices, L2.indices)


@pytest.mark.parametrize("env", [{"SCILMM_NO_LOOKAHEAD": "1"}, {"SCILMM_LOOK_DEPTH": "1"}, {"SCILMM_LOOK_DEPTH": "3"},
                                 {"SCILMM_NO_CHAIN": "1"}, {"SCILMM_NO_MFMA": "1"}, {"SCILMM_CELL_LIMIT": "64"},
                                 {"SCILMM_HOST_CELLS": "1"}, {"SCILMM_CELL_LIMIT": "100000"}, {"SCILMM_PUSH_SLICE": "256"},
                                 {"SCILMM_CHAIN_WIDE": "1000", "SCILMM_CHAIN_CAP": "100000"},
                                 {"SCILMM_CHAIN_WIDE_T": "1", "SCILMM_CHAIN_FULL_T": "100000"},  # 64-column chain windows
                                 {"SCILMM_CHAIN_WIDE_T": "1", "SCILMM_CHAIN_FULL_T": "1"},       # 112-column chain windows
                                 {"SCILMM_DENSE": "0"}, {"SCILMM_DENSE": "1"}, {"SCILMM_DENSE": "1", "SCILMM_NO_MFMA": "1"},
                                 {"SCILMM_DENSE": "1", "SCILMM_NO_LOOKAHEAD": "1"}, {"SCILMM_OUTSIDE": "1"},
                                 {"SCILMM_OUTSIDE": "1", "SCILMM_DENSE": "1"}, {"SCILMM_OUTSIDE": "1", "SCILMM_NO_MFMA": "1"},
                                 {"SCILMM_OUTSIDE": "1", "SCILMM_NO_LOOKAHEAD": "1"}, {"SCILMM_OUTSIDE": "1", "SCILMM_LOOK_DEPTH": "4", "SCILMM_DENSE": "1"}])
def test_alternative_schedules_agree_with_oracle(monkeypatch, env):
    """Every run-time switch selects a different schedule of the SAME arithmetic: all must match the oracle."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("SCILMM_TUNING", "1")
    A, _ = small_pedigree(10000, 0.01, 5)
    n = A.shape[0]
    _check_factor([A, sp.identity(n, format="csr")], [0.35, 0.65], _engine([A, sp.identity(n, format="csr")]), rs=(5, 103))


@pytest.mark.gpu
@pytest.mark.parametrize("env", [{}, {"SCILMM_DENSE": "1"}, {"SCILMM_DENSE": "1", "SCILMM_OUTSIDE": "1"},
                                 {"SCILMM_OUTSIDE": "1"}])
def test_moved_dense_tail_matches_oracle(monkeypatch, env):
    """A pedigree whose dense tail is NOT a chain of the elimination tree (a side branch of near-dense fronts joins it
    and the tail is moved to the end of the order; symbolic.cpp step 7a): every schedule must match the oracle."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("SCILMM_TUNING", "1")
    A, _ = small_pedigree(20000, 0.01, 1)
    n = A.shape[0]
    sym = _engine([A, sp.identity(n, format="csr")])
    base = _engine([A, sp.identity(n, format="csr")], upload=False, dense_relax=-1.0)  # no tail: the plain order
    assert not np.array_equal(base.get("perm"), sym.get("perm"))  # the case really is a moved tail
    bp = sym.get("tail_blk_ptr")
    assert sym.get("tail_blk").size < (bp.size - 1) * (bp.size - 2) // 2  # ... with padding-only blocks to skip
    _check_factor([A, sp.identity(n, format="csr")], [0.35, 0.65], sym, rs=(5, 103))


def test_async_refactorize_matches_blocking_call():
    """scilmm_refactorize_async + scilmm_factor_wait (and the implicit wait of every consumer) = scilmm_refactorize."""
    from scilmm_amd._lib import NotPositiveDefiniteError
    A, _ = small_pedigree(4000, 0.01, 2)
    n = A.shape[0]
    I = sp.identity(n, format="csr")
    sym = _engine([A, I])
    f = sym.factorize([0.4, 0.6])
    f.refactorize([0.7, 0.3])
    ld, L = f.logdet(), f.L().data.copy()
    f.refactorize([0.2, 0.9])
    f.refactorize_async([0.7, 0.3]).wait()
    assert f.logdet() == ld and np.array_equal(f.L().data, L)
    f.refactorize([0.2, 0.9])
    f.refactorize_async([0.7, 0.3])
    assert f.logdet() == ld  # a consumer completes the queued factorization itself
    f.refactorize_async([1.0, -5.0])
    with pytest.raises(NotPositiveDefiniteError):
        f.wait()
    f.refactorize([0.7, 0.3])
    assert f.logdet() == ld


def test_full_size_properties_100k():
    """BASELINE config 2 (100k, sf 0.005) at full size through size-independent properties."""
    from scilmm_amd.harness.pedigree import make_problem
    mats, C, y = make_problem(100000, 0.005, seed=0)
    A = mats[0]
    n = A.shape[0]
    I = sp.identity(n, format="csr")
    sym = _engine([A, I])
    s2 = [0.4, 0.6]
    f = sym.factorize(s2)
    V = (s2[0] * A + s2[1] * I).tocsr()
    rng = np.random.default_rng(0)
    B = rng.standard_normal((n, 5))
    X = f(B)
    assert rel_err(V @ X, B) < 1e-11                       # residual
    assert rel_err(f(V @ B), B) < 1e-11                    # V^-1 (V b) = b
    ld = f.logdet()
    f2 = sym.factorize([2.0 * s2[0], 2.0 * s2[1]])
    assert abs(f2.logdet() - (ld + n * np.log(2.0))) < 1e-9 * abs(ld)   # logdet(cV) = logdet V + n log c
    R = rng.standard_normal((n, 4))
    Z = f.lmul(R)
    assert rel_err(Z.T @ f(Z), R.T @ R) < 1e-10            # Z' V^-1 Z = R'R  (Z = P^T L R)
    U = f(B)
    q = sym.quadforms(0, U)
    assert rel_err(q, ((A @ U) * U).sum(axis=0)) < 1e-11   # fused SpMM + reduce
    assert rel_err(sym.quadforms(1, U), (U * U).sum(axis=0)) < 1e-12
    info = sym.info()
    assert info.nnzL > 1e8 and info.flops > 1e12


def test_factor_L_vs_oracle_on_pedigrees():
    """Factor.L() (SparseCholesky.py:50) entry by entry against the oracle's factor on 10k-scale pedigrees."""
    from oracle import oracle as O
    for n0, sf, seed in ((10000, 0.001, 0), (6000, 0.004, 2)):
        A, _ = small_pedigree(n0, sf, seed)
        n = A.shape[0]
        I = sp.identity(n, format="csr")
        sym = _engine([A, I])
        f = sym.factorize([0.3, 0.7])
        o = O.OracleFactor((0.3 * A + 0.7 * I).tocsr(), f.P())
        Lg, Lo = f.L(), o.L()
        diff = (Lg - Lo).tocsr()
        assert (np.abs(diff.data).max() if diff.nnz else 0.0) < TOL * np.abs(Lo.data).max()
        assert Lg.nnz >= Lo.nnz


def test_full_size_properties_1m():
    """BASELINE config 3 (1M individuals, sf 0.001: the config the headline metric is quoted on) at full size on one
    MI355X through size-independent properties.  ~4 minutes (pedigree generation 40 s, analysis + plan, 1.6 PFLOP
    factorization ~40 s, twice)."""
    from scilmm_amd.harness.pedigree import make_problem
    mats, C, y = make_problem(1000000, 0.001, seed=0)
    A = mats[0]
    n = A.shape[0]
    I = sp.identity(n, format="csr")
    sym = _engine([A, I])
    info = sym.info()
    assert n > 800000 and info.nnzL > 1e10 and info.flops > 1e15
    s2 = [0.4, 0.6]
    f = sym.factorize(s2)
    rng = np.random.default_rng(0)
    B = rng.standard_normal((n, 5))
    X = f(B)
    VX = s2[0] * (A @ X) + s2[1] * X
    assert rel_err(VX, B) < 1e-10                                       # residual
    VB = s2[0] * (A @ B) + s2[1] * B
    assert rel_err(f(VB), B) < 1e-10                                    # V^-1 (V b) = b
    R = rng.standard_normal((n, 4))
    Z = f.lmul(R)
    assert rel_err(Z.T @ f(Z), R.T @ R) < 1e-9                          # Z' V^-1 Z = R'R  (Z = P^T L R)
    q = sym.quadforms(0, X)
    assert rel_err(q, ((A @ X) * X).sum(axis=0)) < 1e-11                # fused SpMM + reduce
    assert rel_err(sym.quadforms(1, X), (X * X).sum(axis=0)) < 1e-12
    ld = f.logdet()
    f.refactorize([2.0 * s2[0], 2.0 * s2[1]])                           # same handle: one 123 GB factor resident
    assert abs(f.logdet() - (ld + n * np.log(2.0))) < 1e-9 * abs(ld)    # logdet(cV) = logdet V + n log c
    assert rel_err(f(B), 0.5 * X) < 1e-10                               # (cV)^-1 b = V^-1 b / c


def test_fp32_fronts_with_fp64_sums_and_refinement(monkeypatch):
    """BASELINE configs[4]'s arithmetic ("fp64 factor with fp32 MFMA fronts") on a pedigree small enough for the
    oracle: dense-tail products on the fp32 pipe, sums in fp64.  Stated tolerances: the factor agrees with the fp64
    oracle to 1e-5 relative (fp32 operand rounding, ~6e-8 per product, amplified by the tail's depth), log-det to
    1e-7 relative, and the REFINED solves (two sweeps against the exact V) to 1e-10 -- the fp64 bar."""
    from oracle import oracle as O
    monkeypatch.setenv("SCILMM_TUNING", "1")
    monkeypatch.setenv("SCILMM_DENSE", "1")  # the tail of a 10k pedigree is narrower than the automatic threshold
    A, _ = small_pedigree(10000, 0.01, 5)
    n = A.shape[0]
    I = sp.identity(n, format="csr")
    s2 = [0.35, 0.65]
    V = (s2[0] * A + s2[1] * I).tocsr()
    sym = _engine([A, I])
    f64 = sym.factorize(s2)
    ld64, L64 = f64.logdet(), f64.L()
    del f64
    sym.set_front_precision(32)
    f = sym.factorize(s2)
    o = O.OracleFactor(V, f.P())
    Lo = o.L()
    dL = (f.L() - Lo).tocsr()
    errL = np.abs(dL.data).max() / np.abs(Lo.data).max()
    assert 0.0 < errL < 1e-5, errL                       # really a different (fp32-front) factor, within the stated bound
    assert np.abs((L64 - Lo).tocsr().data).max() / np.abs(Lo.data).max() < TOL   # the fp64 path is untouched
    assert abs(f.logdet() - o.logdet()) < 1e-7 * abs(o.logdet())
    assert abs(ld64 - o.logdet()) < TOL * abs(o.logdet())
    rng = np.random.default_rng(3)
    B = rng.standard_normal((n, 5))
    assert rel_err(f(B), o(B)) < TOL                      # refined solve
    assert rel_err(V @ f(B), B) < 1e-11
    sym.set_front_precision(64)
    f.refactorize(s2)
    assert abs(f.logdet() - ld64) < 1e-12 * abs(ld64)     # back to the all-fp64 factor (atomics: equal to rounding)


def test_outside_kernel_repeatable_to_rounding(monkeypatch):
    """k_outside (fp64 atomics for the prelude -> tail contributions; the default from a 32k-wide tail on, forced here):
    two factorizations of the same inputs agree to rounding -- not necessarily to the bit."""
    monkeypatch.setenv("SCILMM_TUNING", "1")
    monkeypatch.setenv("SCILMM_OUTSIDE", "1")
    A, _ = small_pedigree(10000, 0.01, 0)
    n = A.shape[0]
    sym = _engine([A, sp.identity(n, format="csr")])
    f = sym.factorize([0.4, 0.6])
    ld1, L1 = f.logdet(), f.L()
    f.refactorize([0.7, 0.2])
    f.refactorize([0.4, 0.6])
    ld2, L2 = f.logdet(), f.L()
    assert abs(ld1 - ld2) < 1e-12 * abs(ld1)
    assert np.array_equal(L1.indices, L2.indices) and np.abs(L1.data - L2.data).max() < 1e-12 * np.abs(L1.data).max()


def test_device_dominance_matches_oracle_and_reference_golden():
    """scilmm_dominance (csrc/dominance.hip) = oracle.dominance = the matrix the reference's own code produced, bit for
    bit (integer gathers + products rounded one by one); the mirror of `dominance(rel, ibd)` drops the zero entries like
    the reference; unknown parents, founders and an empty matrix are covered."""
    import os
    from oracle import oracle as O
    from scilmm_amd import _lib
    from scilmm_amd.Matrices import dominance
    from scilmm_amd.Matrices.Dominance import parents_of
    from scilmm_amd.harness import pedigree as H
    gold = os.path.join(os.path.dirname(__file__), "golden")
    g1 = np.load(os.path.join(gold, "G1_reml_2000.npz"))
    g2 = np.load(os.path.join(gold, "G2_lmm_dominance.npz"))
    shape = tuple(g1["A_shape"])
    A = sp.csr_matrix((g1["A_data"], g1["A_indices"], g1["A_indptr"]), shape=shape)
    rel = sp.csr_matrix((g2["rel_data"], g2["rel_indices"], g2["rel_indptr"]), shape=shape)
    Dref = sp.csr_matrix((g2["D_data"], g2["D_indices"], g2["D_indptr"]), shape=shape)
    Dref.sort_indices()
    D = dominance(rel, A)
    assert np.array_equal(D.indptr, Dref.indptr) and np.array_equal(D.indices, Dref.indices) and np.array_equal(D.data, Dref.data)
    # a larger simulated pedigree (founders have parents -1), full pattern incl. zero coefficients
    par, _, _ = H.simulate_pedigree(20000, 0.005, seed=3)
    A2 = H.ibd_from_parents(par).tocsr()
    full = _lib.dominance(A2, par)
    assert np.array_equal(full.indices, A2.indices) and np.all(full.diagonal() == 1.0)
    Do = O.dominance(par, A2)
    full.eliminate_zeros()
    assert np.array_equal(full.indptr, Do.indptr) and np.array_equal(full.indices, Do.indices) and np.array_equal(full.data, Do.data)
    assert abs(full - full.T).max() == 0.0
    assert _lib.dominance(sp.csr_matrix((0, 0)), np.zeros((0, 2), dtype=np.int32)).shape == (0, 0)
    # the reference sets D_ii = 1 for EVERY i (Dominance.py:41-42), also where ibd stores no diagonal entry
    ibd = sp.csr_matrix(np.array([[1.0, 0.0, 0.5], [0.0, 0.0, 0.0], [0.5, 0.0, 1.0]]))
    ibd.eliminate_zeros()
    relm = sp.csr_matrix((3, 3))
    Dm = dominance(relm, ibd)
    assert np.array_equal(Dm.diagonal(), np.ones(3)) and Dm[0, 2] == 0.0
    with pytest.raises(_lib.ScilmmError):
        _lib.dominance(sp.identity(3, format="csr"), np.array([[-1, -1], [7, -1], [-1, -1]]))


_CPU_PORT_100K = {}


@pytest.mark.parametrize("env", [{}, {"SCILMM_TUNING": "1", "SCILMM_DENSE": "0", "SCILMM_OUTSIDE": "0"}],
                         ids=["default-schedule", "explicit-items"])
def test_100k_config_against_cpu_port(monkeypatch, env):
    """BASELINE configs[1] at FULL size (100k individuals, sf 0.005; 1.7 TFLOP per factorization), value by value
    against the BLAS-3 CPU port (oracle/supernodal_cpu.c) factoring the same V[P][:,P]: log-det and a 103-column
    solve to 1e-10, L*R to 1e-10, and one REML evaluation through the drop-in `bolt_gradient_estimation` against
    `reml_oracle.evaluate` driven by the CPU factor with the same np.random stream (nll 1e-10, gradient 1e-7).
    Run with the default schedule (the dense-tail path of every tail of 8192+ columns: k_dense_b + k_outside, short launches at
    this size) and with the explicit update items that were this size's default until round 3, forced here."""
    import importlib
    from oracle import oracle as O
    from oracle import reml_oracle as RO
    from scilmm_amd.harness.pedigree import make_problem
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    P = importlib.import_module("scilmm_amd.SparseCholesky")
    mats, C, y = make_problem(100000, 0.005, seed=0)
    A = mats[0]
    n = A.shape[0]
    I = sp.identity(n, format="csr")
    y = y / y.std()
    chol = P.SparseCholesky()
    sym = chol.engine_for([A, I])
    s2 = np.array([0.4, 0.6])
    V = (s2[0] * A + s2[1] * I).tocsr()
    arrays, colptr = sym.arrays(), sym.get("pat_colptr")
    ref = _CPU_PORT_100K
    if not ref:
        o = O.CPUPortFactor(arrays, colptr, V)
        rng = np.random.default_rng(7)
        ref["B"] = rng.standard_normal((n, 103))
        ref["R"] = rng.standard_normal((n, 3))
        ref["perm"] = o.P().copy()
        ref["logdet"] = o.logdet()
        ref["X"] = o(ref["B"])
        ref["Z"] = o.lmul(ref["R"])
        np.random.seed(11)
        ref["eval"] = RO.evaluate(np.log(s2), [A, I], C, y, reml=True, sim_num=20,
                                  factor_of=lambda Vx: O.CPUPortFactor(arrays, colptr, Vx))
        del o
    assert np.array_equal(sym.P(), ref["perm"])          # the analysis does not depend on the schedule switches
    f = sym.factorize(s2)
    assert abs(f.logdet() - ref["logdet"]) < TOL * abs(ref["logdet"])
    assert rel_err(f(ref["B"]), ref["X"]) < TOL
    assert rel_err(f.lmul(ref["R"]), ref["Z"]) < TOL
    del f
    np.random.seed(11)
    nll, grad = P.bolt_gradient_estimation(np.log(s2), chol, [A, I], C, y, True, 20, False)
    nll_o, grad_o = ref["eval"]
    assert abs(nll - nll_o) < 1e-10 * abs(nll_o)
    assert np.abs(grad - grad_o).max() < 1e-7 * np.abs(grad_o).max()
    chol.release_factors()


def test_k3_additive_dominance_fp32_fronts_100k(monkeypatch):
    """BASELINE configs[4]'s model AND arithmetic at the largest size a CPU check reaches: 100k individuals, K = 3
    (additive A + dominance D built by the device kernel + identity; reference Estimation/LMM.py:154-169,
    Matrices/Dominance.py:12-43), fp64 factor with fp32 MFMA fronts.
      * the all-fp64 factor of V = 0.3 A + 0.1 D + 0.6 I equals the BLAS-3 CPU port's (log-det, 5-column solve: 1e-10);
      * the fp32-front factor agrees with it to 1e-5 (entries of L), 1e-7 (log-det), and its REFINED solves to 1e-10."""
    from oracle import oracle as O
    from scilmm_amd.harness.pedigree import make_problem
    monkeypatch.setenv("SCILMM_TUNING", "1")
    monkeypatch.setenv("SCILMM_DENSE", "1")  # (pins the dense-tail path; it is also the default at this width: threshold 8192 columns)
    mats, C, y = make_problem(100000, 0.005, seed=0, with_dominance="device")
    A, D = mats
    n = A.shape[0]
    I = sp.identity(n, format="csr")
    assert np.all(D.diagonal() == 1.0) and abs(D - D.T).max() == 0.0 and 0.0 < D.nnz < A.nnz + 1
    s2 = [0.3, 0.1, 0.6]
    V = (s2[0] * A + s2[1] * D + s2[2] * I).tocsr()
    sym = _engine([A, D, I])
    f = sym.factorize(s2)
    o = O.CPUPortFactor(sym.arrays(), sym.get("pat_colptr"), V)
    rng = np.random.default_rng(3)
    B = rng.standard_normal((n, 5))
    Xo = o(B)
    assert abs(f.logdet() - o.logdet()) < TOL * abs(o.logdet())
    assert rel_err(f(B), Xo) < TOL
    ld64, L64 = f.logdet(), f.L().data.copy()
    del f, o
    sym.set_front_precision(32)
    f32 = sym.factorize(s2)
    L32 = f32.L().data
    errL = np.abs(L32 - L64).max() / np.abs(L64).max()
    assert 0.0 < errL < 1e-5, errL
    assert abs(f32.logdet() - ld64) < 1e-7 * abs(ld64)
    assert rel_err(f32(B), Xo) < TOL                       # two refinement sweeps against the exact V (fp64 SpMM)
    assert rel_err(V @ f32(B), B) < 1e-11


def test_lmm_k3_evaluation_matches_oracle_20k():
    """One likelihood + gradient evaluation of the reference's second entry point, `Estimation.LMM` (LMM.py:75-110), with
    K = 3 (A + device-built D + I) on a 20k pedigree, against `reml_oracle.evaluate` on the simplicial C oracle with the
    same permutation and the same np.random stream: nll 1e-10, gradient 1e-7."""
    import importlib
    from oracle import reml_oracle as RO
    from scilmm_amd.harness.pedigree import make_problem
    LM = importlib.import_module("scilmm_amd.Estimation.LMM")
    mats, C, y = make_problem(20000, 0.01, seed=1, with_dominance="device")
    n = mats[0].shape[0]
    K3 = list(mats) + [sp.identity(n, format="csr")]
    Cf = np.hstack([np.ones((n, 1)), C[:, :1]])      # LMM prepends the intercept (LMM.py:156-157)
    chol = LM.SparseCholesky()
    x = np.log([0.3, 0.1, 0.6])
    np.random.seed(9)
    nll, grad = LM.bolt_gradient_estimation(x, chol, K3, Cf, y, True, 30, False)
    perm = chol.engine_for(K3).P()
    np.random.seed(9)
    nll_o, grad_o = RO.evaluate(x, K3, Cf, y, True, 30, perm=perm)
    assert abs(nll - nll_o) < 1e-10 * abs(nll_o)
    assert np.abs(grad - grad_o).max() < 1e-7 * np.abs(grad_o).max()
    chol.release_factors()


def test_ibd_values_built_on_the_device_match_reference_goldens_bit_for_bit():
    """the IBD (numerator relationship) VALUES computed on the device straight into the engine's value slots
    (`scilmm_ibd_values_device`: tabular recursion, one launch per generation sum) -- never on the host, never over PCIe.
    Bit-exact (dyadic rationals) against the matrices the REFERENCE's own Numerator.LD + create_numerator produced
    (goldens G0: relationship_example.csv, G1: 2000-individual simulated pedigree), against the host builder on a 100k
    pedigree, and the factor built from them equals the factor built from uploaded values."""
    import os
    from scilmm_amd import ibd
    from scilmm_amd.Matrices.Dominance import parents_of
    from scilmm_amd.harness import pedigree as H
    gold = os.path.join(os.path.dirname(__file__), "golden")
    g0 = np.load(os.path.join(gold, "G0_relationship_example.npz"), allow_pickle=True)
    g1 = np.load(os.path.join(gold, "G1_reml_2000.npz"))
    g2 = np.load(os.path.join(gold, "G2_lmm_dominance.npz"))
    shape = tuple(g1["A_shape"])
    A1 = sp.csr_matrix((g1["A_data"], g1["A_indices"], g1["A_indptr"]), shape=shape)
    par1 = parents_of(sp.csr_matrix((g2["rel_data"], g2["rel_indices"], g2["rel_indptr"]), shape=shape))
    cases = [(sp.csr_matrix(g0["A"]), parents_of(sp.csr_matrix(g0["rel"]))), (A1, par1)]
    parL, _, _ = H.simulate_pedigree(100000, 0.005, seed=0)
    cases.append((None, parL))
    for Aref, par in cases:
        P = ibd.ibd_pattern_from_parents(par)            # pattern only: no value exists on the host
        if Aref is None:
            Aref = ibd.ibd_from_parents(par)              # (host builder: itself equal to the reference on the goldens)
        Aref = Aref.tocsr()
        Aref.sort_indices()
        assert np.array_equal(P.indices, Aref.indices) and np.array_equal(P.indptr, Aref.indptr)
        n = P.shape[0]
        I = sp.identity(n, format="csr")
        sym = _engine([P, I], upload=False)
        sym.upload_values(skip=(0,))
        sym.ibd_values_from_pedigree(0, par)
        perm, colptr, prow = sym.get("perm"), sym.get("pat_colptr"), sym.get("pat_row")
        got = sp.csc_matrix((sym.values_slots(0), prow, colptr), shape=(n, n))          # tril(A[P][:,P]), permuted labels
        want = sp.tril(Aref[perm][:, perm]).tocsc()
        want.sort_indices()
        got.sort_indices()
        assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices)
        assert np.array_equal(got.data, want.data), np.abs(got.data - want.data).max()
        if n <= 2000:
            f = sym.factorize([0.4, 0.6])
            ref = _engine([Aref, I]).factorize([0.4, 0.6])
            assert f.logdet() == ref.logdet()
    with pytest.raises(Exception):
        sym.ibd_values_from_pedigree(0, par[::-1].copy())   # not in pedigree order


def test_300k_values_against_cpu_port_digest():
    """a VALUE-level check above 100k.  The BLAS-3 CPU port factorized the 300k probe workload once in the
    build container (oracle/make_digest_300k.py, 7.6e13 flops, its own analysis with 512-column blocks) and left a digest:
    log det V and, of V^-1 B for three seeded columns, every 997th row + column sums + column norms.  The HIP engine (its own
    ordering and 128-column blocks) must reproduce them: log-det 1e-10, solution entries 1e-9 of the column's largest entry."""
    import os
    import bench
    path = os.path.join(os.path.dirname(__file__), "golden", "D1_300k_cpu_port_digest.npz")
    g = np.load(path)
    A, C, y = bench.build_problem(str(g["workload"]), 0)
    n = A.shape[0]
    assert n == int(g["n"]) and A.nnz == int(g["nnz_A"]) and abs(float(A.data.sum()) - float(g["a_checksum"])) < 1e-6
    sym = _engine([A, sp.identity(n, format="csr")])
    f = sym.factorize([float(v) for v in g["sigma2"]])
    assert abs(f.logdet() - float(g["logdet"])) < 1e-10 * abs(float(g["logdet"]))
    B = np.random.default_rng(300).standard_normal((n, 3))
    X = f(B)
    scale = np.abs(X).max(axis=0)
    assert np.all(np.abs(X[::int(g["stride"])] - g["X_rows"]).max(axis=0) < 1e-9 * scale)
    assert rel_err(X.sum(axis=0), g["X_sum"]) < 1e-8 and rel_err(np.sqrt((X * X).sum(axis=0)), g["X_norm"]) < 1e-10


def test_dominance_values_built_on_the_device_in_slot_order_bit_for_bit():
    """BASELINE configs[4]'s second variance component built where it is used (`scilmm_dominance_values_device`): from a
    value-less pattern (PatternCSR) and the parent table, A by the tabular recursion and D from A's resident slots.
    Bit-exact against the matrix the REFERENCE's own `dominance(rel, ibd)` produced (golden G2, on G1's pedigree) and
    against the CSR-layout kernel on a 100k pedigree; the factor of A + D + I built this way equals the uploaded one."""
    import os
    from scilmm_amd import _lib, ibd
    from scilmm_amd.factor import PatternCSR
    from scilmm_amd.Matrices.Dominance import parents_of
    from scilmm_amd.harness import pedigree as H
    gold = os.path.join(os.path.dirname(__file__), "golden")
    g1 = np.load(os.path.join(gold, "G1_reml_2000.npz"))
    g2 = np.load(os.path.join(gold, "G2_lmm_dominance.npz"))
    shape = tuple(g1["A_shape"])
    A1 = sp.csr_matrix((g1["A_data"], g1["A_indices"], g1["A_indptr"]), shape=shape)
    D1 = sp.csr_matrix((g2["D_data"], g2["D_indices"], g2["D_indptr"]), shape=shape)   # (zero coefficients dropped, as the reference does)
    par1 = parents_of(sp.csr_matrix((g2["rel_data"], g2["rel_indices"], g2["rel_indptr"]), shape=shape))
    parL, _, _ = H.simulate_pedigree(100000, 0.005, seed=0)
    AL = ibd.ibd_from_parents(parL)
    for Aref, Dref, par in [(A1, D1, par1), (AL, _lib.dominance(AL, parL), parL)]:
        n = Aref.shape[0]
        P = ibd.ibd_pattern_from_parents(par, values=False)
        assert isinstance(P, PatternCSR) and P.data is None and P.nnz == Aref.nnz
        I = sp.identity(n, format="csr")
        sym = _engine([P, P, I], upload=False)
        sym.upload_values()                       # (only the identity has host values)
        with pytest.raises(_lib.ScilmmError):
            sym.dominance_values_from(1, 0, par)  # A's values are not resident yet
        sym.ibd_values_from_pedigree(0, par)
        sym.dominance_values_from(1, 0, par)
        perm, colptr, prow = sym.get("perm"), sym.get("pat_colptr"), sym.get("pat_row")
        for k, M in ((0, Aref), (1, Dref)):
            got = sp.csc_matrix((sym.values_slots(k), prow, colptr), shape=(n, n))
            got.eliminate_zeros()
            want = sp.tril(M.tocsr()[perm][:, perm]).tocsc()
            want.eliminate_zeros()
            want.sort_indices()
            got.sort_indices()
            assert np.array_equal(got.indptr, want.indptr) and np.array_equal(got.indices, want.indices), k
            assert np.array_equal(got.data, want.data), (k, np.abs(got.data - want.data).max())
        f = sym.factorize([0.3, 0.1, 0.6])
        Dfull = Dref if Dref.nnz == Aref.nnz else sp.csr_matrix(Dref + 0 * Aref)   # (same pattern => same analysis)
        ref = _engine([Aref, Dfull, I]).factorize([0.3, 0.1, 0.6])
        # (same assembled values; the 100k schedule sums its prelude -> tail contributions with atomics: rounding-level only)
        assert np.array_equal(f.P(), ref.P()) and abs(f.logdet() - ref.logdet()) <= (0.0 if n <= 2000 else 1e-12) * abs(ref.logdet())


def test_integration_stub_from_the_document_runs():
    """INTEGRATION.md section 2 shows the ~25-line ctypes stub a reference maintainer would put in place of
    scilmm/SparseCholesky.py:16-26.  The code block is taken from the document verbatim (only the library path is made
    absolute) and executed: the object it returns obeys the four-member factor protocol and agrees with the oracle."""
    import os
    import re
    import torch  # noqa: F401  (the HIP runtime torch ships must be the one in the process: see scilmm_amd/_lib.py)
    from oracle import oracle as O
    from scilmm_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    sec = doc[doc.index("## 2."):doc.index("## 3.")]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    assert 'C.CDLL("libscilmm_hip.so")' in code
    ns = {}
    exec(code.replace('C.CDLL("libscilmm_hip.so")', "C.CDLL(%r)" % _lib.LIB_PATH), ns)
    A = random_spd(200, 0.05, 21)
    f = ns["SparseCholesky"]()(A)
    o = O.OracleFactor(A, f.P())
    rng = np.random.default_rng(0)
    B = rng.standard_normal((200, 3))
    assert abs(f.logdet() - o.logdet()) < TOL * abs(o.logdet())
    assert rel_err(f(B), o(B)) < TOL and f(B[:, 0]).shape == (200,)
    assert rel_err(f.L().toarray(), o.L().toarray()) < TOL
    assert sorted(f.P().tolist()) == list(range(200))
