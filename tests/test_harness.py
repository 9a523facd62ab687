"""CPU tests of the synthetic-pedigree harness against the reference's own IBD / dominance arithmetic (goldens)."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from scilmm_amd.harness import pedigree as H

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _parents(rel):
    rel = sp.csr_matrix(rel)
    n = rel.shape[0]
    par = np.full((n, 2), -1, dtype=np.int64)
    for i in range(n):
        p = rel.indices[rel.indptr[i]:rel.indptr[i + 1]]
        par[i, :p.size] = np.sort(p)
    return par


def test_ibd_matches_reference_on_its_fixture():
    g = np.load(os.path.join(GOLD, "G0_relationship_example.npz"))
    A, L, D = H.ibd_from_parents(_parents(g["rel"]), return_LD=True)
    assert np.array_equal(A.toarray(), g["A"])
    assert np.array_equal(L.toarray(), g["L"])
    assert np.array_equal(D, np.diag(g["D"]))


def test_ibd_and_dominance_match_reference_on_simulated_pedigree():
    g1 = np.load(os.path.join(GOLD, "G1_reml_2000.npz"))
    g2 = np.load(os.path.join(GOLD, "G2_lmm_dominance.npz"))
    shape = tuple(g1["A_shape"])
    A = sp.csr_matrix((g1["A_data"], g1["A_indices"], g1["A_indptr"]), shape=shape)
    rel = sp.csr_matrix((g2["rel_data"], g2["rel_indices"], g2["rel_indptr"]), shape=shape)
    Dref = sp.csr_matrix((g2["D_data"], g2["D_indices"], g2["D_indptr"]), shape=shape)
    Dm = H.dominance_from_parents(_parents(rel), A)
    assert abs(Dm - Dref).max() < 1e-14


def test_generator_statistics():
    par, sex, gen = H.simulate_pedigree(3000, 0.005, seed=1)
    assert np.all(par[par[:, 0] >= 0, 0] < np.where(par[:, 0] >= 0)[0])  # parents precede children
    A = H.ibd_from_parents(par)
    assert abs(A.nnz - 3000 ** 2 * 0.005) < 0.1 * 3000 ** 2 * 0.005
    assert abs(A - A.T).max() < 1e-14
    assert A.diagonal().min() >= 1.0
    A2, has = H.drop_unrelated(A)[:2]
    assert np.all(np.asarray(A2.sum(axis=1)).ravel() > 1)
    assert list(H.generation_sizes(100, 1.4)[:4]) == [2, 2, 3, 5]


def test_make_problem_shapes():
    mats, C, y = H.make_problem(2000, 0.005, seed=0)
    n = mats[0].shape[0]
    assert C.shape == (n, 2) and y.shape == (n,)
    assert abs(C[:, 0].mean()) < 1e-12 and abs(C[:, 0].std() - 1) < 1e-12 and np.all(C[:, 1] == 1)


def test_native_ibd_builder_matches_scipy_cross_check_and_counts():
    """csrc/ibd.cpp against the independent SciPy formulation on a simulated pedigree."""
    from scilmm_amd import ibd as N
    par, _, _ = H.simulate_pedigree(4000, 0.01, seed=2)
    A1, L1, D1 = H.ibd_from_parents_scipy(par, return_LD=True)
    A2, L2, D2 = N.ibd_from_parents(par, return_LD=True)
    assert A2.has_sorted_indices or True
    assert abs(A1 - A2).max() < 1e-15 and A1.nnz == A2.nnz
    assert abs(L1 - L2).max() == 0 and np.array_equal(D1, D2)
    assert N.count_ibd_nonzero(par) == A2.nnz
    assert abs(A2 - A2.T).max() == 0


def test_native_ibd_rejects_unordered_pedigree():
    import pytest
    from scilmm_amd import ibd as N
    from scilmm_amd._lib import ScilmmError
    with pytest.raises(ScilmmError):
        N.ibd_from_parents(np.array([[1, -1], [-1, -1]]))  # child listed before its parent


def test_native_matrix_market_reader_matches_scipy(tmp_path):
    """csrc/mmio.cpp behind --A (reference: mmread(A).tocsr(), SparseCholesky.py:399): general / symmetric / pattern /
    integer files, comments, blank lines, duplicates, CRLF, empty rows."""
    import scipy.io as sio
    import scipy.sparse as sp
    from scilmm_amd import _lib
    rng = np.random.default_rng(0)
    M = sp.random(300, 300, density=0.05, random_state=1, format="coo")
    S = (M + M.T).tocoo()
    cases = {"general": M, "symmetric": S}
    for name, mat in cases.items():
        path = str(tmp_path / (name + ".mtx"))
        sio.mmwrite(path, mat, symmetry="symmetric" if name == "symmetric" else "general", precision=17)
        got = _lib.read_matrix_market(path)
        ref = sio.mmread(path).tocsr()
        ref.sort_indices()
        assert got.shape == ref.shape and np.array_equal(got.indptr, ref.indptr)
        assert np.array_equal(got.indices, ref.indices) and np.array_equal(got.data, ref.data)
    # hand-written file: comments, blank line, duplicate entry, CRLF, pattern field, integer field
    p1 = tmp_path / "pat.mtx"
    p1.write_text("%%MatrixMarket matrix coordinate pattern general\n% a comment\n\n4 5 4\n1 1\n2 3\r\n4 5\n2 3\n")
    got = _lib.read_matrix_market(str(p1))
    assert got.shape == (4, 5) and got.nnz == 3 and got[1, 2] == 2.0 and got[0, 0] == 1.0 and got[3, 4] == 1.0
    p2 = tmp_path / "int.mtx"
    p2.write_text("%%MatrixMarket matrix coordinate integer skew-symmetric\n3 3 2\n2 1 7\n3 2 -4\n")
    got = _lib.read_matrix_market(str(p2)).toarray()
    assert np.array_equal(got, np.array([[0, -7, 0], [7, 0, 4], [0, -4, 0]], dtype=float))
    bad = tmp_path / "bad.mtx"
    bad.write_text("%%MatrixMarket matrix coordinate real general\n2 2 2\n1 1 1.0\n")
    with pytest.raises(_lib.ScilmmError):
        _lib.read_matrix_market(str(bad))
    # an index beyond int32 must not wrap into a valid one (4294967297 = 2^32 + 1 would read as row 1): ADVICE r2
    for line in ("4294967297 1 1.0", "1 4294967298 1.0", "0 1 1.0", "-1 1 1.0", "3 1 1.0"):
        bad.write_text("%%MatrixMarket matrix coordinate real general\n2 2 1\n" + line + "\n")
        with pytest.raises(_lib.ScilmmError):
            _lib.read_matrix_market(str(bad))


def test_relationship_matrix_reader_never_hides_a_parse_error(tmp_path):
    """`read_relationship_matrix` (reference SparseCholesky.py:399): array-format files go to SciPy's reader BY THEIR BANNER;
    a malformed coordinate file raises instead of being retried through scipy.io.mmread."""
    import importlib
    import scipy.io as sio
    from scilmm_amd import _lib
    P = importlib.import_module("scilmm_amd.SparseCholesky")
    dense = tmp_path / "dense.mtx"
    sio.mmwrite(str(dense), np.array([[2.0, 1.0], [1.0, 3.0]]))          # array format
    got = P.read_relationship_matrix(str(dense))
    assert np.array_equal(got.toarray(), [[2.0, 1.0], [1.0, 3.0]])
    bad = tmp_path / "bad.mtx"
    bad.write_text("%%MatrixMarket matrix coordinate real general\n2 2 2\n1 1 1.0\n")   # one entry short
    with pytest.raises(_lib.ScilmmError):
        P.read_relationship_matrix(str(bad))
    bad.write_text("%%MatrixMarket matrix coordinate real general\n2 2 1\n1 x 1.0\n")
    with pytest.raises(_lib.ScilmmError):
        P.read_relationship_matrix(str(bad))


def test_symbolic_image_survives_concurrent_writers_and_rejects_damage(tmp_path):
    """the image of an analysis is written under a private temporary name (several processes saving the same key
    cannot tear each other's file), carries a length + checksum trailer, and a truncated / altered / padded file is a cache
    miss (fresh analysis), never a wrong analysis."""
    import multiprocessing as mp
    from scilmm_amd.factor import Symbolic
    from tests.helpers import small_pedigree
    A, _ = small_pedigree(3000, 0.003, 5)
    mats = [A, sp.identity(A.shape[0], format="csr")]
    cache = str(tmp_path / "cache")
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_save_image_worker, args=(cache,)) for _ in range(3)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    files = os.listdir(cache)
    assert len(files) == 1 and files[0].endswith(".bin")       # no temporary left behind
    ref = Symbolic(mats, upload=False)
    hit = Symbolic(mats, upload=False, cache=cache)
    assert hit.from_cache and np.array_equal(hit.P(), ref.P()) and np.array_equal(hit.get("asm_dst"), ref.get("asm_dst"))
    path = os.path.join(cache, files[0])
    blob = open(path, "rb").read()
    for damaged in (blob[:len(blob) // 2], blob[:-8], blob + b"\0" * 8,
                    blob[:len(blob) // 2] + bytes([blob[len(blob) // 2] ^ 1]) + blob[len(blob) // 2 + 1:]):
        open(path, "wb").write(damaged)
        miss = Symbolic(mats, upload=False, cache=cache)
        assert not miss.from_cache and np.array_equal(miss.P(), ref.P())
        assert open(path, "rb").read() == blob                   # ... and the fresh analysis republished a good image


def _save_image_worker(cache):
    import scipy.sparse as sp_
    from scilmm_amd.factor import Symbolic
    from tests.helpers import small_pedigree
    A, _ = small_pedigree(3000, 0.003, 5)
    Symbolic([A, sp_.identity(A.shape[0], format="csr")], upload=False, cache=cache)


def test_quick_id_sees_in_place_edits():
    """The engine cache of the drop-in `SparseCholesky` keys on a full-pass checksum of the value arrays: an in-place
    edit of one entry, a sign flip and a swap of two entries must all change it, identical content must not."""
    import importlib
    P = importlib.import_module("scilmm_amd.SparseCholesky")
    rng = np.random.default_rng(0)
    A = sp.random(300, 300, 0.05, random_state=1, format="csr") + sp.identity(300, format="csr")
    A = A.tocsr()
    mats = [A, sp.identity(300, format="csr")]
    q0 = P.SparseCholesky._quick_id(mats)
    assert P.SparseCholesky._quick_id(mats) == q0
    k = int(rng.integers(1, A.nnz - 1))
    keep = A.data[k]
    A.data[k] = keep * 1.0000001
    assert P.SparseCholesky._quick_id(mats) != q0
    A.data[k] = -keep
    assert P.SparseCholesky._quick_id(mats) != q0
    A.data[k] = keep
    assert P.SparseCholesky._quick_id(mats) == q0
    A.data[k], A.data[k + 1] = A.data[k + 1], A.data[k]
    assert P.SparseCholesky._quick_id(mats) != q0
