"""CPU tests of the host symbolic phase (ordering, etree, column counts, supernodes, schedules) through the C-ABI."""
import numpy as np
import pytest
import scipy.sparse as sp

from scilmm_amd.factor import Symbolic
from tests.helpers import random_spd


def _boolean_cholesky(M, perm):
    Pm = M[perm][:, perm].toarray()
    S = Pm != 0
    n = S.shape[0]
    for k in range(n):
        r = np.where(S[k + 1:, k])[0] + k + 1
        S[np.ix_(r, r)] = True
    return np.tril(S)


@pytest.mark.parametrize("ordering", ["amd", "natural"])
def test_exact_structure_without_relaxation(ordering):
    rng = np.random.default_rng(0)
    for trial in range(12):
        n = int(rng.integers(5, 110))
        M = random_spd(n, float(rng.uniform(0.02, 0.3)), trial)
        sym = Symbolic([M], ordering=ordering, upload=False, relax_small=0, relax_w1=0, relax_w2=0, relax_z3=0.0,
                       dense_relax=-1.0)
        perm = sym.get("perm")
        assert sorted(perm.tolist()) == list(range(n))
        S = _boolean_cholesky(M, perm)
        assert np.array_equal(S.sum(axis=0), sym.get("colcount"))
        info = sym.info()
        assert info.nnzL == S.sum()
        assert info.flops == float((S.sum(axis=0).astype(float) ** 2).sum())
        st, rp, rows = sym.get("sn_start"), sym.get("sn_rowptr"), sym.get("sn_rows")
        for s in range(info.nsuper):
            assert np.array_equal(rows[rp[s]:rp[s + 1]], np.where(S[:, st[s]])[0])
            assert st[s + 1] - st[s] <= 128


def test_relaxed_supernodes_cover_true_structure_and_schedules_are_consistent():
    rng = np.random.default_rng(1)
    for trial in range(8):
        n = int(rng.integers(20, 300))
        M = random_spd(n, float(rng.uniform(0.02, 0.2)), 100 + trial)
        sym = Symbolic([M, sp.identity(n, format="csr")], upload=False)
        perm = sym.get("perm")
        S = _boolean_cholesky(M, perm)
        st, rp, rows = sym.get("sn_start"), sym.get("sn_rowptr"), sym.get("sn_rows")
        lev, par = sym.get("sn_level"), sym.get("sn_parent")
        ns = sym.info().nsuper
        for s in range(ns):
            rr = set(rows[rp[s]:rp[s + 1]].tolist())
            for j in range(st[s], st[s + 1]):
                assert set(np.where(S[:, j])[0].tolist()) <= rr
            if par[s] >= 0:
                assert lev[par[s]] > lev[s] and par[s] > s
        # every update pair (target s <- descendant d) targets a strictly higher level
        up, src, p0, p1 = sym.get("upd_ptr"), sym.get("upd_src"), sym.get("upd_p0"), sym.get("upd_p1")
        for s in range(ns):
            for e in range(up[s], up[s + 1]):
                d = src[e]
                assert lev[d] < lev[s]
                rd = rows[rp[d]:rp[d + 1]]
                assert np.all((rd[p0[e]:p1[e]] >= st[s]) & (rd[p0[e]:p1[e]] < st[s + 1]))
        # combos tile the rows of every pair exactly once
        cp, pair, ta, tb = sym.get("combo_ptr"), sym.get("combo_pair"), sym.get("combo_ta"), sym.get("combo_tb")
        covered = np.zeros(len(src), dtype=np.int64)
        for c in range(cp[-1]):
            covered[pair[c]] += tb[c] - ta[c]
        md = (rp[1:] - rp[:-1])[src]
        assert np.array_equal(covered, md - p0)


def test_user_permutation_is_honoured_exactly():
    M = random_spd(80, 0.1, 7)
    perm = np.random.default_rng(2).permutation(80)
    sym = Symbolic([M], perm=perm, upload=False)
    assert np.array_equal(sym.get("perm"), perm)
    assert np.array_equal(_boolean_cholesky(M, perm).sum(axis=0), sym.get("colcount"))


def test_invalid_permutation_is_rejected():
    from scilmm_amd._lib import ScilmmError
    M = random_spd(10, 0.3, 8)
    with pytest.raises(ScilmmError):
        Symbolic([M], perm=np.zeros(10, dtype=np.int32), upload=False)


def test_amd_fill_is_close_to_superlu_mmd_on_a_pedigree():
    import scipy.sparse.linalg as sla
    from tests.helpers import small_pedigree
    A, _ = small_pedigree(4000, 0.01, 3)
    n = A.shape[0]
    sym = Symbolic([A, sp.identity(n, format="csr")], upload=False)
    lu = sla.splu((0.5 * A + 0.5 * sp.identity(n)).tocsc(), permc_spec="MMD_AT_PLUS_A", diag_pivot_thresh=0.0,
                  options={"SymmetricMode": True})
    assert sym.info().nnzL < 1.15 * lu.L.nnz
    nat = Symbolic([A, sp.identity(n, format="csr")], ordering="natural", upload=False)
    assert sym.info().nnzL < 0.6 * nat.info().nnzL


def test_empty_and_diagonal_inputs():
    I = sp.identity(5, format="csr")
    sym = Symbolic([I], upload=False)
    info = sym.info()
    assert info.nnzL == 5 and info.n_updates == 0
    one = sp.csr_matrix(np.array([[2.0]]))
    assert Symbolic([one], upload=False).info().nnzL == 1


@pytest.mark.parametrize("size,seed", [(10000, 5), (20000, 1)])
def test_dense_tail_is_a_padded_chain(size, seed):
    """Step 7 of the analysis: the fronts from dense_first on have EVERY later column as a row, form a chain in
    index order, cover the true structure, and cost at most dense_relax x the true flops of that tail.  (In the second
    case the tail is not a chain of the elimination tree: a side branch of near-dense fronts is moved to the end of the
    order with it -- still a valid elimination order, so the reference analysis with that permutation has the same fill.)"""
    from tests.helpers import small_pedigree
    A, _ = small_pedigree(size, 0.01, seed)
    n = A.shape[0]
    mats = [A, sp.identity(n, format="csr")]
    sym = Symbolic(mats, upload=False)
    ref = Symbolic(mats, upload=False, perm=sym.get("perm"), dense_relax=-1.0)  # same order, no padding
    df = int(sym.get("dense_first")[0])
    ns = sym.info().nsuper
    assert int(ref.get("dense_first")[0]) == ref.info().nsuper
    assert 0 < df < ns and ns - df >= 4
    st, rp, rows, par = sym.get("sn_start"), sym.get("sn_rowptr"), sym.get("sn_rows"), sym.get("sn_parent")
    cc = ref.get("colcount")
    assert np.array_equal(sym.get("colcount"), cc)  # same fill column by column
    same_blocks = np.array_equal(st, ref.get("sn_start"))
    # (a moved tail -- side branches joined, clique columns sorted -- keeps ITS blocks; a fresh analysis of that order may
    # cut others: then the comparison below goes through the column counts)
    rrp, rrows = ref.get("sn_rowptr"), ref.get("sn_rows")
    fl_dense = fl_true = 0.0
    for s in range(df, ns):
        assert np.array_equal(rows[rp[s]:rp[s + 1]], np.arange(st[s], n))
        assert par[s] == (s + 1 if s + 1 < ns else -1)
        w = st[s + 1] - st[s]
        if same_blocks:
            true_rows = rrows[rrp[s]:rrp[s + 1]]
            assert np.isin(true_rows, rows[rp[s]:rp[s + 1]]).all()
            mt = float(true_rows.size)
        else:
            mt = float(cc[st[s]])  # the first column's count: a lower bound of the front's true row count
        fl_dense += w * float(n - st[s]) ** 2
        fl_true += w * mt ** 2
    assert fl_dense <= (1.12 if same_blocks else 1.25) * fl_true  # (budget 1.10 on the structure BEFORE the region is sorted; the column-count bound is looser still)
    for s in range(df):  # everything below the tail: untouched / consistent with the column counts
        if same_blocks:
            assert np.array_equal(rows[rp[s]:rp[s + 1]], rrows[rrp[s]:rrp[s + 1]])
        else:
            r = rows[rp[s]:rp[s + 1]]
            assert r.size >= cc[st[s]] and np.all(np.diff(r) > 0) and r[0] == st[s]
            assert par[s] == -1 or par[s] > s
    assert sym.info().nnzL == ref.info().nnzL  # the algorithmic count does not include the padding


def test_lower_only_and_full_inputs_give_the_same_analysis():
    """Inputs that store both halves take the row-by-row adjacency path, inputs that store the lower half only (or an
    unsymmetric pattern, whose upper entries are ignored) the scatter path: same permutation, same structure, same
    assembly maps modulo the position of the stored entries."""
    from tests.helpers import small_pedigree
    A, _ = small_pedigree(3000, 0.01, 2)
    n = A.shape[0]
    I = sp.identity(n, format="csr")
    full = Symbolic([A, I], upload=False)
    lower = Symbolic([sp.tril(A, format="csr"), I], upload=False)
    junk = sp.tril(A, format="csr") + sp.triu(sp.random(n, n, 0.001, random_state=1, format="csr"), 1)
    uns = Symbolic([junk.tocsr(), I], upload=False)
    for other in (lower, uns):
        for key in ("perm", "colcount", "sn_start", "sn_rowptr", "sn_rows", "sn_parent", "sn_level"):
            assert np.array_equal(full.get(key), other.get(key)), key
        assert full.info().nnzL == other.info().nnzL and full.info().nnz_pattern == other.info().nnz_pattern


@pytest.mark.parametrize("half", ["full", "lower"])
def test_assembly_maps_place_every_value_in_its_panel_cell(half):
    """h[val_slot] = data[val_src]; L[asm_dst[slot]] += sigma2_k * h[slot] must reproduce the lower triangle of
    P (sum_k sigma2_k A_k) P^T in the panels (rows of a panel = sn_rows of its front), zeros elsewhere."""
    from tests.helpers import small_pedigree
    A, _ = small_pedigree(2500, 0.01, 4)
    n = A.shape[0]
    mats = [A.tocsr(), sp.identity(n, format="csr")]
    given = [sp.tril(m, format="csr") for m in mats] if half == "lower" else mats
    sym = Symbolic(given, upload=False)
    s2 = [0.37, 0.81]
    info = sym.info()
    L = np.zeros(info.nnzL_stored)
    asm = sym.get("asm_dst")
    h_all = np.zeros(info.nnz_pattern)
    for k, m in enumerate(given):
        slot, src = sym.get("val_slot:%d" % k), sym.get("val_src:%d" % k)
        if slot.size == n and np.array_equal(np.sort(slot), np.arange(n)):  # diagonal-only matrix: one value per new row
            L[sym.get("diag_dst")[slot]] += s2[k] * m.data[src]
            continue
        assert np.unique(slot).size == slot.size
        h = np.zeros(info.nnz_pattern)
        h[slot] = m.data[src]
        h_all += s2[k] * h
    L += np.bincount(asm, weights=h_all, minlength=L.size)
    perm = sym.P()
    V = (s2[0] * mats[0] + s2[1] * mats[1]).tocsr()[perm][:, perm].toarray()
    st, rp, rows, loff = sym.get("sn_start"), sym.get("sn_rowptr"), sym.get("sn_rows"), sym.get("sn_loff")
    for s in range(info.nsuper):
        r = rows[rp[s]:rp[s + 1]]
        w = st[s + 1] - st[s]
        panel = L[loff[s]:loff[s] + r.size * w].reshape(w, r.size).T  # column-major panel
        want = V[np.ix_(r, np.arange(st[s], st[s + 1]))]
        assert np.array_equal(np.tril(panel[:w]), np.tril(want[:w])) and np.array_equal(panel[w:], want[w:])
        assert not np.triu(panel[:w], 1).any()                        # nothing lands above the diagonal


def test_large_sparse_graph_blocks_of_the_analysis_agree_with_the_plain_path():
    """n = 90 000 (several 32k-row blocks of the pipelined elimination-tree pass; 5-point grid, randomly labelled):
    the fill equals the one of the independent etree + column-count routine, full / lower-only inputs give the same
    value maps, and val_src always points at a stored LOWER entry."""
    from scilmm_amd import _lib
    rng = np.random.default_rng(0)
    m = 300
    n = m * m
    idx = np.arange(n).reshape(m, m)
    rows = np.concatenate([idx[:, :-1].ravel(), idx[:-1, :].ravel()])
    cols = np.concatenate([idx[:, 1:].ravel(), idx[1:, :].ravel()])
    p0 = rng.permutation(n)
    R = sp.coo_matrix((rng.random(rows.size), (p0[rows], p0[cols])), shape=(n, n)).tocsr()
    A = (R + R.T + sp.identity(n) * 10).tocsr()
    A.sort_indices()
    I = sp.identity(n, format="csr")
    sym = Symbolic([A, I], upload=False)
    perm = sym.get("perm")
    nz, _, _ = _lib.fill_count(A, perm)
    assert sym.info().nnzL == nz
    T = sp.tril(A, format="csr")
    lower = Symbolic([T, I], upload=False)
    assert np.array_equal(lower.get("perm"), perm) and lower.info().nnzL == nz
    s1, r1 = sym.get("val_slot:0"), sym.get("val_src:0")
    s2, r2 = lower.get("val_slot:0"), lower.get("val_src:0")
    a, b = np.argsort(s1), np.argsort(s2)
    assert np.array_equal(s1[a], s2[b]) and np.unique(s1).size == s1.size
    assert np.array_equal(A.data[r1[a]], T.data[r2[b]])
    rowof = np.repeat(np.arange(n), np.diff(A.indptr))
    assert (A.indices[r1] <= rowof[r1]).all()


def test_tail_block_pattern_covers_the_true_structure():
    """tail_blk lists, per dense-tail front, the later tail fronts it reaches: every block of the exact (boolean) Cholesky
    structure that holds an entry must be listed -- the engine leaves the unlisted (target, descendant) pairs out."""
    rng = np.random.default_rng(3)
    seen_tail = skipped = 0
    for trial in range(10):
        n = int(rng.integers(150, 320))
        M = random_spd(n, float(rng.uniform(0.02, 0.08)), 300 + trial)
        sym = Symbolic([M, sp.identity(n, format="csr")], upload=False, max_width=8)
        df, ns = int(sym.get("dense_first")[0]), sym.info().nsuper
        if df >= ns:
            continue
        seen_tail += 1
        S = _boolean_cholesky(M, sym.get("perm"))
        st, bp, blk = sym.get("sn_start"), sym.get("tail_blk_ptr"), sym.get("tail_blk")
        assert bp.size == ns - df + 1
        for d in range(ns - df):
            lst = blk[bp[d]:bp[d + 1]]
            assert np.all(np.diff(lst) > 0) and (lst.size == 0 or lst[0] > d) and (lst.size == 0 or lst[-1] < ns - df)
            cd = slice(st[df + d], st[df + d + 1])
            for f in range(d + 1, ns - df):
                if S[st[df + f]:st[df + f + 1], cd].any():
                    assert f in lst
                elif f not in lst:
                    skipped += 1
    assert seen_tail >= 3 and skipped > 0


def test_random_patterns_moved_tails_factor_correctly_on_the_cpu_oracle():
    """Random SPD patterns, small block widths, every ordering: whenever the dense tail is moved / its dense region sorted,
    the analysis must still describe a valid factorization -- checked numerically with the CPU supernodal oracle driven by
    the analysis' own arrays (residual of a solve) and structurally against an independent analysis under the same
    permutation (same nnz(L))."""
    from oracle import oracle as O
    rng = np.random.default_rng(7)
    tails = moved = 0
    for trial in range(40):
        n = int(rng.integers(3, 500))
        M = random_spd(n, float(rng.uniform(0.005, 0.2)), 1000 + trial)
        mw = int(rng.choice([4, 8, 16, 32, 128]))
        ordering = str(rng.choice(["amd", "natural", "nesdis", "best"]))
        mats = [M, sp.identity(n, format="csr")] if rng.random() < 0.7 else [M]
        sym = Symbolic(mats, upload=False, max_width=mw, ordering=ordering)
        info, perm = sym.info(), sym.get("perm")
        assert sorted(perm.tolist()) == list(range(n))
        tails += info.dense_first < info.nsuper
        moved += not np.array_equal(Symbolic(mats, upload=False, max_width=mw, ordering=ordering, dense_relax=-1.0).get("perm"), perm)
        V = sum(a * m for a, m in zip([0.7, 0.3], mats)).tocsr()
        Lw = sp.tril(V[perm][:, perm]).tocsc()
        Lw.sort_indices()
        assert np.array_equal(Lw.indptr, sym.get("pat_colptr"))
        cpu = O.SupernodalCPU(sym.arrays(), n)
        cpu.assemble(Lw.data.copy())
        cpu.factorize()
        b = rng.standard_normal((n, 2))
        Y = np.asfortranarray(b[perm])
        cpu.solve_permuted(Y)
        x = np.empty_like(b)
        x[perm] = Y
        assert np.abs(V @ x - b).max() < 1e-9 * np.abs(b).max()
        assert Symbolic(mats, upload=False, perm=perm, max_width=mw, dense_relax=-1.0).info().nnzL == info.nnzL
    assert tails >= 20 and moved >= 15


def test_symbolic_image_round_trip(tmp_path):
    """scilmm_symbolic_save / _load: the image of an analysis gives back every array bit for bit, is refused under another
    key (another ordering = another analysis) and when damaged, and a cached handle builds its tile combos on demand."""
    from tests.helpers import small_pedigree
    A, _ = small_pedigree(6000, 0.01, 3)
    n = A.shape[0]
    mats = [A, sp.identity(n, format="csr")]
    d = str(tmp_path)
    s1 = Symbolic(mats, upload=False, cache=d)
    s2 = Symbolic(mats, upload=False, cache=d)
    assert not s1.from_cache and s2.from_cache
    for name in ["perm", "iperm", "parent", "colcount", "sn_start", "sn_parent", "sn_rowptr", "sn_rows", "sn_loff", "sn_level",
                 "level_ptr", "level_fronts", "asm_dst", "diag_dst", "upd_ptr", "upd_src", "upd_p0", "upd_p1", "upd_jp0", "tile_base",
                 "tile_front", "level_tile_ptr", "level_tiles", "level_pair_ptr", "level_pairs", "pat_colptr", "pat_row", "inv_off",
                 "tail_blk_ptr", "tail_blk", "child_ptr", "child_idx", "val_slot:0", "val_src:0", "val_slot:1", "dense_first"]:
        assert np.array_equal(s1.get(name), s2.get(name)), name
    i1, i2 = s1.info(), s2.info()
    assert all(getattr(i1, f) == getattr(i2, f) for f, _ in i1._fields_)
    assert np.array_equal(s1.get("combo_ta"), s2.get("combo_ta")) and s2.get("combo_ta").size > 0
    assert not Symbolic(mats, upload=False, cache=d, ordering="natural").from_cache      # other inputs: other key
    import os
    f = [os.path.join(d, x) for x in os.listdir(d) if x.endswith("%016x.bin" % s1._analysis_key(n, None, "amd", {}))][0]
    with open(f, "r+b") as fh:
        fh.truncate(os.path.getsize(f) // 2)
    s3 = Symbolic(mats, upload=False, cache=d)                                            # damaged image: analysed afresh
    assert not s3.from_cache and np.array_equal(s3.get("perm"), s1.get("perm"))
