/* ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Plain-C CPU restatement of the arithmetic the reference obtains from scikit-sparse / CHOLMOD at its
 * factor boundary (reference scilmm/SparseCholesky.py:22-26; every use of the factor at :30,:32,:40,
 * :50-52,:93,:100; twin scilmm/Estimation/LMM.py:20-24,28,30,38,48-50,94):
 *     L L^T = V[P][:,P],   factor(b) = P^T L^-T L^-1 P b,   factor.logdet() = 2 sum log L_ii,
 *     factor.L() (CSC),    (L R)[argsort(P)].
 * The third-party implementation (scikit-sparse>=0.4.3 -> SuiteSparse CHOLMOD, unpinned, absent from this
 * container) is NOT reproduced; this is a deliberately simple and independent algorithm -- an
 * up-looking simplicial sparse Cholesky driven by elimination-tree reach (textbook formulation) -- so that
 * it shares no code and no data layout with the supernodal HIP engine it checks.
 *
 * Pinning: checked in tests/ against dense LAPACK Cholesky, SciPy SuperLU (symmetric mode) and the
 * golden vectors produced by the reference's own Python driven by a dense factor (tests/golden/).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  int32_t n;
  int64_t* colptr; /* n+1 */
  int32_t* rowidx; /* nnz(L), rows sorted within a column (diagonal first) */
  double* val;
  int32_t* perm;  /* perm[new] = old */
  int32_t* iperm; /* iperm[old] = new */
  int32_t* parent;
} oracle_factor;

static void liu_etree(int32_t n, const int64_t* up, const int32_t* ui, int32_t* parent) {
  /* up/ui: for column k the rows i < k of the (permuted) upper triangle */
  int32_t* anc = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  for (int32_t k = 0; k < n; ++k) {
    parent[k] = -1;
    anc[k] = -1;
    for (int64_t p = up[k]; p < up[k + 1]; ++p) {
      int32_t i = ui[p];
      while (i != -1 && i < k) {
        int32_t nx = anc[i];
        anc[i] = k;
        if (nx == -1) parent[i] = k;
        i = nx;
      }
    }
  }
  free(anc);
}

/* nonzero pattern of row k of L: nodes reached from the entries of upper column k by walking up the
 * etree until a marked node; returned in topological order in s[top..n-1]. */
static int32_t ereach(int32_t n, const int64_t* up, const int32_t* ui, int32_t k, const int32_t* parent, int32_t* s,
                      int32_t* mark) {
  int32_t top = n;
  mark[k] = k;
  for (int64_t p = up[k]; p < up[k + 1]; ++p) {
    int32_t i = ui[p];
    if (i >= k) continue;
    int32_t len = 0;
    for (; mark[i] != k; i = parent[i]) {
      s[len++] = i;
      mark[i] = k;
    }
    while (len > 0) s[--top] = s[--len];
  }
  return top;
}

/* V given as CSR/CSC of the full symmetric matrix (or only its lower triangle by rows): indptr, indices,
 * data, original labels.  perm[new] = old.  Returns NULL when V is not positive definite
 * (*bad_col = failing permuted column). */
oracle_factor* oracle_factorize(int32_t n, const int64_t* indptr, const int32_t* indices, const double* data,
                                const int32_t* perm, int32_t* bad_col) {
  oracle_factor* F = (oracle_factor*)calloc(1, sizeof(oracle_factor));
  F->n = n;
  F->perm = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  F->iperm = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  F->parent = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  for (int32_t i = 0; i < n; ++i) {
    F->perm[i] = perm ? perm[i] : i;
    F->iperm[F->perm[i]] = i;
  }
  /* permuted upper triangle by column (entries with new row <= new col), built from row i, col j<=i */
  int64_t* up = (int64_t*)calloc((size_t)n + 1, sizeof(int64_t));
  for (int32_t i = 0; i < n; ++i)
    for (int64_t p = indptr[i]; p < indptr[i + 1]; ++p) {
      int32_t j = indices[p];
      if (j > i) continue;
      int32_t a = F->iperm[i], b = F->iperm[j];
      up[(a > b ? a : b) + 1]++;
    }
  for (int32_t i = 0; i < n; ++i) up[i + 1] += up[i];
  int64_t unz = up[n];
  int32_t* ui = (int32_t*)malloc(sizeof(int32_t) * (size_t)(unz ? unz : 1));
  double* ux = (double*)malloc(sizeof(double) * (size_t)(unz ? unz : 1));
  int64_t* fill = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
  memcpy(fill, up, sizeof(int64_t) * (size_t)n);
  for (int32_t i = 0; i < n; ++i)
    for (int64_t p = indptr[i]; p < indptr[i + 1]; ++p) {
      int32_t j = indices[p];
      if (j > i) continue;
      int32_t a = F->iperm[i], b = F->iperm[j];
      int32_t col = a > b ? a : b, row = a > b ? b : a;
      int64_t q = fill[col]++;
      ui[q] = row;
      ux[q] = data[p];
    }
  liu_etree(n, up, ui, F->parent);
  /* column counts by row-pattern traversal */
  int32_t* s = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  int32_t* mark = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  int64_t* cnt = (int64_t*)calloc((size_t)n + 1, sizeof(int64_t));
  for (int32_t i = 0; i < n; ++i) mark[i] = -1;
  for (int32_t k = 0; k < n; ++k) {
    int32_t top = ereach(n, up, ui, k, F->parent, s, mark);
    cnt[k + 1]++; /* diagonal */
    for (int32_t t = top; t < n; ++t) cnt[s[t] + 1]++;
  }
  F->colptr = (int64_t*)malloc(sizeof(int64_t) * ((size_t)n + 1));
  F->colptr[0] = 0;
  for (int32_t i = 0; i < n; ++i) F->colptr[i + 1] = F->colptr[i] + cnt[i + 1];
  int64_t lnz = F->colptr[n];
  F->rowidx = (int32_t*)malloc(sizeof(int32_t) * (size_t)(lnz ? lnz : 1));
  F->val = (double*)malloc(sizeof(double) * (size_t)(lnz ? lnz : 1));
  /* numeric up-looking factorization */
  double* x = (double*)calloc((size_t)n, sizeof(double));
  int64_t* cur = (int64_t*)malloc(sizeof(int64_t) * (size_t)n);
  memcpy(cur, F->colptr, sizeof(int64_t) * (size_t)n);
  for (int32_t i = 0; i < n; ++i) mark[i] = -1;
  int ok = 1;
  for (int32_t k = 0; k < n && ok; ++k) {
    int32_t top = ereach(n, up, ui, k, F->parent, s, mark);
    double d = 0.0;
    for (int64_t p = up[k]; p < up[k + 1]; ++p) {
      if (ui[p] == k) d += ux[p];
      else if (ui[p] < k) x[ui[p]] += ux[p];
    }
    for (int32_t t = top; t < n; ++t) {
      int32_t j = s[t];
      double lkj = x[j] / F->val[F->colptr[j]];
      x[j] = 0.0;
      for (int64_t p = F->colptr[j] + 1; p < cur[j]; ++p) x[F->rowidx[p]] -= F->val[p] * lkj;
      d -= lkj * lkj;
      int64_t q = cur[j]++;
      F->rowidx[q] = k;
      F->val[q] = lkj;
    }
    if (!(d > 0.0) || !isfinite(d)) {
      ok = 0;
      if (bad_col) *bad_col = k;
      break;
    }
    int64_t q = cur[k]++;
    F->rowidx[q] = k;
    F->val[q] = sqrt(d);
  }
  free(up); free(ui); free(ux); free(fill); free(s); free(mark); free(cnt); free(x); free(cur);
  if (!ok) {
    free(F->colptr); free(F->rowidx); free(F->val); free(F->perm); free(F->iperm); free(F->parent); free(F);
    return NULL;
  }
  return F;
}

void oracle_free(oracle_factor* F) {
  if (!F) return;
  free(F->colptr); free(F->rowidx); free(F->val); free(F->perm); free(F->iperm); free(F->parent); free(F);
}

int64_t oracle_nnz(const oracle_factor* F) { return F->colptr[F->n]; }

void oracle_export(const oracle_factor* F, int64_t* colptr, int32_t* rowidx, double* val, int32_t* perm) {
  memcpy(colptr, F->colptr, sizeof(int64_t) * ((size_t)F->n + 1));
  memcpy(rowidx, F->rowidx, sizeof(int32_t) * (size_t)F->colptr[F->n]);
  memcpy(val, F->val, sizeof(double) * (size_t)F->colptr[F->n]);
  if (perm) memcpy(perm, F->perm, sizeof(int32_t) * (size_t)F->n);
}

double oracle_logdet(const oracle_factor* F) {
  double s = 0.0;
  for (int32_t j = 0; j < F->n; ++j) s += log(F->val[F->colptr[j]]);
  return 2.0 * s;
}

/* X = V^-1 B; B, X row-major n x r (X may alias B) */
void oracle_solve(const oracle_factor* F, const double* B, int32_t r, double* X) {
  int32_t n = F->n;
  double* y = (double*)malloc(sizeof(double) * (size_t)n * (size_t)r);
  for (int32_t i = 0; i < n; ++i) memcpy(y + (size_t)i * r, B + (size_t)F->perm[i] * r, sizeof(double) * (size_t)r);
  for (int32_t j = 0; j < n; ++j) { /* L y = b */
    double* yj = y + (size_t)j * r;
    double dj = F->val[F->colptr[j]];
    for (int32_t c = 0; c < r; ++c) yj[c] /= dj;
    for (int64_t p = F->colptr[j] + 1; p < F->colptr[j + 1]; ++p) {
      double l = F->val[p];
      double* yi = y + (size_t)F->rowidx[p] * r;
      for (int32_t c = 0; c < r; ++c) yi[c] -= l * yj[c];
    }
  }
  for (int32_t j = n - 1; j >= 0; --j) { /* L^T x = y */
    double* yj = y + (size_t)j * r;
    for (int64_t p = F->colptr[j] + 1; p < F->colptr[j + 1]; ++p) {
      double l = F->val[p];
      const double* yi = y + (size_t)F->rowidx[p] * r;
      for (int32_t c = 0; c < r; ++c) yj[c] -= l * yi[c];
    }
    double dj = F->val[F->colptr[j]];
    for (int32_t c = 0; c < r; ++c) yj[c] /= dj;
  }
  for (int32_t i = 0; i < n; ++i) memcpy(X + (size_t)F->perm[i] * r, y + (size_t)i * r, sizeof(double) * (size_t)r);
  free(y);
}

/* Z = (L R)[argsort(P)] i.e. Z[perm[i]] = (L R)[i]; R, Z row-major n x r  (SparseCholesky.py:50-51) */
void oracle_lmul(const oracle_factor* F, const double* R, int32_t r, double* Z) {
  int32_t n = F->n;
  double* y = (double*)calloc((size_t)n * (size_t)r, sizeof(double));
  for (int32_t j = 0; j < n; ++j) {
    const double* rj = R + (size_t)j * r;
    for (int64_t p = F->colptr[j]; p < F->colptr[j + 1]; ++p) {
      double l = F->val[p];
      double* yi = y + (size_t)F->rowidx[p] * r;
      for (int32_t c = 0; c < r; ++c) yi[c] += l * rj[c];
    }
  }
  for (int32_t i = 0; i < n; ++i) memcpy(Z + (size_t)F->perm[i] * r, y + (size_t)i * r, sizeof(double) * (size_t)r);
  free(y);
}

/* out[c] = sum_i (A U)_ic U_ic for a symmetric A given by its rows (full storage or lower only when
 * lower_only != 0); U row-major n x r   (compute_gradients, SparseCholesky.py:65) */
void oracle_quadforms(int32_t n, const int64_t* indptr, const int32_t* indices, const double* data, int lower_only,
                      const double* U, int32_t r, double* out) {
  for (int32_t c = 0; c < r; ++c) out[c] = 0.0;
  for (int32_t i = 0; i < n; ++i) {
    const double* ui = U + (size_t)i * r;
    for (int64_t p = indptr[i]; p < indptr[i + 1]; ++p) {
      int32_t j = indices[p];
      double a = data[p];
      if (lower_only) {
        if (j > i) continue;
        if (j != i) a *= 2.0;
      }
      const double* uj = U + (size_t)j * r;
      for (int32_t c = 0; c < r; ++c) out[c] += a * ui[c] * uj[c];
    }
  }
}
