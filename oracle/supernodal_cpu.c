/* ORACLE / CPU BASELINE -- TEST AND BENCH INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * Supernodal left-looking LL^T + multi-RHS solves on the host cores, using the BLAS/LAPACK that ships
 * inside SciPy (function pointers handed over from scipy.linalg.cython_blas / cython_lapack by
 * oracle/oracle.py -- nothing is linked).  This is the "own-CPU" stand-in for the reference's
 * sksparse/CHOLMOD supernodal path (reference scilmm/SparseCholesky.py:16-26 with mode='supernodal'),
 * which is absent from this container and from the GPU box; bench.py times it as cpu_baseline
 * (kind "port").  Its results are checked against chol_oracle.c in tests/test_oracle.py.
 *
 * Data layout is the engine's supernodal layout (column-major m x w panels, see symbolic.h) so that
 * the same symbolic analysis drives both; the arithmetic here is plain dsyrk/dgemm/dpotrf/dtrsm.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef void (*dgemm_t)(char*, char*, int*, int*, int*, double*, double*, int*, double*, int*, double*, double*, int*);
typedef void (*dsyrk_t)(char*, char*, int*, int*, double*, double*, int*, double*, double*, int*);
typedef void (*dtrsm_t)(char*, char*, char*, char*, int*, int*, double*, double*, int*, double*, int*);
typedef void (*dpotrf_t)(char*, int*, double*, int*, int*);

static dgemm_t p_dgemm;
static dsyrk_t p_dsyrk;
static dtrsm_t p_dtrsm;
static dpotrf_t p_dpotrf;

void sncpu_set_blas(void* gemm, void* syrk, void* trsm, void* potrf) {
  p_dgemm = (dgemm_t)gemm;
  p_dsyrk = (dsyrk_t)syrk;
  p_dtrsm = (dtrsm_t)trsm;
  p_dpotrf = (dpotrf_t)potrf;
}

/* Lx: panel storage (zeroed + assembled by the caller).  Returns 0 or 1+failing column.
 * sncpu_factorize_range computes the panels of supernodes [s0, s1) only (their descendants must already be final in
 * Lx): the unit of work of the multi-rank rehearsal in oracle/dist_cpu.py. */
int sncpu_factorize_range(int32_t s0, int32_t s1, const int32_t* sn_start, const int64_t* sn_rowptr, const int32_t* sn_rows,
                          const int64_t* sn_loff, const int64_t* upd_ptr, const int32_t* upd_src, const int32_t* upd_p0,
                          const int32_t* upd_p1, int32_t n, double* Lx);

int sncpu_factorize(int32_t ns, const int32_t* sn_start, const int64_t* sn_rowptr, const int32_t* sn_rows,
                    const int64_t* sn_loff, const int64_t* upd_ptr, const int32_t* upd_src, const int32_t* upd_p0,
                    const int32_t* upd_p1, int32_t n, double* Lx) {
  return sncpu_factorize_range(0, ns, sn_start, sn_rowptr, sn_rows, sn_loff, upd_ptr, upd_src, upd_p0, upd_p1, n, Lx);
}

int sncpu_factorize_range(int32_t s0, int32_t s1, const int32_t* sn_start, const int64_t* sn_rowptr, const int32_t* sn_rows,
                          const int64_t* sn_loff, const int64_t* upd_ptr, const int32_t* upd_src, const int32_t* upd_p0,
                          const int32_t* upd_p1, int32_t n, double* Lx) {
  int32_t* pos = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  size_t wcap = 1 << 20;
  double* W = (double*)malloc(sizeof(double) * wcap);
  char N = 'N', T = 'T', Lo = 'L', R = 'R';
  double one = 1.0, zero = 0.0;
  for (int32_t s = s0; s < s1; ++s) {
    int32_t c0 = sn_start[s], w = sn_start[s + 1] - c0;
    const int32_t* rs = sn_rows + sn_rowptr[s];
    int32_t m = (int32_t)(sn_rowptr[s + 1] - sn_rowptr[s]);
    double* P = Lx + sn_loff[s];
    for (int32_t t = 0; t < m; ++t) pos[rs[t]] = t;
    for (int64_t e = upd_ptr[s]; e < upd_ptr[s + 1]; ++e) {
      int32_t d = upd_src[e], p0 = upd_p0[e], p1 = upd_p1[e];
      const int32_t* rd = sn_rows + sn_rowptr[d];
      int32_t md = (int32_t)(sn_rowptr[d + 1] - sn_rowptr[d]);
      int32_t wd = sn_start[d + 1] - sn_start[d];
      const double* Pd = Lx + sn_loff[d];
      int32_t mm = md - p0, nn = p1 - p0;
      size_t need = (size_t)mm * (size_t)nn;
      if (need > wcap) {
        wcap = need * 2;
        free(W);
        W = (double*)malloc(sizeof(double) * wcap);
      }
      /* W (mm x nn) = Ld[p0:md, :] * Ld[p0:p1, :]^T  (tiny products inline: BLAS call overhead dominates) */
      if ((double)mm * nn * wd < 4096.0) {
        for (int32_t q = 0; q < nn; ++q)
          for (int32_t t = q; t < mm; ++t) {
            double acc = 0.0;
            for (int32_t k = 0; k < wd; ++k) acc += Pd[(size_t)k * md + p0 + t] * Pd[(size_t)k * md + p0 + q];
            W[(size_t)q * mm + t] = acc;
          }
      } else {
        p_dgemm(&N, &T, &mm, &nn, &wd, &one, (double*)Pd + p0, &md, (double*)Pd + p0, &md, &zero, W, &mm);
      }
      for (int32_t q = 0; q < nn; ++q) {
        double* col = P + (size_t)(rd[p0 + q] - c0) * m;
        const double* wq = W + (size_t)q * mm;
        for (int32_t t = q; t < mm; ++t) col[pos[rd[p0 + t]]] -= wq[t];
      }
    }
    int info = 0;
    if ((double)m * w * w < 8192.0) {
      /* small front: unblocked right-looking Cholesky of the whole m x w panel */
      for (int32_t j = 0; j < w && info == 0; ++j) {
        double d = P[(size_t)j * m + j];
        if (!(d > 0.0)) { info = j + 1; break; }
        d = sqrt(d);
        P[(size_t)j * m + j] = d;
        for (int32_t i = j + 1; i < m; ++i) P[(size_t)j * m + i] /= d;
        for (int32_t k = j + 1; k < w; ++k) {
          double lkj = P[(size_t)j * m + k];
          for (int32_t i = k; i < m; ++i) P[(size_t)k * m + i] -= P[(size_t)j * m + i] * lkj;
        }
      }
    } else {
      p_dpotrf(&Lo, &w, P, &m, &info);
      if (info == 0 && m > w) {
        int32_t u = m - w;
        p_dtrsm(&R, &Lo, &T, &N, &u, &w, &one, P, &m, P + w, &m);
      }
    }
    if (info != 0) {
      free(pos); free(W);
      return 1 + c0 + (info > 0 ? info - 1 : 0);
    }
    /* clear the strict upper part of the diagonal block (never referenced, keeps exports clean) */
    for (int32_t j = 1; j < w; ++j)
      for (int32_t i = 0; i < j; ++i) P[(size_t)j * m + i] = 0.0;
  }
  free(pos); free(W);
  return 0;
}

/* In-place solve on the PERMUTED right-hand side Y (column-major n x r, leading dimension n). */
void sncpu_solve(int32_t ns, const int32_t* sn_start, const int64_t* sn_rowptr, const int32_t* sn_rows,
                 const int64_t* sn_loff, const double* Lx, int32_t n, int32_t r, double* Y) {
  char N = 'N', T = 'T', Lo = 'L', Le = 'L';
  double one = 1.0, mone = -1.0, zero = 0.0;
  size_t wcap = 1 << 16;
  double* W = (double*)malloc(sizeof(double) * wcap);
  for (int32_t s = 0; s < ns; ++s) { /* forward */
    int32_t c0 = sn_start[s], w = sn_start[s + 1] - c0;
    const int32_t* rs = sn_rows + sn_rowptr[s];
    int32_t m = (int32_t)(sn_rowptr[s + 1] - sn_rowptr[s]);
    const double* P = Lx + sn_loff[s];
    p_dtrsm(&Le, &Lo, &N, &N, &w, &r, &one, (double*)P, &m, Y + c0, &n);
    int32_t u = m - w;
    if (u > 0) {
      size_t need = (size_t)u * (size_t)r;
      if (need > wcap) { wcap = need * 2; free(W); W = (double*)malloc(sizeof(double) * wcap); }
      p_dgemm(&N, &N, &u, &r, &w, &one, (double*)P + w, &m, Y + c0, &n, &zero, W, &u);
      for (int32_t c = 0; c < r; ++c)
        for (int32_t t = 0; t < u; ++t) Y[(size_t)c * n + rs[w + t]] -= W[(size_t)c * u + t];
    }
  }
  for (int32_t s = ns - 1; s >= 0; --s) { /* backward */
    int32_t c0 = sn_start[s], w = sn_start[s + 1] - c0;
    const int32_t* rs = sn_rows + sn_rowptr[s];
    int32_t m = (int32_t)(sn_rowptr[s + 1] - sn_rowptr[s]);
    const double* P = Lx + sn_loff[s];
    int32_t u = m - w;
    if (u > 0) {
      size_t need = (size_t)u * (size_t)r;
      if (need > wcap) { wcap = need * 2; free(W); W = (double*)malloc(sizeof(double) * wcap); }
      for (int32_t c = 0; c < r; ++c)
        for (int32_t t = 0; t < u; ++t) W[(size_t)c * u + t] = Y[(size_t)c * n + rs[w + t]];
      p_dgemm(&T, &N, &w, &r, &u, &mone, (double*)P + w, &m, W, &u, &one, Y + c0, &n);
    }
    p_dtrsm(&Le, &Lo, &T, &N, &w, &r, &one, (double*)P, &m, Y + c0, &n);
  }
  free(W);
}

/* Y = L * R for column-major n x r blocks (leading dimension n), both in PERMUTED labels: the supernodal form of
 * factor.L().dot(R) (reference scilmm/SparseCholesky.py:50).  The strict upper part of every diagonal block is zero
 * after sncpu_factorize, so a panel multiplies as a plain m x w matrix. */
void sncpu_lmul(int32_t ns, const int32_t* sn_start, const int64_t* sn_rowptr, const int32_t* sn_rows, const int64_t* sn_loff,
                const double* Lx, int32_t n, int32_t r, const double* R, double* Y) {
  char N = 'N';
  double one = 1.0, zero = 0.0;
  size_t wcap = 1 << 16;
  double* W = (double*)malloc(sizeof(double) * wcap);
  memset(Y, 0, sizeof(double) * (size_t)n * (size_t)r);
  for (int32_t s = 0; s < ns; ++s) {
    int32_t c0 = sn_start[s], w = sn_start[s + 1] - c0;
    const int32_t* rs = sn_rows + sn_rowptr[s];
    int32_t m = (int32_t)(sn_rowptr[s + 1] - sn_rowptr[s]);
    const double* P = Lx + sn_loff[s];
    size_t need = (size_t)m * (size_t)r;
    if (need > wcap) { wcap = need * 2; free(W); W = (double*)malloc(sizeof(double) * wcap); }
    p_dgemm(&N, &N, &m, &r, &w, &one, (double*)P, &m, (double*)R + c0, &n, &zero, W, &m);
    for (int32_t c = 0; c < r; ++c)
      for (int32_t t = 0; t < m; ++t) Y[(size_t)c * n + rs[t]] += W[(size_t)c * m + t];
  }
  free(W);
}

/* ---- pieces of the factorization and of the sweeps, one front at a time: the units of work of the multi-rank rehearsal
 * (oracle/dist_cpu.py), which runs the distribution rule of the HIP engine -- rank-local panel storage (sn_loff is the
 * RANK's offset table: own panels, ring slots), batched fan-out updates, one collective per tail block in the sweeps. */

/* target s receives the contributions of its descendants d with d_lo <= d < d_hi only */
void sncpu_update_from(int32_t s, int32_t d_lo, int32_t d_hi, const int32_t* sn_start, const int64_t* sn_rowptr,
                       const int32_t* sn_rows, const int64_t* sn_loff, const int64_t* upd_ptr, const int32_t* upd_src,
                       const int32_t* upd_p0, const int32_t* upd_p1, int32_t n, double* Lx) {
  int32_t* pos = (int32_t*)malloc(sizeof(int32_t) * (size_t)n);
  size_t wcap = 1 << 16;
  double* W = (double*)malloc(sizeof(double) * wcap);
  char N = 'N', T = 'T';
  double one = 1.0, zero = 0.0;
  int32_t c0 = sn_start[s];
  const int32_t* rs = sn_rows + sn_rowptr[s];
  int32_t m = (int32_t)(sn_rowptr[s + 1] - sn_rowptr[s]);
  double* P = Lx + sn_loff[s];
  for (int32_t t = 0; t < m; ++t) pos[rs[t]] = t;
  for (int64_t e = upd_ptr[s]; e < upd_ptr[s + 1]; ++e) {
    int32_t d = upd_src[e], p0 = upd_p0[e], p1 = upd_p1[e];
    if (d < d_lo || d >= d_hi) continue;
    const int32_t* rd = sn_rows + sn_rowptr[d];
    int32_t md = (int32_t)(sn_rowptr[d + 1] - sn_rowptr[d]);
    int32_t wd = sn_start[d + 1] - sn_start[d];
    const double* Pd = Lx + sn_loff[d];
    int32_t mm = md - p0, nn = p1 - p0;
    size_t need = (size_t)mm * (size_t)nn;
    if (need > wcap) { wcap = need * 2; free(W); W = (double*)malloc(sizeof(double) * wcap); }
    p_dgemm(&N, &T, &mm, &nn, &wd, &one, (double*)Pd + p0, &md, (double*)Pd + p0, &md, &zero, W, &mm);
    for (int32_t q = 0; q < nn; ++q) {
      double* col = P + (size_t)(rd[p0 + q] - c0) * m;
      const double* wq = W + (size_t)q * mm;
      for (int32_t t = q; t < mm; ++t) col[pos[rd[p0 + t]]] -= wq[t];
    }
  }
  free(pos); free(W);
}

/* potrf + trsm of the (fully updated) panel of front s; returns 0 or 1 + failing column */
int sncpu_finish(int32_t s, const int32_t* sn_start, const int64_t* sn_rowptr, const int64_t* sn_loff, double* Lx) {
  char N = 'N', T = 'T', Lo = 'L', R = 'R';
  double one = 1.0;
  int32_t c0 = sn_start[s], w = sn_start[s + 1] - c0;
  int32_t m = (int32_t)(sn_rowptr[s + 1] - sn_rowptr[s]);
  double* P = Lx + sn_loff[s];
  int info = 0;
  p_dpotrf(&Lo, &w, P, &m, &info);
  if (info != 0) return 1 + c0 + (info > 0 ? info - 1 : 0);
  if (m > w) {
    int32_t u = m - w;
    p_dtrsm(&R, &Lo, &T, &N, &u, &w, &one, P, &m, P + w, &m);
  }
  for (int32_t j = 1; j < w; ++j)
    for (int32_t i = 0; i < j; ++i) P[(size_t)j * m + i] = 0.0;
  return 0;
}

/* forward step of front s on column-major n x r blocks, in two halves: push == 0: Y[c0:c0+w] <- L_ss^-1 Y[c0:c0+w] (needs
 * the diagonal block only); push != 0: ACC[rows below] -= L_21 Y[c0:c0+w] (needs the whole panel: its owner's job).
 * Y and ACC may be the same array (single-process sweep). */
void sncpu_fwd_front(int32_t s, const int32_t* sn_start, const int64_t* sn_rowptr, const int32_t* sn_rows, const int64_t* sn_loff,
                     const double* Lx, int32_t n, int32_t r, double* Y, double* ACC, int32_t push) {
  char N = 'N', Lo = 'L', Le = 'L';
  double one = 1.0, zero = 0.0;
  int32_t c0 = sn_start[s], w = sn_start[s + 1] - c0;
  const int32_t* rs = sn_rows + sn_rowptr[s];
  int32_t m = (int32_t)(sn_rowptr[s + 1] - sn_rowptr[s]);
  const double* P = Lx + sn_loff[s];
  if (!push) {  /* diagonal block only: needs the w x w block, which every rank has (replicated inverse in the HIP engine) */
    p_dtrsm(&Le, &Lo, &N, &N, &w, &r, &one, (double*)P, &m, Y + c0, &n);
    return;
  }
  int32_t u = m - w;
  if (u > 0) {
    double* W = (double*)malloc(sizeof(double) * (size_t)u * (size_t)r);
    p_dgemm(&N, &N, &u, &r, &w, &one, (double*)P + w, &m, Y + c0, &n, &zero, W, &u);
    for (int32_t c = 0; c < r; ++c)
      for (int32_t t = 0; t < u; ++t) ACC[(size_t)c * n + rs[w + t]] -= W[(size_t)c * u + t];
    free(W);
  }
}

/* backward step of front s: Y[c0:c0+w] <- L_ss^-T (Y[c0:c0+w] - L_21^T Y[rows below]) */
void sncpu_bwd_front(int32_t s, const int32_t* sn_start, const int64_t* sn_rowptr, const int32_t* sn_rows, const int64_t* sn_loff,
                     const double* Lx, int32_t n, int32_t r, double* Y) {
  char N = 'N', T = 'T', Lo = 'L', Le = 'L';
  double one = 1.0, mone = -1.0;
  int32_t c0 = sn_start[s], w = sn_start[s + 1] - c0;
  const int32_t* rs = sn_rows + sn_rowptr[s];
  int32_t m = (int32_t)(sn_rowptr[s + 1] - sn_rowptr[s]);
  const double* P = Lx + sn_loff[s];
  int32_t u = m - w;
  if (u > 0) {
    double* W = (double*)malloc(sizeof(double) * (size_t)u * (size_t)r);
    for (int32_t c = 0; c < r; ++c)
      for (int32_t t = 0; t < u; ++t) W[(size_t)c * u + t] = Y[(size_t)c * n + rs[w + t]];
    p_dgemm(&T, &N, &w, &r, &u, &mone, (double*)P + w, &m, W, &u, &one, Y + c0, &n);
    free(W);
  }
  p_dtrsm(&Le, &Lo, &T, &N, &w, &r, &one, (double*)P, &m, Y + c0, &n);
}

/* Y += L[:, columns of s] * R[c0:c0+w]  (one front's share of L * R) */
void sncpu_lmul_front(int32_t s, const int32_t* sn_start, const int64_t* sn_rowptr, const int32_t* sn_rows, const int64_t* sn_loff,
                      const double* Lx, int32_t n, int32_t r, const double* R, double* Y) {
  char N = 'N';
  double one = 1.0, zero = 0.0;
  int32_t c0 = sn_start[s], w = sn_start[s + 1] - c0;
  const int32_t* rs = sn_rows + sn_rowptr[s];
  int32_t m = (int32_t)(sn_rowptr[s + 1] - sn_rowptr[s]);
  const double* P = Lx + sn_loff[s];
  double* W = (double*)malloc(sizeof(double) * (size_t)m * (size_t)r);
  p_dgemm(&N, &N, &m, &r, &w, &one, (double*)P, &m, (double*)R + c0, &n, &zero, W, &m);
  for (int32_t c = 0; c < r; ++c)
    for (int32_t t = 0; t < m; ++t) Y[(size_t)c * n + rs[t]] += W[(size_t)c * m + t];
  free(W);
}
