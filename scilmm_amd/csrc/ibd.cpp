// Native pedigree -> IBD (numerator relationship) builder: SURVEY.md section 8(f) rank 1, the step BEFORE the
// hot path and the producer of its input.  Host C++/OpenMP (the GPU form comes later); same arithmetic
// as the reference's interpreted per-individual loop (scilmm/Matrices/Numerator.py:5-38):
//     L[i,:] = e_i + 1/2 (L[f_i,:] + L[m_i,:]),   D_i = 1 - 1/4 (npar_i + sum F[parents]),
//     F_i = sum_a L[i,a]^2 D_a - 1,               A = L D L^T
// and of Relationship.count_IBD_nonzero (Relationship.py:38-61) for the pattern size.
// Individuals must be in topological order (parents before children), as after the reference's topo_sort.
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <vector>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "../../include/scilmm_hip.h"
#include "host_threads.h"

namespace {

struct Ibd {
  int32_t n = 0;
  std::vector<int64_t> lptr;   // ancestor rows of L (CSR, sorted by ancestor)
  std::vector<int32_t> lidx;
  std::vector<double> lval;
  std::vector<double> D, F;
  std::vector<int64_t> aptr;   // A, both triangles, CSR sorted
  std::vector<int32_t> aidx;
  std::vector<double> aval;
};

int build_L(Ibd& B, const int32_t* par) {
  const int32_t n = B.n;
  // level = generation depth, rows of one level are independent
  std::vector<int32_t> level(n, 0);
  int32_t maxlev = 0;
  for (int32_t i = 0; i < n; ++i) {
    int32_t l = 0;
    for (int k = 0; k < 2; ++k) {
      const int32_t p = par[2 * i + k];
      if (p >= i) return SCILMM_ERR_ARG;  // not topologically ordered
      if (p >= 0) l = std::max(l, level[p] + 1);
    }
    level[i] = l;
    maxlev = std::max(maxlev, l);
  }
  std::vector<std::vector<int32_t>> bylev(maxlev + 1);
  for (int32_t i = 0; i < n; ++i) bylev[level[i]].push_back(i);
  std::vector<std::vector<int32_t>> ridx(n);
  std::vector<std::vector<double>> rval(n);
  B.D.assign(n, 1.0);
  B.F.assign(n, 0.0);
  for (int32_t l = 0; l <= maxlev; ++l) {
    const auto& rows = bylev[l];
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t t = 0; t < (int64_t)rows.size(); ++t) {
      const int32_t i = rows[t];
      const int32_t f = par[2 * i], m = par[2 * i + 1];
      std::vector<int32_t>& oi = ridx[i];
      std::vector<double>& ov = rval[i];
      // merge 1/2 L[f,:] and 1/2 L[m,:] (sorted), then append e_i
      size_t a = 0, b = 0;
      const size_t na = f >= 0 ? ridx[f].size() : 0, nb = m >= 0 ? ridx[m].size() : 0;
      oi.reserve(na + nb + 1);
      ov.reserve(na + nb + 1);
      while (a < na || b < nb) {
        const int32_t ca = a < na ? ridx[f][a] : INT32_MAX, cb = b < nb ? ridx[m][b] : INT32_MAX;
        if (ca == cb) {
          oi.push_back(ca);
          ov.push_back(0.5 * (rval[f][a] + rval[m][b]));
          ++a; ++b;
        } else if (ca < cb) {
          oi.push_back(ca);
          ov.push_back(0.5 * rval[f][a]);
          ++a;
        } else {
          oi.push_back(cb);
          ov.push_back(0.5 * rval[m][b]);
          ++b;
        }
      }
      oi.push_back(i);
      ov.push_back(1.0);
      int npar = 0;
      double fp = 0.0;
      if (f >= 0) { npar++; fp += B.F[f]; }
      if (m >= 0) { npar++; fp += B.F[m]; }
      B.D[i] = 1.0 - 0.25 * (npar + fp);
      double s = 0.0;
      for (size_t k = 0; k + 1 < oi.size(); ++k) s += ov[k] * ov[k] * B.D[oi[k]];
      s += B.D[i];  // own entry: L[i,i] = 1
      B.F[i] = s - 1.0;
    }
  }
  B.lptr.assign(n + 1, 0);
  for (int32_t i = 0; i < n; ++i) B.lptr[i + 1] = B.lptr[i] + (int64_t)ridx[i].size();
  B.lidx.resize(B.lptr[n]);
  B.lval.resize(B.lptr[n]);
#pragma omp parallel for schedule(static)
  for (int32_t i = 0; i < n; ++i) {
    std::copy(ridx[i].begin(), ridx[i].end(), B.lidx.begin() + B.lptr[i]);
    std::copy(rval[i].begin(), rval[i].end(), B.lval.begin() + B.lptr[i]);
  }
  return SCILMM_OK;
}

// A = L D L^T.  count_only: just the number of structural nonzeros (both triangles).
int64_t build_A(Ibd& B, bool count_only, bool pattern_only = false) {
  const int32_t n = B.n;
  const bool verbose = getenv("SCILMM_VERBOSE") != nullptr;
  auto tlast = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    auto now = std::chrono::steady_clock::now();
    if (verbose) fprintf(stderr, "[scilmm ibd] %-24s %8.3f s\n", what, std::chrono::duration<double>(now - tlast).count());
    tlast = now;
  };
  // columns of L: descendants lists
  std::vector<int64_t> cptr(n + 1, 0);
  for (int64_t e = 0; e < (int64_t)B.lidx.size(); ++e) cptr[B.lidx[e] + 1]++;
  for (int32_t a = 0; a < n; ++a) cptr[a + 1] += cptr[a];
  std::vector<int32_t> crow(B.lidx.size());
  std::vector<double> cval(B.lidx.size());
  {
    std::vector<int64_t> fill(cptr.begin(), cptr.end() - 1);
    for (int32_t i = 0; i < n; ++i)
      for (int64_t e = B.lptr[i]; e < B.lptr[i + 1]; ++e) {
        const int64_t f = fill[B.lidx[e]]++;
        crow[f] = i;  // increasing i within a column
        cval[f] = B.lval[e];
      }
  }
  lap("columns of L");
  // lower triangle row by row with a dense accumulator per thread
  std::vector<int64_t> lowcnt(n, 0);
  std::vector<std::vector<int32_t>> lowidx(count_only ? 0 : n);
  std::vector<std::vector<double>> lowval((count_only || pattern_only) ? 0 : n);
#pragma omp parallel
  {
    std::vector<double> acc(n, 0.0);
    std::vector<int32_t> mark(n, -1), touched;
#pragma omp for schedule(dynamic, 64)
    for (int32_t i = 0; i < n; ++i) {
      touched.clear();
      for (int64_t e = B.lptr[i]; e < B.lptr[i + 1]; ++e) {
        const int32_t a = B.lidx[e];
        const double wia = B.lval[e] * B.D[a];
        for (int64_t c = cptr[a]; c < cptr[a + 1]; ++c) {
          const int32_t j = crow[c];
          if (j > i) break;
          if (mark[j] != i) {
            mark[j] = i;
            acc[j] = 0.0;
            touched.push_back(j);
          }
          acc[j] += wia * cval[c];
        }
      }
      lowcnt[i] = (int64_t)touched.size();
      if (!count_only) {
        std::sort(touched.begin(), touched.end());
        lowidx[i] = touched;
        if (!pattern_only) {
          lowval[i].resize(touched.size());
          for (size_t k = 0; k < touched.size(); ++k) lowval[i][k] = acc[touched[k]];
        }
      }
    }
  }
  lap("lower rows");
  int64_t low = 0;
  for (int32_t i = 0; i < n; ++i) low += lowcnt[i];
  const int64_t both = 2 * low - n;  // the diagonal is always present
  if (count_only) return both;
  // symmetrise: row i = lower entries (j <= i) followed by the transposed entries (j > i)
  std::vector<int64_t> upcnt(n, 0);
  for (int32_t i = 0; i < n; ++i)
    for (int32_t j : lowidx[i])
      if (j != i) upcnt[j]++;
  B.aptr.assign(n + 1, 0);
  for (int32_t i = 0; i < n; ++i) B.aptr[i + 1] = B.aptr[i] + lowcnt[i] + upcnt[i];
  B.aidx.resize(B.aptr[n]);
  if (!pattern_only) B.aval.resize(B.aptr[n]);
  lap("count upper");
  std::vector<int64_t> fill(n);
#pragma omp parallel for schedule(static)
  for (int32_t i = 0; i < n; ++i) {
    std::copy(lowidx[i].begin(), lowidx[i].end(), B.aidx.begin() + B.aptr[i]);
    if (!pattern_only) std::copy(lowval[i].begin(), lowval[i].end(), B.aval.begin() + B.aptr[i]);
    fill[i] = B.aptr[i] + lowcnt[i];
  }
  lap("copy lower");
  for (int32_t i = 0; i < n; ++i)  // increasing i => upper parts come out sorted
    for (size_t k = 0; k < lowidx[i].size(); ++k) {
      const int32_t j = lowidx[i][k];
      if (j == i) continue;
      const int64_t f = fill[j]++;
      B.aidx[f] = i;
      if (!pattern_only) B.aval[f] = lowval[i][k];
    }
  lap("scatter upper");
  return both;
}

}  // namespace

struct scilmm_ibd {
  Ibd B;
};

extern "C" {

int scilmm_ibd_build(int32_t n, const int32_t* parents, int32_t count_only, scilmm_ibd** out, int64_t* nnz) {
  scilmm::use_host_threads();
  if (n < 0 || !parents || !out || !nnz) return SCILMM_ERR_ARG;
  scilmm_ibd* h = new scilmm_ibd();
  h->B.n = n;
  int st = build_L(h->B, parents);
  if (st != SCILMM_OK) {
    delete h;
    return st;
  }
  *nnz = build_A(h->B, count_only == 1, count_only == 2);
  *out = h;
  return SCILMM_OK;
}

int scilmm_ibd_sizes(const scilmm_ibd* h, int64_t* nnz_A, int64_t* nnz_L) {
  if (!h) return SCILMM_ERR_ARG;
  if (nnz_A) *nnz_A = (int64_t)h->B.aidx.size();
  if (nnz_L) *nnz_L = (int64_t)h->B.lidx.size();
  return SCILMM_OK;
}

int scilmm_ibd_export(const scilmm_ibd* h, int64_t* a_indptr, int32_t* a_indices, double* a_data, int64_t* l_indptr,
                      int32_t* l_indices, double* l_data, double* D, double* F) {
  if (!h) return SCILMM_ERR_ARG;
  const Ibd& B = h->B;
  const size_t n = (size_t)B.n;
  if (a_indptr && !B.aptr.empty()) {
    std::memcpy(a_indptr, B.aptr.data(), sizeof(int64_t) * (n + 1));
    if (a_indices) std::memcpy(a_indices, B.aidx.data(), sizeof(int32_t) * B.aidx.size());
    if (a_data && !B.aval.empty()) std::memcpy(a_data, B.aval.data(), sizeof(double) * B.aval.size());
  }
  if (l_indptr) {
    std::memcpy(l_indptr, B.lptr.data(), sizeof(int64_t) * (n + 1));
    if (l_indices) std::memcpy(l_indices, B.lidx.data(), sizeof(int32_t) * B.lidx.size());
    if (l_data) std::memcpy(l_data, B.lval.data(), sizeof(double) * B.lval.size());
  }
  if (D) std::memcpy(D, B.D.data(), sizeof(double) * n);
  if (F) std::memcpy(F, B.F.data(), sizeof(double) * n);
  return SCILMM_OK;
}

void scilmm_ibd_free(scilmm_ibd* h) { delete h; }

}  // extern "C"
