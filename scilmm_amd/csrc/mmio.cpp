// Streaming MatrixMarket reader (host, C++/OpenMP) behind the CLI's --A flag.
//
// Role: the reference loads the relationship matrix with scipy.io.mmread(path).tocsr()
// (reference scilmm/SparseCholesky.py:399) -- an interpreted line-by-line parse that takes tens of minutes for the
// ~1e9-entry text file of the 1M-individual config, i.e. far longer than the fit once the solver runs on the GPU
// (SURVEY section 8f rank 3).  Here the file is memory-mapped, cut into chunks at line boundaries, parsed by all
// host cores (std::from_chars: correctly rounded, locale-free) and turned into CSR by a counting sort; duplicates
// are summed and symmetric / skew-symmetric storage is expanded, exactly what mmread(...).tocsr() yields.
// Supported: "matrix coordinate {real|integer|pattern} {general|symmetric|skew-symmetric}".
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <charconv>
#include <cstdint>
#include <cstring>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../../include/scilmm_hip.h"
#include "host_threads.h"

struct scilmm_mm {
  int32_t nrows = 0, ncols = 0;
  std::vector<int64_t> indptr;
  std::vector<int32_t> indices;
  std::vector<double> data;
  std::string err;
};

namespace {

struct Entry { int32_t i, j; double v; };

inline const char* skip_ws(const char* p, const char* e) {
  while (p < e && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
  return p;
}

// one data line "i j [v]" starting at p (not beyond e); returns the position after the line, or nullptr on a bad line
inline const char* parse_line(const char* p, const char* e, bool pattern, Entry* out, bool* blank) {
  p = skip_ws(p, e);
  if (p >= e || *p == '\n') {
    *blank = true;
    return p < e ? p + 1 : e;
  }
  *blank = false;
  long long i = 0, j = 0;
  auto r1 = std::from_chars(p, e, i);
  if (r1.ec != std::errc()) return nullptr;
  p = skip_ws(r1.ptr, e);
  auto r2 = std::from_chars(p, e, j);
  if (r2.ec != std::errc()) return nullptr;
  p = r2.ptr;
  double v = 1.0;
  if (!pattern) {
    p = skip_ws(p, e);
    if (p < e && *p == '+') ++p;
    auto r3 = std::from_chars(p, e, v);
    if (r3.ec != std::errc()) return nullptr;
    p = r3.ptr;
  }
  while (p < e && *p != '\n') ++p;
  // range-check BEFORE narrowing: an index such as 4294967297 must not wrap into a valid one (the caller compares the
  // int32 fields with the header's dimensions; -1 fails that test for every dimension)
  out->i = (i >= 1 && i <= (long long)INT32_MAX) ? (int32_t)(i - 1) : -1;
  out->j = (j >= 1 && j <= (long long)INT32_MAX) ? (int32_t)(j - 1) : -1;
  out->v = v;
  return p < e ? p + 1 : e;
}

}  // namespace

extern "C" {

int scilmm_mm_read(const char* path, scilmm_mm** out, int32_t* nrows, int32_t* ncols, int64_t* nnz) {
  scilmm::use_host_threads();
  if (!path || !out) return SCILMM_ERR_ARG;
  scilmm_mm* M = new scilmm_mm();
  *out = M;
  const int fd = open(path, O_RDONLY);
  if (fd < 0) { M->err = std::string("cannot open ") + path; return SCILMM_ERR_ARG; }
  struct stat sb;
  if (fstat(fd, &sb) != 0 || sb.st_size == 0) { close(fd); M->err = "empty or unreadable file"; return SCILMM_ERR_ARG; }
  const size_t len = (size_t)sb.st_size;
  const char* base = (const char*)mmap(nullptr, len, PROT_READ, MAP_PRIVATE, fd, 0);
  close(fd);
  if (base == MAP_FAILED) { M->err = "mmap failed"; return SCILMM_ERR_ARG; }
  struct Unmap { const char* b; size_t l; ~Unmap() { munmap((void*)b, l); } } unmap{base, len};
  (void)madvise((void*)base, len, MADV_SEQUENTIAL);
  const char* p = base;
  const char* end = base + len;
  // ---- banner
  const char* eol = (const char*)memchr(p, '\n', (size_t)(end - p));
  std::string banner(p, eol ? eol : end);
  std::transform(banner.begin(), banner.end(), banner.begin(), [](unsigned char c) { return (char)tolower(c); });
  if (banner.rfind("%%matrixmarket", 0) != 0 || banner.find("matrix") == std::string::npos ||
      banner.find("coordinate") == std::string::npos) {
    M->err = "not a MatrixMarket coordinate file";
    return SCILMM_ERR_ARG;
  }
  const bool pattern = banner.find("pattern") != std::string::npos;
  if (banner.find("complex") != std::string::npos || banner.find("hermitian") != std::string::npos) {
    M->err = "complex / hermitian MatrixMarket files are not supported";
    return SCILMM_ERR_ARG;
  }
  const bool skew = banner.find("skew-symmetric") != std::string::npos;
  const bool symm = !skew && banner.find("symmetric") != std::string::npos;
  p = eol ? eol + 1 : end;
  // ---- comments, then the size line
  while (p < end && (*p == '%' || *p == '\n' || *p == '\r')) {
    const char* q = (const char*)memchr(p, '\n', (size_t)(end - p));
    p = q ? q + 1 : end;
  }
  long long nr = 0, nc = 0, nz = 0;
  {
    const char* q = skip_ws(p, end);
    auto a = std::from_chars(q, end, nr);
    q = skip_ws(a.ptr, end);
    auto b = std::from_chars(q, end, nc);
    q = skip_ws(b.ptr, end);
    auto c = std::from_chars(q, end, nz);
    if (a.ec != std::errc() || b.ec != std::errc() || c.ec != std::errc() || nr < 0 || nc < 0 || nz < 0 || nr > 0x7fffffff ||
        nc > 0x7fffffff) {
      M->err = "bad size line";
      return SCILMM_ERR_ARG;
    }
    const char* q2 = (const char*)memchr(c.ptr, '\n', (size_t)(end - c.ptr));
    p = q2 ? q2 + 1 : end;
  }
  M->nrows = (int32_t)nr;
  M->ncols = (int32_t)nc;
  // ---- parallel parse: chunk boundaries moved forward to the next line start
  int nth = 1;
#ifdef _OPENMP
#pragma omp parallel
#pragma omp single
  nth = omp_get_num_threads();
#endif
  const size_t body = (size_t)(end - p);
  const int nchunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)nth * 4, body / (1 << 16) + 1));
  std::vector<const char*> cut(nchunk + 1);
  cut[0] = p;
  cut[nchunk] = end;
  for (int c = 1; c < nchunk; ++c) {
    const char* q = p + body * (size_t)c / (size_t)nchunk;
    const char* r = (const char*)memchr(q, '\n', (size_t)(end - q));
    cut[c] = r ? r + 1 : end;
  }
  std::vector<std::vector<Entry>> parts(nchunk);
  int bad = 0;
#pragma omp parallel for schedule(dynamic, 1)
  for (int c = 0; c < nchunk; ++c) {
    const char* q = cut[c];
    const char* e = cut[c + 1];
    std::vector<Entry>& v = parts[c];
    v.reserve((size_t)(e - q) / 12 + 16);
    while (q < e) {
      Entry en;
      bool blank = false;
      const char* nx = parse_line(q, e, pattern, &en, &blank);
      if (!nx || (!blank && (en.i < 0 || en.i >= nr || en.j < 0 || en.j >= nc))) {
#pragma omp atomic write
        bad = 1;
        break;
      }
      if (!blank) v.push_back(en);
      q = nx;
    }
  }
  if (bad) { M->err = "malformed entry line"; return SCILMM_ERR_ARG; }
  int64_t stored = 0;
  for (auto& v : parts) stored += (int64_t)v.size();
  if (stored != nz) { M->err = "entry count differs from the size line"; return SCILMM_ERR_ARG; }
  // ---- CSR by counting sort over rows (mirrored entries included), then per-row sort + duplicate sum
  std::vector<int64_t> cnt((size_t)nr + 1, 0);
  for (auto& v : parts)
    for (const Entry& en : v) {
      cnt[(size_t)en.i + 1]++;
      if ((symm || skew) && en.i != en.j) cnt[(size_t)en.j + 1]++;
    }
  for (int64_t r = 0; r < nr; ++r) cnt[(size_t)r + 1] += cnt[(size_t)r];
  const int64_t total = cnt[(size_t)nr];
  std::vector<int32_t> col((size_t)total);
  std::vector<double> val((size_t)total);
  {
    std::vector<int64_t> fill(cnt.begin(), cnt.end() - 1);
    for (auto& v : parts) {
      for (const Entry& en : v) {
        int64_t f = fill[(size_t)en.i]++;
        col[(size_t)f] = en.j;
        val[(size_t)f] = en.v;
        if ((symm || skew) && en.i != en.j) {
          f = fill[(size_t)en.j]++;
          col[(size_t)f] = en.i;
          val[(size_t)f] = skew ? -en.v : en.v;
        }
      }
      std::vector<Entry>().swap(v);
    }
  }
  std::vector<int64_t> keep((size_t)nr + 1, 0);
#pragma omp parallel
  {
    std::vector<std::pair<int32_t, double>> tmp;
#pragma omp for schedule(dynamic, 256)
    for (int64_t r = 0; r < nr; ++r) {
      const int64_t b = cnt[(size_t)r], e = cnt[(size_t)r + 1];
      tmp.resize((size_t)(e - b));
      for (int64_t t = b; t < e; ++t) tmp[(size_t)(t - b)] = {col[(size_t)t], val[(size_t)t]};
      std::stable_sort(tmp.begin(), tmp.end(), [](const std::pair<int32_t, double>& x, const std::pair<int32_t, double>& y) { return x.first < y.first; });
      int64_t w = b;
      for (size_t t = 0; t < tmp.size(); ++t) {
        if (w > b && col[(size_t)(w - 1)] == tmp[t].first) {
          val[(size_t)(w - 1)] += tmp[t].second;  // duplicates are summed in file order, as COO -> CSR does
        } else {
          col[(size_t)w] = tmp[t].first;
          val[(size_t)w] = tmp[t].second;
          ++w;
        }
      }
      keep[(size_t)r + 1] = w - b;
    }
  }
  M->indptr.assign((size_t)nr + 1, 0);
  for (int64_t r = 0; r < nr; ++r) M->indptr[(size_t)r + 1] = M->indptr[(size_t)r] + keep[(size_t)r + 1];
  M->indices.resize((size_t)M->indptr[(size_t)nr]);
  M->data.resize(M->indices.size());
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < nr; ++r) {
    const int64_t src = cnt[(size_t)r], dst = M->indptr[(size_t)r], k = keep[(size_t)r + 1];
    std::copy(col.begin() + src, col.begin() + src + k, M->indices.begin() + dst);
    std::copy(val.begin() + src, val.begin() + src + k, M->data.begin() + dst);
  }
  if (nrows) *nrows = M->nrows;
  if (ncols) *ncols = M->ncols;
  if (nnz) *nnz = (int64_t)M->indices.size();
  return SCILMM_OK;
}

int scilmm_mm_export(const scilmm_mm* M, int64_t* indptr, int32_t* indices, double* data) {
  if (!M || !indptr || !indices || !data) return SCILMM_ERR_ARG;
  std::copy(M->indptr.begin(), M->indptr.end(), indptr);
  std::copy(M->indices.begin(), M->indices.end(), indices);
  std::copy(M->data.begin(), M->data.end(), data);
  return SCILMM_OK;
}

const char* scilmm_mm_error(const scilmm_mm* M) { return M ? M->err.c_str() : "null handle"; }

void scilmm_mm_free(scilmm_mm* M) { delete M; }

}  // extern "C"
