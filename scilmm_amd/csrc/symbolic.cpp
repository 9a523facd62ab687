// Host-side symbolic analysis: union pattern, fill-reducing ordering, elimination tree, postorder,
// column counts, (relaxed) supernodes, supernode row structures, left-looking update schedule,
// value-assembly maps.  See symbolic.h.  Replaces cholmod_analyze as reached from the reference at
// scilmm/SparseCholesky.py:22-26 / scilmm/Estimation/LMM.py:20-24, but runs once per pattern.
//
// All algorithms are written from their published descriptions (Liu 1990 elimination tree with
// path compression; Gilbert, Ng & Peyton 1994 skeleton column counts; Ashcraft & Grimes 1989 relaxed
// supernode amalgamation); no third-party source was available in this container.
#include "host_threads.h"
#include "symbolic.h"

#include <algorithm>
#include <atomic>
#include <cassert>
#include <cstring>
#include <numeric>
#include <queue>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <unistd.h>

namespace scilmm {

namespace {

// team size of the bucket passes that count / fill with relaxed atomics: beyond a socket's worth of cores the cache-line
// traffic of the shared counters costs more than the extra threads bring (measured on the 256-core GPU host)
#define BUCKET_THREADS std::min(16, host_threads())

// CSC-lower (strict or with diagonal) <-> CSR-lower transpose of a pattern.
void transpose_pattern(int32_t n, const std::vector<int64_t>& ptr, const std::vector<int32_t>& idx,
                       std::vector<int64_t>& tptr, std::vector<int32_t>& tidx) {
  // (count and fill with relaxed atomics on all host cores; the lists of a row come out in arbitrary order, which
  // its only reader -- Liu's elimination-tree algorithm -- does not depend on)
  tptr.assign(n + 1, 0);
  const int64_t nz = (int64_t)idx.size();
#pragma omp parallel for schedule(static) num_threads(BUCKET_THREADS)
  for (int64_t e = 0; e < nz; ++e) __atomic_fetch_add(&tptr[idx[e] + 1], 1, __ATOMIC_RELAXED);
  for (int32_t i = 0; i < n; ++i) tptr[i + 1] += tptr[i];
  tidx.resize(idx.size());
  std::vector<int64_t> fill(tptr.begin(), tptr.end() - 1);
#pragma omp parallel for schedule(dynamic, 4096) num_threads(BUCKET_THREADS)
  for (int32_t j = 0; j < n; ++j)
    for (int64_t e = ptr[j]; e < ptr[j + 1]; ++e) tidx[__atomic_fetch_add(&fill[idx[e]], 1, __ATOMIC_RELAXED)] = j;
}

// Liu's algorithm. rptr/ridx: for each row i the columns k < i with A_ik != 0.
void etree(int32_t n, const std::vector<int64_t>& rptr, const std::vector<int32_t>& ridx, std::vector<int32_t>& parent) {
  parent.assign(n, -1);
  std::vector<int32_t> anc(n, -1);
  for (int32_t i = 0; i < n; ++i) {
    for (int64_t e = rptr[i]; e < rptr[i + 1]; ++e) {
      int32_t k = ridx[e];
      while (k != -1 && k < i) {
        int32_t nx = anc[k];
        anc[k] = i;
        if (nx == -1) parent[k] = i;
        k = nx;
      }
    }
  }
}

// Postorder with children visited in increasing label order. post[k] = node visited k-th.
void postorder(int32_t n, const std::vector<int32_t>& parent, std::vector<int32_t>& post) {
  std::vector<int32_t> head(n, -1), next(n, -1);
  for (int32_t j = n - 1; j >= 0; --j) {
    if (parent[j] == -1) continue;
    next[j] = head[parent[j]];
    head[parent[j]] = j;
  }
  post.resize(n);
  int32_t k = 0;
  std::vector<int32_t> stack;
  for (int32_t r = 0; r < n; ++r) {
    if (parent[r] != -1) continue;
    stack.push_back(r);
    while (!stack.empty()) {
      int32_t p = stack.back();
      int32_t c = head[p];
      if (c == -1) {
        post[k++] = p;
        stack.pop_back();
      } else {
        head[p] = next[c];
        stack.push_back(c);
      }
    }
  }
}

int32_t find_root(std::vector<int32_t>& anc, int32_t x) {
  int32_t r = x;
  while (anc[r] != r) r = anc[r];
  while (anc[x] != r) {
    int32_t nx = anc[x];
    anc[x] = r;
    x = nx;
  }
  return r;
}

// Skeleton column counts for an arbitrary (topologically valid) labelling; post[k] = k-th node of a
// postorder of the etree.  cptr/cidx: for each column j the rows i > j with A_ij != 0.
void column_counts(int32_t n, const std::vector<int32_t>& parent, const std::vector<int32_t>& post,
                   const std::vector<int64_t>& cptr, const std::vector<int32_t>& cidx, std::vector<int32_t>& cc) {
  std::vector<int32_t> first(n, -1);
  std::vector<int64_t> delta(n);
  for (int32_t k = 0; k < n; ++k) {
    int32_t j = post[k];
    delta[j] = (first[j] == -1) ? 1 : 0;  // leaf of the etree
    for (; j != -1 && first[j] == -1; j = parent[j]) first[j] = k;
  }
  std::vector<int32_t> maxfirst(n, -1), prevleaf(n, -1), anc(n);
  std::iota(anc.begin(), anc.end(), 0);
  for (int32_t k = 0; k < n; ++k) {
    const int32_t j = post[k];
    if (parent[j] != -1) delta[parent[j]]--;
    for (int64_t e = cptr[j]; e < cptr[j + 1]; ++e) {
      int32_t i = cidx[e];
      if (i <= j) continue;
      if (first[j] <= maxfirst[i]) continue;  // j is not a leaf of row subtree i
      maxfirst[i] = first[j];
      int32_t jprev = prevleaf[i];
      prevleaf[i] = j;
      delta[j]++;
      if (jprev != -1) {
        int32_t q = find_root(anc, jprev);
        delta[q]--;
      }
    }
    if (parent[j] != -1) anc[j] = parent[j];
  }
  cc.resize(n);
  for (int32_t k = 0; k < n; ++k) {
    const int32_t j = post[k];
    cc[j] = (int32_t)delta[j];
    if (parent[j] != -1) delta[parent[j]] += delta[j];
  }
}

}  // namespace

// Fill statistics of the Cholesky factor of a symmetric pattern under a given permutation (perm[new] = old):
// elimination tree + skeleton column counts only (no supernodes, no schedules).  Used by the ordering study and by
// the nested-dissection code to compare candidate orderings.
void fill_count(int32_t n, const int64_t* g_ptr, const int32_t* g_idx, const int32_t* perm, int64_t* nnzL, double* flops,
                int32_t* max_cc, int32_t* colcount_out) {
  std::vector<int32_t> iperm(n);
  for (int32_t i = 0; i < n; ++i) iperm[perm ? perm[i] : i] = i;
  std::vector<int64_t> cptr(n + 1, 0), rptr;
  std::vector<int32_t> cidx, ridx;
  for (int32_t i = 0; i < n; ++i)
    for (int64_t e = g_ptr[i]; e < g_ptr[i + 1]; ++e) {
      const int32_t j = g_idx[e];
      if (j < 0 || j >= n || j >= i) continue;  // each undirected edge once (from its larger endpoint)
      cptr[std::min(iperm[i], iperm[j]) + 1]++;
    }
  for (int32_t i = 0; i < n; ++i) cptr[i + 1] += cptr[i];
  cidx.resize(cptr[n]);
  {
    std::vector<int64_t> fill(cptr.begin(), cptr.end() - 1);
    for (int32_t i = 0; i < n; ++i)
      for (int64_t e = g_ptr[i]; e < g_ptr[i + 1]; ++e) {
        const int32_t j = g_idx[e];
        if (j < 0 || j >= n || j >= i) continue;
        const int32_t a = iperm[i], b = iperm[j];
        cidx[fill[std::min(a, b)]++] = std::max(a, b);
      }
  }
  transpose_pattern(n, cptr, cidx, rptr, ridx);
  std::vector<int32_t> parent, post, cc;
  etree(n, rptr, ridx, parent);
  postorder(n, parent, post);
  column_counts(n, parent, post, cptr, cidx, cc);
  int64_t nz = 0;
  double fl = 0;
  int32_t mx = 0;
  for (int32_t j = 0; j < n; ++j) {
    nz += cc[j];
    fl += (double)cc[j] * (double)cc[j];
    mx = std::max(mx, cc[j]);
  }
  if (nnzL) *nnzL = nz;
  if (flops) *flops = fl;
  if (max_cc) *max_cc = mx;
  if (colcount_out) std::memcpy(colcount_out, cc.data(), sizeof(int32_t) * (size_t)n);
}

Symbolic* symbolic_analyze(int32_t n, int32_t K, const int64_t* const* indptr, const int32_t* const* indices,
                           const int32_t* perm_in, const SymbolicOptions& opts) {
  use_host_threads();
  Symbolic* S = new Symbolic();
  S->n = n;
  S->K = K;
  const bool verbose = getenv("SCILMM_VERBOSE") != nullptr;
  auto tlast = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    auto now = std::chrono::steady_clock::now();
    if (verbose) fprintf(stderr, "[scilmm symbolic] %-28s %8.3f s\n", what, std::chrono::duration<double>(now - tlast).count());
    tlast = now;
  };
  // ---------------------------------------------------------------- 1. symmetric adjacency of the union pattern
  // G = the union pattern without its diagonal, both halves, lists ascending: every later step of the analysis (the
  // ordering, the elimination tree, the permuted pattern) reads it by vertex, with no scatter pass of its own.
  S->is_diag.assign(K, 1);
  for (int32_t k = 0; k < K; ++k) {
    bool diag = true;
    for (int32_t i = 0; i < n && diag; ++i)
      for (int64_t e = indptr[k][i]; e < indptr[k][i + 1]; ++e)
        if (indices[k][e] != i) { diag = false; break; }
    S->is_diag[k] = diag ? 1 : 0;
  }
  std::vector<int64_t> gptr(n + 1, 0);
  std::vector<int32_t> gidx;
  // Fast path, inputs stored with both halves (what SciPy hands over): row i of G is the merged row i of the inputs,
  // one pass over the rows on all cores.  Structural symmetry is verified by comparing an order-independent 64-bit
  // hash sum of the lower entries (i, j) with that of the mirrored upper entries; inputs that store one half only, or
  // unsymmetric ones, take the scatter path below, which reads the lower half alone.
  bool have_g = false;
  {
    std::vector<int64_t> ub(n + 1, 0);
    for (int32_t i = 0; i < n; ++i) {
      int64_t len = 0;
      for (int32_t k = 0; k < K; ++k) len += indptr[k][i + 1] - indptr[k][i];
      ub[i + 1] = ub[i] + len;
    }
    std::vector<int32_t> stage(ub[n]);
    std::vector<int32_t> cnt(n), low(n);
    uint64_t hlo = 0, hup = 0;
    auto mix = [](uint64_t x) {
      x += 0x9e3779b97f4a7c15ull;
      x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
      x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
      return x ^ (x >> 31);
    };
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : hlo, hup)
    for (int32_t i = 0; i < n; ++i) {
      int32_t* r = stage.data() + ub[i];
      int64_t m = 0;
      for (int32_t k = 0; k < K; ++k)
        for (int64_t e = indptr[k][i]; e < indptr[k][i + 1]; ++e) {
          const int32_t j = indices[k][e];
          if (j >= 0 && j < n && j != i) r[m++] = j;
        }
      bool sorted = true;
      for (int64_t t = 1; t < m; ++t)
        if (r[t - 1] >= r[t]) { sorted = false; break; }
      if (!sorted) {
        std::sort(r, r + m);
        m = std::unique(r, r + m) - r;
      }
      int32_t lo = 0;
      for (int64_t t = 0; t < m; ++t) {
        const uint32_t j = (uint32_t)r[t];
        if (r[t] < i) { hlo += mix(((uint64_t)(uint32_t)i << 32) | j); ++lo; }
        else hup += mix(((uint64_t)j << 32) | (uint32_t)i);
      }
      cnt[i] = (int32_t)m;
      low[i] = lo;
    }
    int64_t nlow = 0, nall = 0;
    for (int32_t i = 0; i < n; ++i) { nlow += low[i]; nall += cnt[i]; }
    if (hlo == hup && nall == 2 * nlow) {
      for (int32_t i = 0; i < n; ++i) gptr[i + 1] = gptr[i] + cnt[i];
      gidx.resize(gptr[n]);
#pragma omp parallel for schedule(dynamic, 4096)
      for (int32_t i = 0; i < n; ++i) std::copy(stage.data() + ub[i], stage.data() + ub[i] + cnt[i], gidx.begin() + gptr[i]);
      S->nnz_pattern = nlow + n;
      have_g = true;
    }
  }
  if (verbose) fprintf(stderr, "[scilmm symbolic] inputs %s\n", have_g ? "store both halves: adjacency read row by row" : "do not store both halves symmetrically: lower half scattered");
  if (!have_g) {
  std::vector<int64_t> uptr(n + 1, 0);
  std::vector<int32_t> uidx;
  {
    std::vector<int64_t> cnt(n, 0);
#pragma omp parallel for schedule(dynamic, 1024)
    for (int32_t i = 0; i < n; ++i) {
      std::vector<int32_t> tmp;
      tmp.push_back(i);
      for (int32_t k = 0; k < K; ++k)
        for (int64_t e = indptr[k][i]; e < indptr[k][i + 1]; ++e) {
          int32_t j = indices[k][e];
          if (j < 0 || j >= n) continue;
          if (j <= i) tmp.push_back(j);
        }
      std::sort(tmp.begin(), tmp.end());
      cnt[i] = std::unique(tmp.begin(), tmp.end()) - tmp.begin();
    }
    for (int32_t i = 0; i < n; ++i) uptr[i + 1] = uptr[i] + cnt[i];
    uidx.resize(uptr[n]);
#pragma omp parallel for schedule(dynamic, 1024)
    for (int32_t i = 0; i < n; ++i) {
      std::vector<int32_t> tmp;
      tmp.push_back(i);
      for (int32_t k = 0; k < K; ++k)
        for (int64_t e = indptr[k][i]; e < indptr[k][i + 1]; ++e) {
          int32_t j = indices[k][e];
          if (j < 0 || j >= n) continue;
          if (j <= i) tmp.push_back(j);
        }
      std::sort(tmp.begin(), tmp.end());
      tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
      std::copy(tmp.begin(), tmp.end(), uidx.begin() + uptr[i]);
    }
  }
  S->nnz_pattern = uptr[n];

  // G = the union pattern without its diagonal, both halves, lists ascending: every later step of the analysis (the
  // ordering, the elimination tree, the permuted pattern) reads it by vertex, with no scatter pass of its own.
  // Built on all cores: the lower half of a vertex is its own row; the upper half is counted and filled with relaxed
  // atomics and then sorted, so the result does not depend on the thread schedule.
  {
    std::vector<int64_t> up(n, 0);
#pragma omp parallel for schedule(dynamic, 4096) num_threads(BUCKET_THREADS)
    for (int32_t i = 0; i < n; ++i)
      for (int64_t e = uptr[i]; e < uptr[i + 1]; ++e) {
        const int32_t j = uidx[e];
        if (j != i) __atomic_fetch_add(&up[j], 1, __ATOMIC_RELAXED);
      }
    for (int32_t i = 0; i < n; ++i) gptr[i + 1] = gptr[i] + (uptr[i + 1] - uptr[i] - 1) + up[i];
    gidx.resize(gptr[n]);
    std::vector<int64_t> fill(n);
#pragma omp parallel for schedule(dynamic, 4096)
    for (int32_t i = 0; i < n; ++i) {
      int64_t f = gptr[i];
      for (int64_t e = uptr[i]; e < uptr[i + 1]; ++e)
        if (uidx[e] != i) gidx[f++] = uidx[e];
      fill[i] = f;
    }
#pragma omp parallel for schedule(dynamic, 4096) num_threads(BUCKET_THREADS)
    for (int32_t i = 0; i < n; ++i)
      for (int64_t e = uptr[i]; e < uptr[i + 1]; ++e) {
        const int32_t j = uidx[e];
        if (j != i) gidx[__atomic_fetch_add(&fill[j], 1, __ATOMIC_RELAXED)] = i;
      }
#pragma omp parallel for schedule(dynamic, 1024)
    for (int32_t i = 0; i < n; ++i) std::sort(gidx.begin() + (gptr[i + 1] - up[i]), gidx.begin() + gptr[i + 1]);
    std::vector<int32_t>().swap(uidx);
  }
  }
  lap("symmetric adjacency");
  std::vector<int32_t> perm(n);
  if (opts.ordering == 2) {
    if (!perm_in) { S->error = "user ordering requested but no permutation given"; return S; }
    std::vector<uint8_t> seen(n, 0);
    for (int32_t i = 0; i < n; ++i) {
      int32_t p = perm_in[i];
      if (p < 0 || p >= n || seen[p]) { S->error = "invalid user permutation"; return S; }
      seen[p] = 1;
      perm[i] = p;
    }
  } else if (opts.ordering == 1) {
    std::iota(perm.begin(), perm.end(), 0);
  } else if (opts.ordering == 3 || opts.ordering == 4) {
    NdOptions ndo;
    ndo.oksep = opts.nd_oksep;
    NdStats nds;
    nd_order(n, gptr.data(), gidx.data(), perm.data(), ndo, &nds);
    if (verbose)
      fprintf(stderr, "[scilmm symbolic] nested dissection: %d -> %d compressed vertices, %lld separators, largest %lld\n", n,
              nds.n_compressed, (long long)nds.n_separators, (long long)nds.top_separator);
    if (opts.ordering == 4) {
      // keep whichever ordering gives fewer factor flops (elimination tree + column counts only: cheap)
      std::vector<int32_t> p2(n);
      amd_order(n, gptr.data(), gidx.data(), p2.data(), opts.amd_dense);
      double f_nd = 0, f_amd = 0;
      fill_count(n, gptr.data(), gidx.data(), perm.data(), nullptr, &f_nd, nullptr, nullptr);
      fill_count(n, gptr.data(), gidx.data(), p2.data(), nullptr, &f_amd, nullptr, nullptr);
      if (verbose) fprintf(stderr, "[scilmm symbolic] factor flops: nested dissection %.4g, minimum degree %.4g\n", f_nd, f_amd);
      if (f_amd < f_nd) perm.swap(p2);
    }
  } else {
    amd_order(n, gptr.data(), gidx.data(), perm.data(), opts.amd_dense);
  }
  std::vector<int32_t> iperm(n);
  for (int32_t i = 0; i < n; ++i) iperm[perm[i]] = i;

  lap("ordering");
  // ---------------------------------------------------------------- 3. etree + postorder straight from G
  // Liu's algorithm walks the rows of the permuted matrix in order; row i' is vertex perm[i'] and its entries left of
  // the diagonal are the neighbours with a smaller new label -- no permuted copy of the pattern is needed for it.
  // The neighbour lists are filtered (smaller new label only) and relabelled on all cores, a block of rows at a time;
  // the sequential part reads the result as a stream.  nlarger[v] = neighbours with a larger label: the size of v's
  // column in the permuted pattern (unchanged by the postorder below -- adjacent vertices are ancestor and descendant).
  std::vector<int32_t> parent(n, -1);
  std::vector<int32_t> nlarger(n);
  auto etree_from_g = [&](const std::vector<int32_t>& perm, const std::vector<int32_t>& iperm, std::vector<int32_t>& parent,
                          std::vector<int32_t>& nlarger) {
    parent.assign(n, -1);
    nlarger.assign(n, 0);
    std::vector<int32_t> anc(n, -1);
    constexpr int32_t BLK = 32768;
    const int32_t nblk = (n + BLK - 1) / BLK;
    std::vector<int32_t> buf[2], bcnt[2];
    std::vector<int64_t> boff[2];
    for (int h = 0; h < 2; ++h) { bcnt[h].resize(BLK); boff[h].resize(BLK + 1); }
    auto filter_row = [&](int h, int32_t i0, int32_t i) {
      const int32_t v = perm[i];
      int32_t* o = buf[h].data() + boff[h][i - i0];
      int32_t m = 0;
      for (int64_t e = gptr[v]; e < gptr[v + 1]; ++e) {
        const int32_t k = iperm[gidx[e]];
        if (k < i) o[m++] = k;
      }
      bcnt[h][i - i0] = m;
      nlarger[v] = (int32_t)(gptr[v + 1] - gptr[v]) - m;
    };
    auto consume = [&](int h, int32_t i0, int32_t i1) {
      for (int32_t i = i0; i < i1; ++i) {
        const int32_t* o = buf[h].data() + boff[h][i - i0];
        for (int32_t t = 0; t < bcnt[h][i - i0]; ++t) {
          int32_t k = o[t];
          while (k != -1 && k < i) {
            const int32_t nx = anc[k];
            anc[k] = i;
            if (nx == -1) parent[k] = i;
            k = nx;
          }
        }
      }
    };
    // block b is consumed by one thread while the others filter block b + 1
    for (int32_t b = -1; b < nblk; ++b) {
      const int hn = (b + 1) & 1;
      const int32_t n0 = (b + 1) * BLK, n1 = std::min<int64_t>(n, (int64_t)(b + 2) * BLK);
      if (b + 1 < nblk) {
        boff[hn][0] = 0;
        for (int32_t i = n0; i < n1; ++i) boff[hn][i - n0 + 1] = boff[hn][i - n0] + (gptr[perm[i] + 1] - gptr[perm[i]]);
        if ((int64_t)buf[hn].size() < boff[hn][n1 - n0]) buf[hn].resize(boff[hn][n1 - n0]);
      }
      std::atomic<int32_t> next{n0};
      bool consumed = false;
#pragma omp parallel
      {
#ifdef _OPENMP
        const bool consumer = omp_get_thread_num() == 0 && omp_get_num_threads() > 1;
#else
        const bool consumer = false;
#endif
        if (consumer) {
          if (b >= 0) consume(b & 1, b * BLK, std::min<int64_t>(n, (int64_t)(b + 1) * BLK));
          consumed = true;
        } else if (b + 1 < nblk) {
          for (;;) {
            const int32_t i = next.fetch_add(64, std::memory_order_relaxed);
            if (i >= n1) break;
            for (int32_t q = i; q < std::min(n1, i + 64); ++q) filter_row(hn, n0, q);
          }
        }
      }
      if (!consumed && b >= 0) consume(b & 1, b * BLK, std::min<int64_t>(n, (int64_t)(b + 1) * BLK));  // team of one
    }
  };
  etree_from_g(perm, iperm, parent, nlarger);
  std::vector<int32_t> post;
  postorder(n, parent, post);
  // A user-supplied permutation is honoured exactly (parity with an oracle factor of the same P);
  // otherwise the ordering is composed with the etree postorder (same fill, contiguous supernodes).
  if (opts.ordering != 2) {
    std::vector<int32_t> perm2(n), pinv(n);
    for (int32_t k = 0; k < n; ++k) {
      perm2[k] = perm[post[k]];
      pinv[post[k]] = k;
    }
    std::vector<int32_t> par2(n);
    for (int32_t j = 0; j < n; ++j) par2[pinv[j]] = parent[j] == -1 ? -1 : pinv[parent[j]];
    parent.swap(par2);
    perm.swap(perm2);
    for (int32_t i = 0; i < n; ++i) iperm[perm[i]] = i;
    std::iota(post.begin(), post.end(), 0);
  }
  lap("etree+postorder");
  // ---------------------------------------------------------------- 4. permuted strict-lower pattern by column
  // column c = vertex perm[c]; its rows are the neighbours with a larger new label, sorted (column by column, no scatter)
  std::vector<int64_t> cptr(n + 1, 0);
  std::vector<int32_t> cidx;
  {
    for (int32_t c = 0; c < n; ++c) cptr[c + 1] = cptr[c] + nlarger[perm[c]];
    cidx.resize(cptr[n]);
#pragma omp parallel for schedule(dynamic, 1024)
    for (int32_t c = 0; c < n; ++c) {
      const int32_t v = perm[c];
      int64_t f = cptr[c];
      for (int64_t e = gptr[v]; e < gptr[v + 1]; ++e) {
        const int32_t r = iperm[gidx[e]];
        if (r > c) cidx[f++] = r;
      }
      std::sort(cidx.begin() + cptr[c], cidx.begin() + cptr[c + 1]);
    }
  }
  lap("permuted pattern");
  // ---------------------------------------------------------------- 5. column counts
  std::vector<int32_t> cc;
  column_counts(n, parent, post, cptr, cidx, cc);
  S->nnzL = 0;
  S->flops = 0;
  for (int32_t j = 0; j < n; ++j) {
    S->nnzL += cc[j];
    S->flops += (double)cc[j] * (double)cc[j];
  }

  lap("column counts");
  // ---------------------------------------------------------------- 6. supernodes (fundamental, then relaxed)
  struct SN { int32_t start, end, m; int64_t zeros; };
  std::vector<SN> out;
  {
    int32_t j = 0;
    std::vector<int32_t> nchild(n, 0);
    for (int32_t c = 0; c < n; ++c)
      if (parent[c] != -1) nchild[parent[c]]++;
    while (j < n) {
      int32_t s = j;
      while (j + 1 < n && parent[j] == j + 1 && cc[j + 1] == cc[j] - 1 && nchild[j + 1] == 1) ++j;
      ++j;
      SN p{s, j, cc[s], 0};
      // relaxed amalgamation with the immediately preceding supernode while it is a child
      while (!out.empty()) {
        SN& c = out.back();
        int32_t pc = parent[c.end - 1];
        if (pc < p.start || pc >= p.end) break;
        int64_t wc = c.end - c.start, wp = p.end - p.start;
        int64_t mnew = wc + p.m;
        int64_t wnew = wc + wp;
        int64_t z = c.zeros + p.zeros + wc * (wc + p.m - c.m);
        double tot = (double)wnew * (double)mnew - (double)wnew * (double)(wnew - 1) / 2.0;
        double frac = (double)z / tot;
        bool merge = (wnew <= opts.relax_small) || (wnew <= opts.relax_w1 && frac < opts.relax_z1) ||
                     (wnew <= opts.relax_w2 && frac < opts.relax_z2) || (frac < opts.relax_z3);
        // no width cap here: a chain that merges without (much) fill is merged whole and then cut into blocks of
        // exactly max_width columns below, so that the update kernel's 128x128 tiles are full in dense regions
        if (!merge) break;
        p.start = c.start;
        p.m = (int32_t)mnew;
        p.zeros = z;
        out.pop_back();
      }
      out.push_back(p);
    }
  }
  // optional splitting of very wide supernodes (keeps kernels' LDS tiles bounded)
  if (opts.max_width > 0) {
    std::vector<SN> sp;
    for (auto& s : out) {
      int32_t w = s.end - s.start;
      if (w <= opts.max_width) { sp.push_back(s); continue; }
      for (int32_t st = s.start; st < s.end; st += opts.max_width) {
        const int32_t ww = std::min<int32_t>(opts.max_width, s.end - st);
        sp.push_back(SN{st, st + ww, s.m - (st - s.start), 0});
      }
    }
    out.swap(sp);
  }
  int32_t ns = (int32_t)out.size();
  S->nsuper = ns;
  S->sn_start.resize(ns + 1);
  for (int32_t s = 0; s < ns; ++s) S->sn_start[s] = out[s].start;
  S->sn_start[ns] = n;
  std::vector<int32_t> snode_of(n);
  for (int32_t s = 0; s < ns; ++s)
    for (int32_t j = out[s].start; j < out[s].end; ++j) snode_of[j] = s;

  lap("supernodes");
  // ---------------------------------------------------------------- 7. row structure of every supernode
  S->sn_rowptr.assign(ns + 1, 0);
  S->sn_parent.assign(ns, -1);
  {
    // rows contributed by the input pattern itself (the columns of A inside the front), on all cores; what is left
    // for the sequential sweep below is the merge of the children's row lists (a child always precedes its parent)
    std::vector<std::vector<int32_t>> arows(ns);
#pragma omp parallel
    {
      std::vector<int32_t> mk(n, -1);
#pragma omp for schedule(dynamic, 64)
      for (int32_t s = 0; s < ns; ++s) {
        const int32_t c0 = out[s].start, c1 = out[s].end;
        std::vector<int32_t>& a = arows[s];
        for (int32_t j = c0; j < c1; ++j)
          for (int64_t e = cptr[j]; e < cptr[j + 1]; ++e) {
            const int32_t i = cidx[e];
            if (i >= c1 && mk[i] != s) { mk[i] = s; a.push_back(i); }
          }
      }
    }
    std::vector<int32_t> chead(ns, -1), cnext(ns, -1);
    std::vector<int32_t> mark(n, -1);
    std::vector<int32_t> tmp;
    S->sn_rows.reserve((size_t)(S->nnzL / 4 + n));
    for (int32_t s = 0; s < ns; ++s) {
      int32_t c0 = out[s].start, c1 = out[s].end;
      tmp.swap(arows[s]);
      std::vector<int32_t>().swap(arows[s]);
      for (int32_t i : tmp) mark[i] = s;
      for (int32_t c = chead[s]; c != -1; c = cnext[c]) {
        int64_t b = S->sn_rowptr[c], e2 = S->sn_rowptr[c + 1];
        int32_t wc = out[c].end - out[c].start;
        for (int64_t e = b + wc; e < e2; ++e) {
          int32_t i = S->sn_rows[e];
          if (i >= c1 && mark[i] != s) { mark[i] = s; tmp.push_back(i); }
        }
      }
      for (int32_t j = c0; j < c1; ++j) S->sn_rows.push_back(j);
      const size_t first = S->sn_rows.size();
      if ((int64_t)tmp.size() * 16 > (int64_t)(n - c1)) {
        // long list (the fronts of the trailing clique hold most of the later columns): read it off the marks
        for (int32_t i = c1; i < n; ++i)
          if (mark[i] == s) S->sn_rows.push_back(i);
      } else {
        std::sort(tmp.begin(), tmp.end());
        S->sn_rows.insert(S->sn_rows.end(), tmp.begin(), tmp.end());
      }
      S->sn_rowptr[s + 1] = (int64_t)S->sn_rows.size();
      if (!tmp.empty()) {
        int32_t p = snode_of[S->sn_rows[first]];
        S->sn_parent[s] = p;
        cnext[s] = chead[p];
        chead[p] = s;
      }
      tmp.clear();
    }
  }
  lap("row structures");
  // ---------------------------------------------------------------- 7a. dense tail: which fronts
  // The top of a pedigree factor (16.6k columns at the 100k config, 170k at 1M; > 75 % / > 99 % of the flops) consists
  // of fronts whose row lists are "almost every later column".  Padding those lists to EVERY later column (explicit
  // zeros, like relaxed amalgamation) turns that part into one dense lower-triangular matrix cut into block columns:
  // updates inside it need no index lists, no descriptors and no gather.
  // The tail T is an ancestor-closed set of fronts, grown from the roots of the supernodal tree downwards: among the
  // fronts whose parent is already in T the one with the longest true row list is taken next, as long as its own list
  // fills at least half of its padded one and the padded flop count of the whole tail stays within dense_relax of the
  // true one.  T need not be a chain of the tree: at the 1M config two chains of near-dense fronts (45 x 128 columns
  // with 126k rows each beside the main one) merge 46 levels above the tail's start; as a side branch their updates
  // of the tail went through the gather path at 11 TFLOP/s (a fifth of the factorization time), inside T they are
  // k_dense work.  The fronts of T are moved to the end of the elimination order (children still precede parents, so
  // the fill is unchanged) in the reverse order in which they were taken.
  // A user-supplied permutation is never changed: then T is the trailing chain (parent = next front) only.
  S->dense_first = ns;
  int32_t best = ns;
  std::thread recount;  // (column counts of the final order, when the tail is moved: joined before the return)
  if (opts.dense_relax > 0.0 && opts.ordering == 2) {
    double fl_dense = 0.0, fl_true = 0.0;
    for (int32_t q = ns - 1; q >= 0; --q) {
      if (q < ns - 1 && S->sn_parent[q] != q + 1) break;
      const double w = out[q].end - out[q].start, mt = (double)(S->sn_rowptr[q + 1] - S->sn_rowptr[q]), md = (double)(n - out[q].start);
      fl_dense += w * md * md;
      fl_true += w * mt * mt;
      if (fl_dense <= opts.dense_relax * fl_true) best = q;
      else if (fl_dense > 1.5 * fl_true) break;
    }
  } else if (opts.dense_relax > 0.0) {
    std::vector<int32_t> chead(ns, -1), cnext(ns, -1);
    for (int32_t q = 0; q < ns; ++q)
      if (S->sn_parent[q] != -1) { cnext[q] = chead[S->sn_parent[q]]; chead[S->sn_parent[q]] = q; }
    std::priority_queue<std::pair<int64_t, int32_t>> heap;  // (true rows, front): longest list first, then the later front
    for (int32_t q = 0; q < ns; ++q)
      if (S->sn_parent[q] == -1) heap.push({S->sn_rowptr[q + 1] - S->sn_rowptr[q], q});
    std::vector<int32_t> taken;
    double fl_dense = 0.0, fl_true = 0.0;
    int64_t cols_after = 0, best_cols = 0, wide_cols = 0;
    double best_ratio = 0.0, wide_ratio = 0.0;
    size_t nbest = 0, nwide = 0;
    const char* tun = getenv("SCILMM_TUNING");
    const char* eel = (tun && tun[0] == '1') ? getenv("SCILMM_TAIL_ELIG") : nullptr;
    const double tail_elig = eel ? atof(eel) : 0.5;
    FILE* tail_dump = getenv("SCILMM_TAIL_DUMP") ? fopen(getenv("SCILMM_TAIL_DUMP"), "w") : nullptr;  // diagnostic: every candidate
    if (tail_dump) fprintf(tail_dump, "taken,front,w,true_rows,padded_rows,fl_dense_before,fl_true_before\n");
    while (!heap.empty()) {
      const int32_t q = heap.top().second;
      heap.pop();
      const double w = out[q].end - out[q].start, mt = (double)(S->sn_rowptr[q + 1] - S->sn_rowptr[q]), md = (double)cols_after + w;
      if (tail_dump) fprintf(tail_dump, "%d,%d,%.0f,%.0f,%.0f,%.6g,%.6g\n", (int)taken.size(), q, w, mt, md, fl_dense, fl_true);
      // (0.5: padding such a front costs at most 4 x its true flops, the ratio between the dense and the gather kernel.
      // SCILMM_TUNING=1 SCILMM_TAIL_ELIG=x: at 1M 0.3 takes 27 more fronts, starts the tail 4 levels earlier -- fewer
      // fronts behind k_outside, 7 x the cells -- and is slower, 30.2 vs 29.6 s.)
      if (mt < tail_elig * md) continue;  // its list only gets relatively shorter as T grows: never eligible again
      fl_dense += w * md * md;
      fl_true += w * mt * mt;
      taken.push_back(q);
      cols_after += (int64_t)w;
      if (fl_dense <= opts.dense_relax * fl_true) { nbest = taken.size(); best_cols = cols_after; best_ratio = fl_dense / fl_true; }
      if (fl_dense <= opts.dense_relax_wide * fl_true) { nwide = taken.size(); wide_cols = cols_after; wide_ratio = fl_dense / fl_true; }
      if (fl_dense > std::max(1.5, opts.dense_relax_wide) * fl_true) break;
      for (int32_t c = chead[q]; c != -1; c = cnext[c]) heap.push({S->sn_rowptr[c + 1] - S->sn_rowptr[c], c});
    }
    if (tail_dump) fclose(tail_dump);
    if (best_cols >= opts.dense_wide_cols && nwide > nbest) { nbest = nwide; best_cols = wide_cols; best_ratio = wide_ratio; }
    if (nbest >= 4) {
      taken.resize(nbest);
      std::reverse(taken.begin(), taken.end());
      best = ns - (int32_t)nbest;
      // The top of T is (all but) a clique: every front's list holds >= 99 % of the later columns (1200 of the 1393
      // fronts at the 1M config, fill 0.997 - 1.000) and is padded to all of them anyway.  The structure of every
      // column BELOW that region depends only on which columns precede it, not on the order inside the region, so the
      // region's columns may be sorted freely: by the number of lower tail fronts that have them as a row.  What a lower
      // front does NOT reach (12 - 50 % of the region for the fronts below it at 1M) then sits together at the region's
      // start, as whole 128-column blocks that the dense update can skip (tail_blk below), instead of being spread over
      // every block as padding.  (The true fill inside the region changes by a fraction of a percent with its order:
      // the column counts are taken again for the final order.)
      size_t ncl = 0;
      {
        int64_t cols = 0;
        for (size_t t = nbest; t-- > 0;) {
          const int32_t q = taken[t];
          const int64_t w = out[q].end - out[q].start;
          if ((double)(S->sn_rowptr[q + 1] - S->sn_rowptr[q]) < 0.99 * (double)(cols + w)) break;
          cols += w;
          ++ncl;
        }
      }
      const size_t nlow = nbest - ncl;
      std::vector<int32_t> clique_cols;  // old labels in their new order
      if (ncl >= 2) {
        std::vector<uint8_t> in_cl(ns, 0);
        for (size_t t = nlow; t < nbest; ++t) in_cl[taken[t]] = 1;
        std::vector<int32_t> cnt(n, 0);
        for (size_t t = 0; t < nlow; ++t) {
          const int32_t q = taken[t];
          for (int64_t e = S->sn_rowptr[q] + (out[q].end - out[q].start); e < S->sn_rowptr[q + 1]; ++e)
            if (in_cl[snode_of[S->sn_rows[e]]]) cnt[S->sn_rows[e]]++;
        }
        for (size_t t = nlow; t < nbest; ++t)
          for (int32_t j = out[taken[t]].start; j < out[taken[t]].end; ++j) clique_cols.push_back(j);
        if (nlow > 0) std::stable_sort(clique_cols.begin(), clique_cols.end(), [&](int32_t x, int32_t y) { return cnt[x] < cnt[y]; });
      }
      // new front order = the others as they are, then T; new labels front by front (clique: column by column)
      std::vector<uint8_t> inT(ns, 0);
      for (int32_t q : taken) inT[q] = 1;
      std::vector<int32_t> order;
      order.reserve(ns);
      for (int32_t q = 0; q < ns; ++q)
        if (!inT[q]) order.push_back(q);
      order.insert(order.end(), taken.begin(), taken.end());
      std::vector<int32_t> newlab(n), newsn(ns);
      std::vector<SN> out2(ns);
      const int32_t k_cl = ncl >= 2 ? ns - (int32_t)ncl : ns;  // first clique front (new index)
      {
        int32_t c = 0;
        size_t ci = 0;
        for (int32_t k = 0; k < ns; ++k) {
          const int32_t q = order[k];
          newsn[q] = k;
          out2[k] = out[q];
          out2[k].start = c;
          const int32_t w = out[q].end - out[q].start;
          if (k >= k_cl) {
            for (int32_t t = 0; t < w; ++t) newlab[clique_cols[ci++]] = c++;
          } else {
            for (int32_t j = out[q].start; j < out[q].end; ++j) newlab[j] = c++;
          }
          out2[k].end = c;
        }
      }
      bool in_place = true;
      for (int32_t j = 0; j < n && in_place; ++j) in_place = newlab[j] == j;
      if (verbose)
        fprintf(stderr, "[scilmm symbolic] dense tail: %zu fronts (%zu of them a clique), %lld columns, padded / true flops %.3f%s\n", nbest, ncl,
                (long long)best_cols, best_ratio, in_place ? "" : " (moved to the end of the order)");
      if (!in_place) {
        std::vector<int32_t> perm2(n);
        for (int32_t j = 0; j < n; ++j) perm2[newlab[j]] = perm[j];
        perm.swap(perm2);
        for (int32_t i = 0; i < n; ++i) iperm[perm[i]] = i;
        // row lists: relabel, sort, store in the new front order (a clique front: every later column)
        std::vector<int64_t> rp2(ns + 1, 0);
        for (int32_t k = 0; k < ns; ++k)
          rp2[k + 1] = rp2[k] + (k >= k_cl ? (int64_t)(n - out2[k].start) : S->sn_rowptr[order[k] + 1] - S->sn_rowptr[order[k]]);
        std::vector<int32_t> rows2(rp2[ns]);
#pragma omp parallel for schedule(dynamic, 64)
        for (int32_t k = 0; k < ns; ++k) {
          int32_t* o = rows2.data() + rp2[k];
          if (k >= k_cl) {
            for (int32_t r = out2[k].start; r < n; ++r) o[r - out2[k].start] = r;
            continue;
          }
          const int32_t q = order[k];
          const int64_t b = S->sn_rowptr[q], e = S->sn_rowptr[q + 1];
          for (int64_t t = b; t < e; ++t) o[t - b] = newlab[S->sn_rows[t]];
          std::sort(o + (out[q].end - out[q].start), o + (e - b));  // (own columns stay first and ascending)
        }
        S->sn_rows.swap(rows2);
        S->sn_rowptr.swap(rp2);
        out.swap(out2);
        for (int32_t k = 0; k < ns; ++k) {
          S->sn_start[k] = out[k].start;
          for (int32_t j = out[k].start; j < out[k].end; ++j) snode_of[j] = k;
        }
        // parents (of fronts and of columns) and column counts follow from the row lists
        for (int32_t k = 0; k < ns; ++k) {
          const int64_t b = S->sn_rowptr[k], e = S->sn_rowptr[k + 1];
          const int32_t w = out[k].end - out[k].start;
          S->sn_parent[k] = e - b > w ? snode_of[S->sn_rows[b + w]] : -1;
          for (int32_t j = out[k].start; j + 1 < out[k].end; ++j) parent[j] = j + 1;
          parent[out[k].end - 1] = e - b > w ? S->sn_rows[b + w] : -1;
        }
        // the permuted pattern under the new labels, again straight from G
#pragma omp parallel for schedule(dynamic, 1024)
        for (int32_t c = 0; c < n; ++c) {
          const int32_t v = perm[c];
          int64_t m = 0;
          for (int64_t e = gptr[v]; e < gptr[v + 1]; ++e) m += iperm[gidx[e]] > c;
          cptr[c + 1] = m;
        }
        cptr[0] = 0;
        for (int32_t c = 0; c < n; ++c) cptr[c + 1] += cptr[c];
#pragma omp parallel for schedule(dynamic, 1024)
        for (int32_t c = 0; c < n; ++c) {
          const int32_t v = perm[c];
          int64_t f = cptr[c];
          for (int64_t e = gptr[v]; e < gptr[v + 1]; ++e) {
            const int32_t r = iperm[gidx[e]];
            if (r > c) cidx[f++] = r;
          }
          std::sort(cidx.begin() + cptr[c], cidx.begin() + cptr[c + 1]);
        }
        // true column counts (nnz(L), flops) of the final order: elimination tree + postorder + skeleton counts again.
        // Nothing below reads them (they are reported numbers), so the recount runs BESIDE the rest of the analysis on a
        // quarter of the host threads and is joined before the function returns: 0.4 of the 1.7 s analysis of the 100k
        // config, 2 of 20 s at 1M.  It only reads perm / iperm / G / the permuted pattern, none of which changes any more.
        recount = std::thread([&]() {
#ifdef _OPENMP
          omp_set_num_threads(std::max(1, host_threads() / 4));
#endif
          std::vector<int32_t> tpar, tnl, tpost;
          etree_from_g(perm, iperm, tpar, tnl);
          postorder(n, tpar, tpost);
          column_counts(n, tpar, tpost, cptr, cidx, cc);
        });
      }
    }
  }
  S->perm = perm;
  S->iperm = iperm;
  S->parent = parent;
  lap("dense tail selection");
  // ---------------------------------------------------------------- 7b. dense tail: padding
  if (opts.dense_relax > 0.0) {
    if (ns - best >= 4) {
      // algorithmic update flops of the fronts about to be padded, on their TRUE row lists (same formula as step 10)
      double true_tail = 0.0;
      for (int32_t d = 0; d < ns; ++d) {
        const int64_t rb = S->sn_rowptr[d], re = S->sn_rowptr[d + 1];
        const int32_t w = out[d].end - out[d].start;
        int64_t t = rb + w;
        while (t < re) {
          const int32_t sq = snode_of[S->sn_rows[t]];
          int64_t t2 = t;
          while (t2 < re && snode_of[S->sn_rows[t2]] == sq) ++t2;
          if (d >= best) {
            const double nq = (double)(t2 - t), below = (double)(re - t2);
            true_tail += (double)w * (nq * (nq + 1.0) + 2.0 * nq * below);
          }
          t = t2;
        }
      }
      // block pattern of the true structure (which later tail fronts does a tail front reach at all)
      S->tail_blk_ptr.assign((size_t)(ns - best) + 1, 0);
      for (int32_t q = best; q < ns; ++q) {
        const int64_t rb = S->sn_rowptr[q] + (out[q].end - out[q].start), re = S->sn_rowptr[q + 1];
        int32_t last = -1;
        for (int64_t t = rb; t < re; ++t) {
          const int32_t f = snode_of[S->sn_rows[t]];  // rows ascending => fronts ascending
          if (f != last) { S->tail_blk.push_back(f - best); last = f; }
        }
        S->tail_blk_ptr[(size_t)(q - best) + 1] = (int64_t)S->tail_blk.size();
      }
      S->update_flops_pad = -true_tail;  // completed in step 10: executed(tail) - true(tail)
      S->dense_flops = true_tail;
      S->dense_first = best;
      S->sn_rows.resize((size_t)S->sn_rowptr[best]);
      for (int32_t q = best; q < ns; ++q) {
        for (int32_t r = out[q].start; r < n; ++r) S->sn_rows.push_back(r);
        S->sn_rowptr[q + 1] = (int64_t)S->sn_rows.size();
        S->sn_parent[q] = q + 1 < ns ? q + 1 : -1;  // padded: the tail is a chain in index order
      }
    }
  }
  lap("dense tail padding");
  // ---------------------------------------------------------------- 8. panel offsets, levels
  S->sn_loff.assign(ns + 1, 0);
  for (int32_t s = 0; s < ns; ++s) {
    int64_t m = S->sn_rowptr[s + 1] - S->sn_rowptr[s];
    int64_t w = out[s].end - out[s].start;
    int64_t sz = m * w;
    sz = (sz + 1) & ~(int64_t)1;  // keep every panel 16-byte aligned
    S->sn_loff[s + 1] = S->sn_loff[s] + sz;
  }
  S->nnzL_stored = S->sn_loff[ns];
  S->sn_level.assign(ns, 0);
  for (int32_t s = 0; s < ns; ++s) {
    int32_t p = S->sn_parent[s];
    if (p != -1) S->sn_level[p] = std::max(S->sn_level[p], S->sn_level[s] + 1);
  }
  // (SCILMM_TUNING=1 SCILMM_TAIL_DELAY=k starts the tail chain k levels later, so that more of the prelude lies below
  // it and goes through k_outside instead of the gather path: at 1M k = 24 moves 96 % of the remaining gather combos
  // there, 2.8M -> 6.5M outside items, and the factorization takes the same 29.6 s -- the two paths cost the same.)
  if (const char* e = (getenv("SCILMM_TUNING") && getenv("SCILMM_TUNING")[0] == '1') ? getenv("SCILMM_TAIL_DELAY") : nullptr) {
    if (S->dense_first < ns) {
      S->sn_level[S->dense_first] += atoi(e);
      for (int32_t q = S->dense_first + 1; q < ns; ++q) S->sn_level[q] = std::max(S->sn_level[q], S->sn_level[q - 1] + 1);
    }
  }
  S->nlevels = 0;
  for (int32_t s = 0; s < ns; ++s) S->nlevels = std::max(S->nlevels, S->sn_level[s] + 1);

  // children lists (increasing order)
  S->child_ptr.assign(ns + 1, 0);
  for (int32_t s = 0; s < ns; ++s)
    if (S->sn_parent[s] != -1) S->child_ptr[S->sn_parent[s] + 1]++;
  for (int32_t s = 0; s < ns; ++s) S->child_ptr[s + 1] += S->child_ptr[s];
  S->child_idx.resize(S->child_ptr[ns]);
  {
    std::vector<int64_t> fill(S->child_ptr.begin(), S->child_ptr.end() - 1);
    for (int32_t s = 0; s < ns; ++s)
      if (S->sn_parent[s] != -1) S->child_idx[fill[S->sn_parent[s]]++] = s;
  }

  lap("offsets/levels/children");
  // ---------------------------------------------------------------- 9. value-assembly maps
  // pattern slots are numbered in permuted CSC order with the diagonal first in each column.
  {
    std::vector<int64_t> slot_ptr(n + 1, 0);
    for (int32_t j = 0; j < n; ++j) slot_ptr[j + 1] = slot_ptr[j] + 1 + (cptr[j + 1] - cptr[j]);
    S->asm_dst.resize(slot_ptr[n]);
    S->diag_dst.resize(n);
    S->pat_colptr = slot_ptr;
    S->pat_row.resize(slot_ptr[n]);
#pragma omp parallel for schedule(dynamic, 4096)
    for (int32_t j = 0; j < n; ++j) {
      int64_t sl = slot_ptr[j];
      S->pat_row[sl++] = j;
      for (int64_t e = cptr[j]; e < cptr[j + 1]; ++e) S->pat_row[sl++] = cidx[e];
    }
    S->inv_off.assign(ns + 1, 0);
    for (int32_t s = 0; s < ns; ++s) {
      int64_t w = out[s].end - out[s].start;
      S->inv_off[s + 1] = S->inv_off[s] + ((w * w + 1) & ~(int64_t)1);
    }
#pragma omp parallel
    {
      std::vector<int32_t> pos(n, -1);
#pragma omp for schedule(dynamic, 16)
      for (int32_t s = 0; s < ns; ++s) {
        int64_t rb = S->sn_rowptr[s], re = S->sn_rowptr[s + 1];
        int64_t m = re - rb;
        const bool tail = s >= S->dense_first;  // rows = every column from the front's first on
        const int32_t c0 = out[s].start;
        if (!tail)
          for (int64_t t = rb; t < re; ++t) pos[S->sn_rows[t]] = (int32_t)(t - rb);
        for (int32_t j = c0; j < out[s].end; ++j) {
          int64_t colbase = S->sn_loff[s] + (int64_t)(j - c0) * m;
          int64_t sl = slot_ptr[j];
          S->asm_dst[sl] = colbase + (j - c0);
          S->diag_dst[j] = S->asm_dst[sl];
          ++sl;
          if (tail)
            for (int64_t e = cptr[j]; e < cptr[j + 1]; ++e) S->asm_dst[sl++] = colbase + (cidx[e] - c0);
          else
            for (int64_t e = cptr[j]; e < cptr[j + 1]; ++e) S->asm_dst[sl++] = colbase + pos[cidx[e]];
        }
      }
    }
    lap("  pattern slots -> panels");
    // per input matrix: where does each stored lower entry go
    S->val_slot.resize(K);
    S->val_src.resize(K);
    for (int32_t k = 0; k < K; ++k) {
      if (S->is_diag[k]) {
        for (int32_t i = 0; i < n; ++i)
          for (int64_t e = indptr[k][i]; e < indptr[k][i + 1]; ++e) {
            S->val_slot[k].push_back(iperm[i]);
            S->val_src[k].push_back(e);
          }
        continue;
      }
      // Fast path -- matrix k stores both halves, rows strictly ascending (canonical CSR).  Vertex v's pattern column
      // c = new(v) is walked once: positions of its rows go to a thread-local table, and every entry (v, u) of row v
      // with new(u) >= c finds its slot there -- the column is warm, nothing is searched.  The value is read from the
      // stored LOWER entry: (v, u) itself if u <= v, else its mirror (u, v), whose index inside row u is the number of
      // smaller columns in that row = the running count of mirrors seen while the rows are swept in ascending order
      // (done per range of target rows, one range per thread, so the counts need no atomics).
      bool fast = true;
      {
        uint64_t hlo = 0, hup = 0;
        int64_t bad = 0;
        auto mix = [](uint64_t x) {
          x += 0x9e3779b97f4a7c15ull;
          x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
          x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
          return x ^ (x >> 31);
        };
#pragma omp parallel for schedule(dynamic, 1024) reduction(+ : hlo, hup, bad)
        for (int32_t i = 0; i < n; ++i)
          for (int64_t e = indptr[k][i]; e < indptr[k][i + 1]; ++e) {
            const int32_t j = indices[k][e];
            if (j < 0 || j >= n || (e > indptr[k][i] && indices[k][e - 1] >= j)) { ++bad; continue; }
            if (j < i) hlo += mix(((uint64_t)(uint32_t)i << 32) | (uint32_t)j);
            else if (j > i) hup += mix(((uint64_t)(uint32_t)j << 32) | (uint32_t)i);
          }
        fast = bad == 0 && hlo == hup;
      }
      if (fast) {
        const int64_t* ip = indptr[k];
        const int32_t* ix = indices[k];
        const int64_t nzk = ip[n];
        // mirror positions of the upper entries
        std::vector<int32_t> mir(nzk);
        {
          const int R = std::max(1, host_threads());
          std::vector<int64_t> lowcum(n + 1, 0);
#pragma omp parallel for schedule(static)
          for (int32_t i = 0; i < n; ++i) lowcum[i + 1] = std::lower_bound(ix + ip[i], ix + ip[i + 1], i) - (ix + ip[i]);
          for (int32_t i = 0; i < n; ++i) lowcum[i + 1] += lowcum[i];
          std::vector<int32_t> cut(R + 1, n);
          cut[0] = 0;
          for (int q = 1; q < R; ++q)
            cut[q] = (int32_t)(std::lower_bound(lowcum.begin(), lowcum.end(), lowcum[n] * q / R) - lowcum.begin());
          for (int q = 1; q <= R; ++q) cut[q] = std::max(cut[q], cut[q - 1]);
#pragma omp parallel for schedule(dynamic, 1)
          for (int q = 0; q < R; ++q) {
            const int32_t i0 = cut[q], i1 = cut[q + 1];
            if (i1 <= i0) continue;
            std::vector<int32_t> cur(i1 - i0, 0);
            for (int32_t j = 0; j < i1; ++j) {  // rows j >= i1 have no upper entry below i1
              const int32_t* b = ix + ip[j];
              const int32_t* e = ix + ip[j + 1];
              const int32_t* lo = std::lower_bound(b, e, std::max(i0, j + 1));
              for (const int32_t* t = lo; t < e && *t < i1; ++t) mir[t - ix] = cur[*t - i0]++;
            }
          }
        }
        std::vector<int64_t> optr(n + 1, 0);
#pragma omp parallel for schedule(dynamic, 1024)
        for (int32_t v = 0; v < n; ++v) {
          const int32_t c = iperm[v];
          int64_t m = 0;
          for (int64_t e = ip[v]; e < ip[v + 1]; ++e) m += iperm[ix[e]] >= c;
          optr[v + 1] = m;
        }
        for (int32_t v = 0; v < n; ++v) optr[v + 1] += optr[v];
        S->val_slot[k].resize(optr[n]);
        S->val_src[k].resize(optr[n]);
#pragma omp parallel
        {
          std::vector<int32_t> pos(n, -1);
#pragma omp for schedule(dynamic, 256)
          for (int32_t v = 0; v < n; ++v) {
            const int32_t c = iperm[v];
            for (int64_t e = cptr[c]; e < cptr[c + 1]; ++e) pos[cidx[e]] = (int32_t)(e - cptr[c]);
            int64_t t = optr[v];
            for (int64_t e = ip[v]; e < ip[v + 1]; ++e) {
              const int32_t u = ix[e], r = iperm[u];
              if (r < c) continue;
              S->val_slot[k][t] = r == c ? slot_ptr[c] : slot_ptr[c] + 1 + pos[r];
              S->val_src[k][t] = u <= v ? e : ip[u] + mir[e];
              ++t;
            }
          }
        }
        continue;
      }
      // General inputs (one half stored, unsorted rows, duplicates):
      // every stored lower entry (i, j) looks its pattern slot up: column min(new i, new j), row max, found by bisection
      // in the sorted column -- independent per entry, all cores, no scatter.  (A duplicate of an entry inside one
      // matrix maps to the same slot; the value upload keeps one of them.)
      std::vector<int64_t> lptr(n + 1, 0);
#pragma omp parallel for schedule(static)
      for (int32_t i = 0; i < n; ++i) {
        int64_t c = 0;
        for (int64_t e = indptr[k][i]; e < indptr[k][i + 1]; ++e) c += indices[k][e] >= 0 && indices[k][e] <= i;
        lptr[i + 1] = c;
      }
      for (int32_t i = 0; i < n; ++i) lptr[i + 1] += lptr[i];
      S->val_slot[k].resize(lptr[n]);
      S->val_src[k].resize(lptr[n]);
#pragma omp parallel for schedule(dynamic, 256)
      for (int32_t i = 0; i < n; ++i) {
        int64_t t = lptr[i];
        const int32_t a = iperm[i];
        for (int64_t e = indptr[k][i]; e < indptr[k][i + 1]; ++e) {
          const int32_t j = indices[k][e];
          if (j < 0 || j > i) continue;
          const int32_t bq = iperm[j];
          const int32_t c = std::min(a, bq), r = std::max(a, bq);
          int64_t sl = slot_ptr[c];
          if (r != c) {
            const int32_t* lo = cidx.data() + cptr[c];
            const int32_t* hi = cidx.data() + cptr[c + 1];
            sl += 1 + (std::lower_bound(lo, hi, r) - lo);
          }
          S->val_slot[k][t] = sl;
          S->val_src[k][t] = e;
          ++t;
        }
      }
    }
    S->nnz_pattern = slot_ptr[n];
  }
  lap("  entries -> pattern slots");
  // ---------------------------------------------------------------- 10. left-looking update schedule
  {
    S->upd_ptr.assign(ns + 1, 0);
    for (int32_t d = 0; d < ns; ++d) {
      int64_t rb = S->sn_rowptr[d], re = S->sn_rowptr[d + 1];
      int32_t w = out[d].end - out[d].start;
      int32_t prev = -1;
      for (int64_t t = rb + w; t < re; ++t) {
        int32_t s = snode_of[S->sn_rows[t]];
        if (s != prev) { S->upd_ptr[s + 1]++; prev = s; }
      }
    }
    for (int32_t s = 0; s < ns; ++s) S->upd_ptr[s + 1] += S->upd_ptr[s];
    int64_t nu = S->upd_ptr[ns];
    S->upd_src.resize(nu);
    S->upd_p0.resize(nu);
    S->upd_p1.resize(nu);
    S->upd_jp0.resize(nu);
    std::vector<int64_t> fill(S->upd_ptr.begin(), S->upd_ptr.end() - 1);
    for (int32_t d = 0; d < ns; ++d) {  // increasing d => each target's list is in increasing descendant order
      int64_t rb = S->sn_rowptr[d], re = S->sn_rowptr[d + 1];
      int32_t w = out[d].end - out[d].start;
      int64_t t = rb + w;
      while (t < re) {
        int32_t s = snode_of[S->sn_rows[t]];
        int64_t t2 = t;
        while (t2 < re && snode_of[S->sn_rows[t2]] == s) ++t2;
        int64_t f = fill[s]++;
        S->upd_src[f] = d;
        S->upd_p0[f] = (int32_t)(t - rb);
        S->upd_p1[f] = (int32_t)(t2 - rb);
        S->upd_jp0[f] = (S->sn_rows[t2 - 1] - S->sn_rows[t] == (int32_t)(t2 - 1 - t)) ? S->sn_rows[t] - out[s].start : -1;
        {
          const double nq = (double)(t2 - t), below = (double)(re - t2);
          S->update_flops += (double)w * (nq * (nq + 1.0) + 2.0 * nq * below);
          if (d >= S->dense_first) S->update_flops_pad += (double)w * (nq * (nq + 1.0) + 2.0 * nq * below);
        }
        t = t2;
      }
    }
  }
  // level lists, big fronts first inside a level
  S->level_ptr.assign(S->nlevels + 1, 0);
  for (int32_t s = 0; s < ns; ++s) S->level_ptr[S->sn_level[s] + 1]++;
  for (int32_t l = 0; l < S->nlevels; ++l) S->level_ptr[l + 1] += S->level_ptr[l];
  S->level_fronts.resize(ns);
  {
    std::vector<int32_t> fill(S->level_ptr.begin(), S->level_ptr.end() - 1);
    for (int32_t s = 0; s < ns; ++s) S->level_fronts[fill[S->sn_level[s]]++] = s;
  }
  lap("update schedule");
  // ---------------------------------------------------------------- 11. target tiles (their combos are built lazily)
  {
    const int32_t TM = opts.tile_rows;
    S->tile_rows = TM;
    S->tile_base.assign(ns + 1, 0);
    for (int32_t s = 0; s < ns; ++s) {
      int64_t m = S->sn_rowptr[s + 1] - S->sn_rowptr[s];
      S->tile_base[s + 1] = S->tile_base[s] + (m + TM - 1) / TM;
    }
    int64_t nt = S->tile_base[ns];
    S->tile_front.resize(nt);
    for (int32_t s = 0; s < ns; ++s)
      for (int64_t g = S->tile_base[s]; g < S->tile_base[s + 1]; ++g) S->tile_front[g] = s;
    // per-level tile lists in tile order (the update plan balances its own work items; trsm / L*R tiles cost the same)
    S->level_tile_ptr.assign(S->nlevels + 1, 0);
    for (int64_t g = 0; g < nt; ++g) S->level_tile_ptr[S->sn_level[S->tile_front[g]] + 1]++;
    for (int32_t l = 0; l < S->nlevels; ++l) S->level_tile_ptr[l + 1] += S->level_tile_ptr[l];
    S->level_tiles.resize(nt);
    {
      std::vector<int64_t> fill(S->level_tile_ptr.begin(), S->level_tile_ptr.end() - 1);
      for (int64_t g = 0; g < nt; ++g) S->level_tiles[fill[S->sn_level[S->tile_front[g]]]++] = (int32_t)g;
    }
    // per-level update-pair lists (by target level)
    S->level_pair_ptr.assign(S->nlevels + 1, 0);
    for (int32_t s = 0; s < ns; ++s) S->level_pair_ptr[S->sn_level[s] + 1] += S->upd_ptr[s + 1] - S->upd_ptr[s];
    for (int32_t l = 0; l < S->nlevels; ++l) S->level_pair_ptr[l + 1] += S->level_pair_ptr[l];
    S->level_pairs.resize(S->upd_src.size());
    {
      std::vector<int64_t> fill(S->level_pair_ptr.begin(), S->level_pair_ptr.end() - 1);
      for (int32_t s = 0; s < ns; ++s)
        for (int64_t e = S->upd_ptr[s]; e < S->upd_ptr[s + 1]; ++e) S->level_pairs[fill[S->sn_level[s]]++] = (int32_t)e;
    }
  }
  if (recount.joinable()) {
    recount.join();
    S->nnzL = 0;
    S->flops = 0;
    for (int32_t j = 0; j < n; ++j) {
      S->nnzL += cc[j];
      S->flops += (double)cc[j] * (double)cc[j];
    }
    lap("column counts of the final order (joined)");
  }
  std::vector<int32_t>().swap(gidx);
  S->colcount = cc;
  return S;
}

// Step 11b, on demand: for every 128-row tile of every target panel the list of descendant row ranges ("combos")
// that land in it.  keep_front (optional, [nsuper]) restricts the enumeration to the targets a rank owns in a
// multi-GPU run; the lists of the other tiles stay empty.
void build_tile_combos(Symbolic* S, const uint8_t* keep_front, bool skip_dense, const uint8_t* skip_desc) {
  use_host_threads();
  const int32_t ns = S->nsuper;
  const int32_t TM = S->tile_rows;
  const int64_t nt = S->tile_base[ns];
  {
    // pass 1: count combos per tile, pass 2: fill. Rows of d beyond p0 are merged against rows of s.
    std::vector<int64_t> cnt(nt + 1, 0);
    for (int pass = 0; pass < 2; ++pass) {
      std::vector<int64_t> fill;
      if (pass == 1) {
        for (int64_t g = 0; g < nt; ++g) cnt[g + 1] += cnt[g];
        S->combo_ptr.assign(cnt.begin(), cnt.end());
        S->combo_pair.resize(cnt[nt]);
        S->combo_ta.resize(cnt[nt]);
        S->combo_tb.resize(cnt[nt]);
        S->combo_ip0.resize(cnt[nt]);
        fill.assign(cnt.begin(), cnt.end() - 1);
      }
#pragma omp parallel for schedule(dynamic, 64)
      for (int32_t s = 0; s < ns; ++s) {
        if (keep_front && !keep_front[s]) continue;
        const int32_t* rs = S->sn_rows.data() + S->sn_rowptr[s];
        int64_t ms = S->sn_rowptr[s + 1] - S->sn_rowptr[s];
        for (int64_t e = S->upd_ptr[s]; e < S->upd_ptr[s + 1]; ++e) {
          int32_t d = S->upd_src[e];
          if (skip_dense && s >= S->dense_first && d >= S->dense_first) continue;
          if (skip_desc && s >= S->dense_first && skip_desc[d]) continue;
          const int32_t* rd = S->sn_rows.data() + S->sn_rowptr[d];
          int32_t md = (int32_t)(S->sn_rowptr[d + 1] - S->sn_rowptr[d]);
          int32_t t = S->upd_p0[e];
          int64_t pos = 0;
          while (t < md) {
            // position of rd[t] in rs (exists by construction): gallop from the previous position
            const int32_t* it = std::lower_bound(rs + pos, rs + ms, rd[t]);
            pos = it - rs;
            int64_t tile = pos / TM;
            int64_t tile_end_pos = std::min<int64_t>((tile + 1) * TM, ms);
            int32_t lastlabel = rs[tile_end_pos - 1];
            int32_t t2 = (int32_t)(std::upper_bound(rd + t, rd + md, lastlabel) - rd);
            int64_t g = S->tile_base[s] + tile;
            if (pass == 0) {
              cnt[g + 1]++;   // each (s) handled by exactly one thread: no race on its own tiles
            } else {
              int64_t f = fill[g]++;
              S->combo_pair[f] = (int32_t)e;
              S->combo_ta[f] = t;
              S->combo_tb[f] = t2;
              const int64_t pos_last = std::lower_bound(rs + pos, rs + ms, rd[t2 - 1]) - rs;
              S->combo_ip0[f] = (pos_last - pos == (int64_t)(t2 - 1 - t)) ? (int32_t)(pos - tile * TM) : -1;
            }
            t = t2;
          }
        }
      }
    }
  }
  S->combos_built = true;
}

// ------------------------------------------------------------------------------------------------
// On-disk / shared-memory image of an analysis (the reference keeps its stage artefacts on disk,
// scilmm/IBDCompute.py:82-84; here the 20 s analysis of the 1M config need not be redone by every process of a node).
// Plain binary: a header, then every member of Symbolic in the order of the list below -- which is the single place that
// names them, for writing and for reading.  The tile combos (built lazily per rank) are not part of the image.
namespace {
constexpr uint64_t kImageMagic = 0x53434c4d53594d34ull;  // "SCLMSYM4" (4: length + checksum trailer)

// Every byte that passes through pod() / vec() is counted and folded into a 64-bit checksum (four interleaved
// multiply-xorshift lanes over 8-byte words: memory speed); the writer appends (bytes, checksum) as a trailer and the
// reader refuses an image whose trailer does not match what it read -- a truncated or torn file (two writers on one
// name, a copy cut short) is then a cache miss, never a wrong analysis.
struct ImageIO {
  FILE* fp;
  bool write;
  bool ok = true;
  uint64_t bytes = 0;
  uint64_t lane[4] = {0x9e3779b97f4a7c15ull, 0xc2b2ae3d27d4eb4full, 0x165667b19e3779f9ull, 0x27d4eb2f165667c5ull};
  void fold(const void* p, size_t nbytes) {
    const unsigned char* b = (const unsigned char*)p;
    size_t i = 0;
    uint64_t l0 = lane[0], l1 = lane[1], l2 = lane[2], l3 = lane[3];
    for (; i + 32 <= nbytes; i += 32) {
      uint64_t w[4];
      std::memcpy(w, b + i, 32);
      l0 = (l0 ^ w[0]) * 0x9e3779b97f4a7c15ull; l0 ^= l0 >> 29;
      l1 = (l1 ^ w[1]) * 0xc2b2ae3d27d4eb4full; l1 ^= l1 >> 31;
      l2 = (l2 ^ w[2]) * 0x165667b19e3779f9ull; l2 ^= l2 >> 30;
      l3 = (l3 ^ w[3]) * 0x27d4eb2f165667c5ull; l3 ^= l3 >> 28;
    }
    for (; i < nbytes; ++i) { l0 = (l0 ^ b[i]) * 0x100000001b3ull; l0 ^= l0 >> 32; }
    lane[0] = l0; lane[1] = l1; lane[2] = l2; lane[3] = l3;
    bytes += nbytes;
  }
  uint64_t checksum() const {
    uint64_t h = bytes;
    for (int q = 0; q < 4; ++q) { h = (h ^ lane[q]) * 0x9e3779b97f4a7c15ull; h ^= h >> 32; }
    return h;
  }
  template <typename T>
  void pod(T& v) {
    if (!ok) return;
    ok = write ? fwrite(&v, sizeof(T), 1, fp) == 1 : fread(&v, sizeof(T), 1, fp) == 1;
    if (ok) fold(&v, sizeof(T));
  }
  template <typename T>
  void vec(std::vector<T>& v) {
    uint64_t cnt = (uint64_t)v.size();
    pod(cnt);
    if (!ok) return;
    if (!write) {
      if (cnt > ((uint64_t)1 << 40) / sizeof(T)) { ok = false; return; }
      v.resize((size_t)cnt);
    }
    if (cnt) ok = write ? fwrite(v.data(), sizeof(T), (size_t)cnt, fp) == cnt : fread(v.data(), sizeof(T), (size_t)cnt, fp) == cnt;
    if (ok && cnt) fold(v.data(), sizeof(T) * (size_t)cnt);
  }
  template <typename T>
  void vecvec(std::vector<std::vector<T>>& v) {
    uint64_t cnt = (uint64_t)v.size();
    pod(cnt);
    if (!ok) return;
    if (!write) v.resize((size_t)cnt);
    for (auto& x : v) vec(x);
  }
};

void image_fields(ImageIO& io, Symbolic& S) {
  io.pod(S.n); io.pod(S.K); io.pod(S.nsuper); io.pod(S.dense_first); io.pod(S.tile_rows); io.pod(S.nlevels);
  io.pod(S.nnz_pattern); io.pod(S.nnzL); io.pod(S.nnzL_stored); io.pod(S.flops); io.pod(S.update_flops);
  io.pod(S.dense_flops); io.pod(S.update_flops_pad);
  io.vec(S.perm); io.vec(S.iperm); io.vec(S.parent); io.vec(S.colcount); io.vec(S.sn_start); io.vec(S.sn_parent);
  io.vec(S.sn_rowptr); io.vec(S.sn_rows); io.vec(S.sn_loff); io.vec(S.sn_level); io.vec(S.child_ptr); io.vec(S.child_idx);
  io.vec(S.upd_ptr); io.vec(S.upd_src); io.vec(S.upd_p0); io.vec(S.upd_p1); io.vec(S.upd_jp0);
  io.vec(S.tile_base); io.vec(S.tile_front);
  io.vec(S.level_tile_ptr); io.vec(S.level_tiles); io.vec(S.level_pair_ptr); io.vec(S.level_pairs);
  io.vec(S.level_ptr); io.vec(S.level_fronts);
  io.vec(S.asm_dst); io.vec(S.diag_dst); io.vec(S.pat_colptr); io.vec(S.pat_row); io.vec(S.inv_off);
  io.vec(S.tail_blk_ptr); io.vec(S.tail_blk);
  io.vecvec(S.val_slot); io.vecvec(S.val_src); io.vec(S.is_diag);
}
}  // namespace

bool symbolic_save(const Symbolic& S, const char* path, uint64_t key) {
  // a name of this writer's own (pid + clock + address entropy): ranks that miss the cache at the same time each write a
  // complete image and the LAST rename wins -- none truncates a file another one is still writing
  char suffix[96];
  const uint64_t salt = (uint64_t)std::chrono::steady_clock::now().time_since_epoch().count() ^ (uint64_t)(uintptr_t)&S;
  snprintf(suffix, sizeof suffix, ".tmp.%ld.%016llx", (long)getpid(), (unsigned long long)salt);
  const std::string tmp = std::string(path) + suffix;
  FILE* fp = fopen(tmp.c_str(), "wbx");  // (x: fail rather than share a name)
  if (!fp) return false;
  ImageIO io{fp, true};
  uint64_t magic = kImageMagic, k = key, nb = (uint64_t)SCILMM_NB;
  io.pod(magic); io.pod(k); io.pod(nb);
  image_fields(io, const_cast<Symbolic&>(S));
  uint64_t trailer[2] = {io.bytes, io.checksum()};
  const bool wrote = io.ok && fwrite(trailer, sizeof(uint64_t), 2, fp) == 2;
  const bool ok = (fclose(fp) == 0) && wrote;
  if (!ok) { remove(tmp.c_str()); return false; }
  if (rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); return false; }  // readers never see a half-written image
  return true;
}

Symbolic* symbolic_load(const char* path, uint64_t key) {
  FILE* fp = fopen(path, "rb");
  if (!fp) return nullptr;
  ImageIO io{fp, false};
  uint64_t magic = 0, k = 0, nb = 0;
  io.pod(magic); io.pod(k); io.pod(nb);
  if (!io.ok || magic != kImageMagic || k != key || nb != (uint64_t)SCILMM_NB) { fclose(fp); return nullptr; }
  Symbolic* S = new Symbolic();
  image_fields(io, *S);
  uint64_t trailer[2] = {0, 0};
  const bool trailer_ok = io.ok && fread(trailer, sizeof(uint64_t), 2, fp) == 2 && trailer[0] == io.bytes && trailer[1] == io.checksum() &&
                          fgetc(fp) == EOF;
  fclose(fp);
  // structural sanity: sizes must agree with the header fields, indices must stay inside what they index
  bool sane = trailer_ok && S->n >= 0 && S->nsuper >= 0 && (int64_t)S->perm.size() == S->n && (int64_t)S->iperm.size() == S->n &&
              (int64_t)S->sn_start.size() == (int64_t)S->nsuper + 1 && (int64_t)S->sn_rowptr.size() == (int64_t)S->nsuper + 1 &&
              (int64_t)S->sn_loff.size() >= S->nsuper && (int64_t)S->asm_dst.size() == S->nnz_pattern &&
              (int64_t)S->diag_dst.size() == S->n && (int64_t)S->pat_colptr.size() == (int64_t)S->n + 1 &&
              (int64_t)S->val_slot.size() == S->K && (int64_t)S->val_src.size() == S->K &&
              (int64_t)S->level_ptr.size() == (int64_t)S->nlevels + 1;
  if (sane) {
    const int64_t n = S->n, nst = S->nnzL_stored;
    bool ok_perm = true, ok_rows = true, ok_asm = true;
    for (int64_t i = 0; i < n; ++i) {
      const int32_t p = S->perm[(size_t)i];
      if (p < 0 || p >= n || S->iperm[(size_t)p] != (int32_t)i) { ok_perm = false; break; }
    }
    const int64_t nrows = (int64_t)S->sn_rows.size();
    if (nrows != S->sn_rowptr[(size_t)S->nsuper]) ok_rows = false;
#pragma omp parallel for reduction(&& : ok_rows) num_threads(host_threads()) if (nrows > (1 << 20))
    for (int64_t t = 0; t < nrows; ++t) ok_rows = ok_rows && S->sn_rows[(size_t)t] >= 0 && S->sn_rows[(size_t)t] < n;
    const int64_t nasm = (int64_t)S->asm_dst.size();
#pragma omp parallel for reduction(&& : ok_asm) num_threads(host_threads()) if (nasm > (1 << 20))
    for (int64_t t = 0; t < nasm; ++t) ok_asm = ok_asm && S->asm_dst[(size_t)t] >= 0 && S->asm_dst[(size_t)t] < nst;
    for (int64_t j = 0; j < n && ok_asm; ++j) ok_asm = S->diag_dst[(size_t)j] >= 0 && S->diag_dst[(size_t)j] < nst;
    sane = ok_perm && ok_rows && ok_asm;
  }
  if (!sane) { delete S; return nullptr; }
  S->combos_built = false;
  // (tile combo arrays of a fresh analysis are sized by build_tile_combos on demand)
  return S;
}

}  // namespace scilmm
