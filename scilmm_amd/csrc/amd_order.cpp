// Approximate-minimum-degree fill-reducing ordering on a quotient graph (host, C++).
//
// Role on the hot path: replaces the ordering step of cholmod_analyze that the reference triggers
// on every likelihood evaluation (reference scilmm/SparseCholesky.py:22-26 -> sksparse
// cholesky(..., ordering_method='nesdis')).  Here it runs ONCE per sparsity pattern.
//
// Written from the published algorithm (Amestoy, Davis, Duff, "An approximate minimum degree
// ordering algorithm", SIMAX 1996): quotient graph with element absorption, approximate external
// degrees, mass elimination, hash-based supervariable detection, aggressive absorption, dense-row
// deferral.  No third-party source was available or consulted in this container.
#include <cstdint>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>

namespace scilmm {

namespace {
constexpr int64_t DEAD = -1;

struct Amd {
  int32_t n;
  std::vector<int32_t> iw;       // adjacency workspace
  int64_t pfree;
  std::vector<int64_t> pe;       // start of list in iw, DEAD if node is gone
  std::vector<int32_t> len, elen, nv, degree, head, next, last, hhead, parent_of, order_pos;
  std::vector<int64_t> w;
  std::vector<uint8_t> is_elem;

  void compact() {
    // garbage-collect iw: keep only live lists (variables with nv>0 alive, live elements)
    // mark the first entry of each live list with -(node+1) after saving it in pe
    std::vector<int32_t> firsts(n);
    for (int32_t i = 0; i < n; ++i) {
      if (pe[i] >= 0 && len[i] > 0) {
        firsts[i] = iw[pe[i]];
        iw[pe[i]] = -(i + 1);
      }
    }
    int64_t dst = 0, src = 0;
    while (src < pfree) {
      int32_t v = iw[src];
      if (v < 0) {
        int32_t i = -v - 1;
        int64_t l = len[i];
        pe[i] = dst;
        iw[dst++] = firsts[i];
        for (int64_t t = 1; t < l; ++t) iw[dst++] = iw[src + t];
        src += l;
      } else {
        ++src;
      }
    }
    pfree = dst;
  }
};
}  // namespace

// g_ptr/g_idx: symmetric adjacency WITHOUT the diagonal (both directions present), 0-based.
// perm_out[k] = original index of the k-th pivot.
void amd_order(int32_t n, const int64_t* g_ptr, const int32_t* g_idx, int32_t* perm_out, double dense_factor) {
  if (n == 0) return;
  Amd a;
  a.n = n;
  int64_t nz = g_ptr[n];
  int64_t iwlen = nz + nz / 4 + 4 * (int64_t)n + 64;
  a.iw.assign(iwlen, 0);
  a.pe.assign(n, DEAD);
  a.len.assign(n, 0);
  a.elen.assign(n, 0);
  a.nv.assign(n, 1);
  a.degree.assign(n, 0);
  a.head.assign(n + 1, -1);
  a.next.assign(n, -1);
  a.last.assign(n, -1);
  a.hhead.assign(n, -1);
  a.parent_of.assign(n, -1);
  a.order_pos.assign(n, -1);
  a.w.assign(n, 1);
  a.is_elem.assign(n, 0);
  if (nz > 0) std::memcpy(a.iw.data(), g_idx, sizeof(int32_t) * nz);  // (an edgeless graph hands over a null g_idx)
  a.pfree = nz;

  int32_t dense = (dense_factor <= 0) ? n : (int32_t)std::max(16.0, dense_factor * std::sqrt((double)n));
  if (dense > n) dense = n;

  std::vector<int32_t> pivots;  // elimination sequence of principal pivots
  pivots.reserve(n);
  std::vector<int32_t> dense_nodes;
  int32_t nel = 0;
  for (int32_t i = 0; i < n; ++i) {
    a.pe[i] = g_ptr[i];
    a.len[i] = (int32_t)(g_ptr[i + 1] - g_ptr[i]);
    a.degree[i] = a.len[i];
  }
  // dense rows are deferred to the end (treated as already-eliminated for degree purposes)
  for (int32_t i = 0; i < n; ++i) {
    if (a.degree[i] > dense) {
      dense_nodes.push_back(i);
      a.nv[i] = 0;
      a.pe[i] = DEAD;
      a.elen[i] = -1;
      nel++;
    }
  }
  auto deg_insert = [&](int32_t i, int32_t d) {
    int32_t h = a.head[d];
    a.next[i] = h;
    a.last[i] = -1;
    if (h != -1) a.last[h] = i;
    a.head[d] = i;
  };
  auto deg_remove = [&](int32_t i) {
    int32_t d = a.degree[i];
    int32_t nx = a.next[i], pv = a.last[i];
    if (nx != -1) a.last[nx] = pv;
    if (pv != -1) a.next[pv] = nx; else a.head[d] = nx;
  };
  for (int32_t i = 0; i < n; ++i) {
    if (a.nv[i] == 0) continue;
    if (a.degree[i] == 0) {
      // isolated node: eliminate right away
      a.order_pos[i] = (int32_t)pivots.size();
      pivots.push_back(i);
      a.is_elem[i] = 1;
      a.pe[i] = DEAD;
      a.elen[i] = -1;
      nel++;
      a.w[i] = 0;
    } else {
      deg_insert(i, a.degree[i]);
    }
  }
  int64_t wflg = 2;
  int32_t mindeg = 0;
  std::vector<int32_t>& iw = a.iw;

  while (nel < n) {
    // ---- select pivot of minimum approximate degree
    int32_t me = -1;
    for (; mindeg < n; ++mindeg) {
      me = a.head[mindeg];
      if (me != -1) break;
    }
    if (me == -1) break;  // should not happen
    deg_remove(me);
    int32_t elenme = a.elen[me];
    int32_t nvpiv = a.nv[me];
    nel += nvpiv;
    a.nv[me] = -nvpiv;
    int32_t degme = 0;
    int64_t pme1, pme2;
    if (elenme == 0) {
      // in-place construction of the new element from me's variable list
      pme1 = a.pe[me];
      pme2 = pme1 - 1;
      for (int64_t p = pme1; p < pme1 + a.len[me]; ++p) {
        int32_t i = iw[p];
        int32_t nvi = a.nv[i];
        if (nvi > 0) {
          degme += nvi;
          a.nv[i] = -nvi;
          iw[++pme2] = i;
          deg_remove(i);
        }
      }
    } else {
      // need fresh space: worst case size is bounded by degree[me] entries ... ensure room
      int64_t need = 0;
      {
        int64_t p = a.pe[me];
        for (int32_t k = 0; k < elenme; ++k) need += a.len[iw[p + k]];
        need += a.len[me] - elenme;
      }
      if (a.pfree + need > (int64_t)iw.size()) {
        a.compact();
        if (a.pfree + need > (int64_t)iw.size()) iw.resize(a.pfree + need + (int64_t)iw.size() / 4);
      }
      int64_t p = a.pe[me];
      pme1 = a.pfree;
      int32_t slenme = a.len[me] - elenme;
      for (int32_t knt1 = 1; knt1 <= elenme + 1; ++knt1) {
        int32_t e;
        int64_t pj;
        int32_t ln;
        if (knt1 > elenme) {
          e = me;
          pj = p;
          ln = slenme;
        } else {
          e = iw[p++];
          pj = a.pe[e];
          ln = a.len[e];
        }
        for (int32_t knt2 = 0; knt2 < ln; ++knt2) {
          int32_t i = iw[pj++];
          int32_t nvi = a.nv[i];
          if (nvi > 0) {
            degme += nvi;
            a.nv[i] = -nvi;
            iw[a.pfree++] = i;
            deg_remove(i);
          }
        }
        if (e != me) {
          // absorb element e into me
          a.pe[e] = DEAD;
          a.parent_of[e] = me;
          a.w[e] = 0;
        }
      }
      pme2 = a.pfree - 1;
    }
    a.degree[me] = degme;
    a.pe[me] = pme1;
    a.len[me] = (int32_t)(pme2 - pme1 + 1);
    a.elen[me] = -2;  // flag: element
    a.is_elem[me] = 1;
    a.order_pos[me] = (int32_t)pivots.size();
    pivots.push_back(me);

    // ---- scan 1: compute |Le \ Lme| for all elements adjacent to variables of Lme
    for (int64_t pq = pme1; pq <= pme2; ++pq) {
      int32_t i = iw[pq];
      int32_t eln = a.elen[i];
      if (eln > 0) {
        int32_t nvi = -a.nv[i];
        int64_t wnvi = wflg - nvi;
        int64_t p = a.pe[i];
        for (int32_t k = 0; k < eln; ++k) {
          int32_t e = iw[p + k];
          int64_t we = a.w[e];
          if (we >= wflg) we -= nvi;
          else if (we != 0) we = a.degree[e] + wnvi;
          a.w[e] = we;
        }
      }
    }
    // ---- scan 2: degree update, pruning, hashing
    for (int64_t pq = pme1; pq <= pme2; ++pq) {
      int32_t i = iw[pq];
      int64_t p1 = a.pe[i];
      int64_t p2 = p1 + a.elen[i] - 1;
      int64_t pn = p1;
      uint64_t hash = 0;
      int64_t deg = 0;
      for (int64_t p = p1; p <= p2; ++p) {
        int32_t e = iw[p];
        int64_t we = a.w[e];
        if (we != 0) {
          int64_t dext = we - wflg;
          if (dext > 0) {
            deg += dext;
            iw[pn++] = e;
            hash += (uint64_t)e;
          } else {
            // aggressive absorption: Le is a subset of Lme
            a.pe[e] = DEAD;
            a.parent_of[e] = me;
            a.w[e] = 0;
          }
        }
      }
      a.elen[i] = (int32_t)(pn - p1 + 1);
      int64_t p3 = pn;
      int64_t p4 = p1 + a.len[i];
      for (int64_t p = p2 + 1; p < p4; ++p) {
        int32_t j = iw[p];
        int32_t nvj = a.nv[j];
        if (nvj > 0) {
          deg += nvj;
          iw[pn++] = j;
          hash += (uint64_t)j;
        }
      }
      if (a.elen[i] == 1 && p3 == pn) {
        // mass elimination: i has no neighbours left outside Lme
        a.pe[i] = DEAD;
        a.parent_of[i] = me;
        int32_t nvi = -a.nv[i];
        degme -= nvi;
        nvpiv += nvi;
        nel += nvi;
        a.nv[i] = 0;
        a.elen[i] = -1;
      } else {
        a.degree[i] = (int32_t)std::min<int64_t>(a.degree[i], deg);
        // make room for the new element at the front of i's list
        iw[pn] = iw[p3];
        iw[p3] = iw[p1];
        iw[p1] = me;
        a.len[i] = (int32_t)(pn - p1 + 1);
        int32_t h = (int32_t)(hash % (uint64_t)n);
        // insert into hash bucket
        a.next[i] = a.hhead[h];
        a.hhead[h] = i;
        a.last[i] = h;
      }
    }
    a.degree[me] = degme;
    // bump the flag past every stale value
    {
      int64_t lemax = std::max<int64_t>(degme, 1);
      wflg += lemax + n;  // int64 flag space: no overflow handling needed
    }
    // ---- supervariable detection
    for (int64_t pq = pme1; pq <= pme2; ++pq) {
      int32_t i = iw[pq];
      if (a.nv[i] >= 0) continue;
      int32_t h = a.last[i];
      int32_t b = a.hhead[h];
      if (b == -1) continue;
      a.hhead[h] = -1;  // take the whole bucket
      // process chain b -> next ...
      int32_t ii = b;
      while (ii != -1 && a.next[ii] != -1) {
        int32_t ln = a.len[ii];
        int32_t eln = a.elen[ii];
        for (int64_t p = a.pe[ii] + 1; p < a.pe[ii] + ln; ++p) a.w[iw[p]] = wflg;
        int32_t jlast = ii;
        int32_t j = a.next[ii];
        while (j != -1) {
          bool ok = (a.len[j] == ln) && (a.elen[j] == eln);
          for (int64_t p = a.pe[j] + 1; ok && p < a.pe[j] + ln; ++p)
            if (a.w[iw[p]] != wflg) ok = false;
          if (ok) {
            // j is indistinguishable from ii: merge
            a.pe[j] = DEAD;
            a.parent_of[j] = ii;
            a.nv[ii] += a.nv[j];  // both negative here
            a.nv[j] = 0;
            a.elen[j] = -1;
            j = a.next[j];
            a.next[jlast] = j;
          } else {
            jlast = j;
            j = a.next[j];
          }
        }
        wflg++;
        ii = a.next[ii];
      }
    }
    // ---- finalize: restore nv, final approximate degrees, re-insert in degree lists
    int64_t p = pme1;
    int32_t nleft = n - nel;
    for (int64_t pq = pme1; pq <= pme2; ++pq) {
      int32_t i = iw[pq];
      int32_t nvi = -a.nv[i];
      if (nvi > 0) {
        a.nv[i] = nvi;
        int64_t deg = (int64_t)a.degree[i] + degme - nvi;
        deg = std::min<int64_t>(deg, nleft - nvi);
        if (deg < 0) deg = 0;
        a.degree[i] = (int32_t)deg;
        deg_insert(i, (int32_t)deg);
        if ((int32_t)deg < mindeg) mindeg = (int32_t)deg;
        iw[p++] = i;
      }
    }
    a.nv[me] = nvpiv;
    a.len[me] = (int32_t)(p - pme1);
    if (a.len[me] == 0) {
      a.pe[me] = DEAD;
      a.w[me] = 0;
    }
    if (elenme != 0) a.pfree = p;  // reclaim tail of the freshly built element
  }

  // ---- build the permutation: pivots in order, each followed by the variables absorbed into it
  // resolve every non-pivot variable to the pivot it was eliminated with
  std::vector<int32_t> root(n, -1);
  for (int32_t i = 0; i < n; ++i) {
    if (a.order_pos[i] >= 0) { root[i] = i; continue; }
  }
  std::vector<uint8_t> isdense(n, 0);
  for (int32_t d : dense_nodes) isdense[d] = 1;
  std::vector<int32_t> stack;
  for (int32_t i = 0; i < n; ++i) {
    if (root[i] != -1 || isdense[i]) continue;
    // follow parent_of chain through merged variables until a pivot is met
    stack.clear();
    int32_t j = i;
    while (j != -1 && root[j] == -1) {
      stack.push_back(j);
      j = a.parent_of[j];
    }
    int32_t r = (j == -1) ? -1 : root[j];
    for (int32_t s : stack) root[s] = r;
  }
  // count members per pivot
  std::vector<int64_t> cnt(pivots.size() + 1, 0);
  for (int32_t i = 0; i < n; ++i) {
    if (isdense[i]) continue;
    int32_t r = root[i];
    if (r < 0) continue;
    cnt[a.order_pos[r] + 1]++;
  }
  for (size_t k = 0; k < pivots.size(); ++k) cnt[k + 1] += cnt[k];
  std::vector<int64_t> fill(cnt.begin(), cnt.end() - 1);
  int64_t placed = 0;
  // pivot itself first
  for (size_t k = 0; k < pivots.size(); ++k) {
    perm_out[fill[k]++] = pivots[k];
    placed++;
  }
  for (int32_t i = 0; i < n; ++i) {
    if (isdense[i]) continue;
    int32_t r = root[i];
    if (r < 0 || r == i) continue;
    perm_out[fill[a.order_pos[r]]++] = i;
    placed++;
  }
  // any unresolved node (defensive) and dense nodes go last
  std::vector<uint8_t> seen(n, 0);
  for (int64_t k = 0; k < placed; ++k) seen[perm_out[k]] = 1;
  for (int32_t i = 0; i < n; ++i)
    if (!seen[i] && !isdense[i]) perm_out[placed++] = i;
  // dense nodes ordered by increasing original degree
  std::sort(dense_nodes.begin(), dense_nodes.end(), [&](int32_t x, int32_t y) {
    int64_t dx = g_ptr[x + 1] - g_ptr[x], dy = g_ptr[y + 1] - g_ptr[y];
    return dx != dy ? dx < dy : x < y;
  });
  for (int32_t d : dense_nodes) perm_out[placed++] = d;
}

}  // namespace scilmm
