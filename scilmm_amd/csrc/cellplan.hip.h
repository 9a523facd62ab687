// Device-side construction of the cell-wise update plan (the lists k_sparse_cells consumes).
//
// The host used to enumerate every target cell of every small update pair (39 M cells at the 100k pedigree,
// 382 M at 300k: seconds to minutes of sorting and gigabytes of staging per pattern).  Here the host only lists
// the small combos; the device expands them to cells, orders them by (stream class, level, target address) with
// a stable radix sort, cuts the groups of equal target address and moves the long groups behind the short ones
// of their level -- the same layout as before, built in tens of milliseconds.  Emission order and the sorts
// are deterministic, so the order of the contributions to a cell (and with it every bit of L) is reproducible.
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

namespace scilmm {

struct CellCombo {
  int64_t loff_d;    // L offset of the descendant panel
  int64_t rowoff_d;  // its row list in sn_rows
  int64_t tgt_loff;  // L offset of the target panel
  int64_t tgt_rows;  // sn_rows index of the first row of the target TILE
  int32_t md, wd;    // descendant panel rows, width
  int32_t ta, nt;    // descendant rows [ta, ta+nt) land in the tile
  int32_t p0, nq;    // descendant rows [p0, p0+nq) are target columns
  int32_t ip0;       // >= 0: tile positions ip0.. (consecutive); -1: search
  int32_t ms;        // target panel rows (leading dimension)
  int32_t R0, nrow;  // first panel row of the tile, rows in the tile
  int32_t c0s;       // first column label of the target front
  int32_t level, cls;  // level of the target; 0 early, 1 late (main stream), 2 late (rest stream)
};

constexpr unsigned long long CELL_INVALID = ~0ull;
constexpr int CELL_LEVEL_SHIFT = 40, CELL_CLASS_SHIFT = 60;
constexpr unsigned long long CELL_DST_MASK = (1ull << CELL_LEVEL_SHIFT) - 1;

// one thread per POTENTIAL cell (combo, t, q); cells above the diagonal of a diagonal tile get the invalid key
__global__ __launch_bounds__(256) void k_emit_cells(int64_t total, int64_t ncombo, const CellCombo* __restrict__ cc,
                                                    const int64_t* __restrict__ off, const int32_t* __restrict__ sn_rows,
                                                    unsigned long long* __restrict__ key, uint32_t* __restrict__ idx,
                                                    int64_t* __restrict__ st, int64_t* __restrict__ sq, int32_t* __restrict__ md,
                                                    int32_t* __restrict__ wd, unsigned long long* __restrict__ n_invalid) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    int64_t lo = 0, hi = ncombo;  // last combo with off[c] <= i
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (off[mid] <= i) lo = mid; else hi = mid;
    }
    const CellCombo c = cc[lo];
    const int64_t local = i - off[lo];
    const int32_t tt = (int32_t)(local / c.nq), qq = (int32_t)(local - (int64_t)tt * c.nq);
    const int32_t t = c.ta + tt, q = c.p0 + qq;
    int64_t R;
    if (c.ip0 >= 0) {
      R = (int64_t)c.R0 + c.ip0 + tt;
    } else {
      const int32_t lab = sn_rows[c.rowoff_d + t];
      const int32_t* rs = sn_rows + c.tgt_rows;
      int a = 0, b = c.nrow;
      while (a < b) {
        const int mid = (a + b) >> 1;
        if (rs[mid] < lab) a = mid + 1; else b = mid;
      }
      R = (int64_t)c.R0 + a;
    }
    const int64_t j = (int64_t)sn_rows[c.rowoff_d + q] - c.c0s;
    const bool valid = R >= j;  // the strict upper part of a diagonal block is never referenced
    const unsigned long long dst = (unsigned long long)(c.tgt_loff + j * (int64_t)c.ms + R);
    key[i] = valid ? (((unsigned long long)c.cls << CELL_CLASS_SHIFT) | ((unsigned long long)c.level << CELL_LEVEL_SHIFT) | dst)
                   : CELL_INVALID;
    idx[i] = (uint32_t)i;
    st[i] = c.loff_d + t;
    sq[i] = c.loff_d + q;
    md[i] = c.md;
    wd[i] = c.wd;
    if (!valid) atomicAdd(n_invalid, 1ull);
  }
}

// group key with the "long group" bit between level and address: long groups sort behind the short ones of a level
__global__ void k_group_keys(int64_t ng, const unsigned long long* __restrict__ ukey, const int64_t* __restrict__ ucnt,
                             int64_t long_limit, unsigned long long* __restrict__ gkey, uint32_t* __restrict__ gidx) {
  const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  const unsigned long long k = ukey[g];
  const unsigned long long hi = k >> CELL_LEVEL_SHIFT;  // (class, level)
  const unsigned long long lng = ucnt[g] > long_limit ? 1ull : 0ull;
  // dst needs < 39 bits here (L offsets below 5.5e11 doubles = 4.4 TB): one bit of the address field is the flag
  gkey[g] = (hi << CELL_LEVEL_SHIFT) | (lng << (CELL_LEVEL_SHIFT - 1)) | (k & (CELL_DST_MASK >> 1));
  gidx[g] = (uint32_t)g;
}

__global__ void k_gather_counts(int64_t ng, const uint32_t* __restrict__ order, const int64_t* __restrict__ ucnt,
                                int64_t* __restrict__ cnt2) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < ng) cnt2[k] = ucnt[order[k]];
}

// final arrays: group k (new order) takes the entries of old group order[k]
__global__ void k_finish_groups(int64_t ng, const unsigned long long* __restrict__ gkey_sorted, int64_t* __restrict__ udst) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < ng) udst[k] = (int64_t)(gkey_sorted[k] & (CELL_DST_MASK >> 1));
}

// groups per (class, level, long) bucket = distance between the bucket boundaries in the sorted group keys
// (one binary search per bucket; counting with atomics took 0.7 s at 4e8 groups)
__global__ void k_bucket_counts(int32_t nbuckets, int32_t NL, int64_t ng, const unsigned long long* __restrict__ gkey_sorted,
                                unsigned int* __restrict__ counters) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nbuckets) return;
  auto bound = [&](int bb) -> int64_t {  // first group whose key is >= the smallest key of bucket bb
    if (bb >= nbuckets) return ng;
    const unsigned long long cls = (unsigned long long)(bb / (2 * NL)), rem = (unsigned long long)(bb % (2 * NL));
    const unsigned long long key = (cls << CELL_CLASS_SHIFT) | ((rem >> 1) << CELL_LEVEL_SHIFT) | ((rem & 1ull) << (CELL_LEVEL_SHIFT - 1));
    int64_t lo = 0, hi = ng;
    while (lo < hi) {
      const int64_t mid = (lo + hi) >> 1;
      if (gkey_sorted[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
  };
  counters[b] = (unsigned int)(bound(b + 1) - bound(b));
}

__global__ __launch_bounds__(256) void k_gather_entries(int64_t ne, int64_t ng, const int64_t* __restrict__ grp2,
                                                        const uint32_t* __restrict__ order, const int64_t* __restrict__ ustart,
                                                        const uint32_t* __restrict__ sidx, const int64_t* __restrict__ st,
                                                        const int64_t* __restrict__ sq, const int32_t* __restrict__ md,
                                                        const int32_t* __restrict__ wd, int64_t* __restrict__ ost,
                                                        int64_t* __restrict__ osq, int32_t* __restrict__ omd, int32_t* __restrict__ owd) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < ne; e += stride) {
    int64_t lo = 0, hi = ng;  // last group with grp2[k] <= e
    while (hi - lo > 1) {
      const int64_t mid = (lo + hi) >> 1;
      if (grp2[mid] <= e) lo = mid; else hi = mid;
    }
    const int64_t src = ustart[order[lo]] + (e - grp2[lo]);
    const uint32_t i = sidx[src];
    ost[e] = st[i];
    osq[e] = sq[i];
    omd[e] = md[i];
    owd[e] = wd[i];
  }
}

}  // namespace scilmm
