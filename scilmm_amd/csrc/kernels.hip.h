// Hand-written CDNA4 (gfx950) kernels of the numeric phase.  Wave = 64 lanes; 256-thread workgroups (one
// wave per SIMD) except the update and chain-sweep kernels (512 threads, two waves per SIMD: a single wave
// cannot keep the fp64 matrix pipe busy); fp64 MFMA v_mfma_f64_16x16x4_f64 for every GEMM-shaped step.
//
// Data layout in HBM:
//   L        : supernodal panels, column-major m_s x w_s, leading dimension m_s, at sn_loff[s]
//   invD     : w_s x w_s inverse of every diagonal block (column-major) at inv_off[s]
//   vals_k   : A_k values permuted into "pattern slot" order (permuted CSC of tril(union pattern))
//   W / X    : right-hand sides, PERMUTED row order, row-major n x rp (rp = r rounded up to 16)
//
// MFMA operand convention used everywhere (cdna_hip_programming.md §3, f64 note):
//   D[M][N] += sum_k Aop[M][k] * Bop[k][N];  lane l supplies Aop[l&15][l>>4] and Bop[l>>4][l&15];
//   lane l receives D[(l>>4) + 4*reg][l&15], reg = 0..3.
#pragma once
#include <hip/hip_runtime.h>
#ifndef SCILMM_KC
#define SCILMM_KC 16
#endif
#ifndef SCILMM_NB
#define SCILMM_NB 128
#endif
#ifndef SCILMM_UPD_WAVES
#define SCILMM_UPD_WAVES 2
#endif
#include <stdint.h>

namespace scilmm {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int NB = SCILMM_NB; // max supernode block width (symbolic max_width must be <= NB)
constexpr int NJB = NB / 16;  // 16-column MFMA tiles across a block
constexpr int TM = 128;       // target rows per tile
constexpr int KCS = 16;       // k-chunk of the trsm / solve kernels
constexpr int KC = SCILMM_KC;  // k-chunk of the update kernel: 2 buffers x 16 x (144 + 144) doubles = 74 KB -> two workgroups per CU
constexpr int LDA = TM + 16;  // k-major LDS leading dims: (ld*8 B) == 128 mod 256 -> conflict-free b64 reads
constexpr int LDB = NB + 16;
constexpr int RPMAX = 128;    // max padded RHS columns per pass
constexpr int LDP = 34;       // [k][q] LDS image of a panel slice (q-chunk of 32)

struct DevSym {
  int32_t n, nsuper;
  const int32_t* sn_start;
  const int64_t* sn_rowptr;
  const int32_t* sn_rows;
  const int64_t* sn_loff;
  const int64_t* inv_off;
  const int32_t* upd_src;
  const int32_t* upd_p0;
  const int32_t* upd_p1;
  const int32_t* tile_front;
  const int64_t* tile_base;
  const int64_t* combo_ptr;
  const int32_t* combo_pair;
  const int32_t* combo_ta;
  const int32_t* combo_tb;
  const int64_t* asm_dst;
  const int64_t* diag_dst;
  const int64_t* pat_colptr;
  const int32_t* pat_row;
  const int32_t* perm;
};

struct ValPtrs {
  const double* v[8];
  double s2[8];
  int32_t count;
};

__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// V = sum_k s2_k A_k scattered into the (zeroed) panels.  HBM-bound: per pattern entry reads 8 B per
// general matrix + 8 B map, writes 8 B.
__global__ void k_assemble(int64_t nnz, const int64_t* __restrict__ dst, ValPtrs vp, double* __restrict__ L) {
  int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < nnz; e += stride) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 8; ++k)
      if (k < vp.count) v += vp.s2[k] * vp.v[k][e];
    const int64_t d = dst[e];
    if (d >= 0) L[d] = v;  // (-1: a panel this rank does not hold -- distributed tail)
  }
}

// dst[i] += src[i]  (multi-GPU forward sweep: the all-reduced tail contributions join the replicated right-hand side)
__global__ void k_add_rows(int64_t count, const double* __restrict__ src, double* __restrict__ dst) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) dst[i] += src[i];
}

// multi-GPU: the status word travels through the communication layer as a double (its buffers are all fp64)
__global__ void k_status_pack(const int32_t* __restrict__ status, double* __restrict__ out) { *out = (double)*status; }
__global__ void k_status_unpack(const double* __restrict__ in, int32_t* __restrict__ status) { *status = (int32_t)*in; }

__global__ void k_add_diag(int32_t n, const int64_t* __restrict__ diag_dst, ValPtrs vp, double* __restrict__ L) {
  int32_t j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= n) return;
  double v = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k)
    if (k < vp.count) v += vp.s2[k] * vp.v[k][j];
  if (diag_dst[j] >= 0) L[diag_dst[j]] += v;
}

// ------------------------------------------------------------------------------------------------
// Shared MFMA stage: acc[jb][ib] (16 x 16 each) += Bs^T[j][k] * As[k][i] over kc4 staged k values.
// D[M=j][N=i]; wave wv owns target rows i in [32 wv, 32 wv + 32), all NB columns j.
template <bool MFMA>
__device__ __forceinline__ void tile_mma(const double* __restrict__ As, const double* __restrict__ Bs, int kc4, int ncb,
                                         int lane, int wv, d4 (&acc)[NJB][2]) {
  const int li = lane & 15, lk = lane >> 4;
  if (MFMA) {
    const double* ap = As + lk * LDA + 32 * wv + li;
    const double* bp = Bs + lk * LDB + li;
    if (ncb == NJB) {
      // full-width target block: straight-line body, all LDS reads issued ahead of the MFMAs
#pragma unroll 2
      for (int k4 = 0; k4 < kc4; k4 += 4) {
        const double b0 = ap[k4 * LDA], b1 = ap[k4 * LDA + 16];
        double a[NJB];
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) a[jb] = bp[k4 * LDB + 16 * jb];
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) {
          acc[jb][0] = mfma_f64(a[jb], b0, acc[jb][0]);
          acc[jb][1] = mfma_f64(a[jb], b1, acc[jb][1]);
        }
      }
    } else {
      for (int k4 = 0; k4 < kc4; k4 += 4) {
        const double b0 = ap[k4 * LDA], b1 = ap[k4 * LDA + 16];
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) {
          if (jb < ncb) {
            const double a = bp[k4 * LDB + 16 * jb];
            acc[jb][0] = mfma_f64(a, b0, acc[jb][0]);
            acc[jb][1] = mfma_f64(a, b1, acc[jb][1]);
          }
        }
      }
    }
  } else {
    // scalar restatement of the same tile product in the same accumulator layout (debug path)
    for (int k = 0; k < kc4; ++k) {
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            acc[jb][ib][r] += Bs[k * LDB + 16 * jb + lk + 4 * r] * As[k * LDA + 32 * wv + 16 * ib + li];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Left-looking supernodal update of one 128-row tile of a target panel:
//   P_s[tile, :] -= sum over combos (d, rows ta..ta+nt of d)  L_d[ta:ta+nt, :] * L_d[p0:p0+nq, :]^T
// Rows/columns of each descendant are gathered into TARGET coordinates inside LDS, so the
// accumulators stay in registers across all descendants.  The K dimension (columns of all descendants,
// concatenated) is streamed in chunks of KC through a double-buffered LDS image: the global loads of
// chunk i+1 are in flight in registers while chunk i feeds the MFMAs; one barrier per chunk.
// A work item is (tile, combo range [cb,ce), slot): slot < 0 subtracts straight into the panel,
// slot >= 0 writes the partial product to scratch (split-K; folded in by k_reduce).
struct ComboDesc {
  int64_t loff;     // L offset of the descendant panel
  int64_t rowoff;   // offset of its row list in sn_rows
  int32_t md, wd;   // panel rows (leading dimension) and width (K extent)
  int32_t ta, nt;   // descendant rows [ta, ta+nt) land in this tile
  int32_t p0, nq;   // descendant rows [p0, p0+nq) are the target's columns
  int32_t ip0;      // >= 0: rows land at consecutive tile positions ip0..; -1: look each one up
  int32_t jp0;      // >= 0: columns land at consecutive target columns jp0..; -1: look each one up
  int32_t ilo, ihi; // first / last tile position touched (rows are sorted, so everything lies in between)
  int32_t jlo, jhi; // first / last target column touched
};

struct UpdWork {
  int32_t tile;
  int32_t slot;     // partial-slot index or -1
  int64_t cb, ce;   // combo range
};

constexpr int UPD_THREADS = 512;  // update kernel: eight waves per workgroup

// Eight-wave variant of the tile product: wave wv owns 16 target rows, acc[jb] is the 16 x 16 tile of columns 16 jb..
// jb0..jb1: 16-column tiles this chunk touches (the rest of the accumulators is skipped)
template <bool MFMA>
__device__ __forceinline__ void tile_mma8(const double* __restrict__ As, const double* __restrict__ Bs, int kc4, int jb0,
                                          int jb1, int lane, int wv, d4 (&acc)[NJB]) {
  const int li = lane & 15, lk = lane >> 4;
  if (MFMA) {
    const double* ap = As + lk * LDA + 16 * wv + li;
    const double* bp = Bs + lk * LDB + li;
    if (jb0 == 0 && jb1 == NJB - 1) {
#pragma unroll 2
      for (int k4 = 0; k4 < kc4; k4 += 4) {
        const double b0 = ap[k4 * LDA];
        double a[NJB];
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) a[jb] = bp[k4 * LDB + 16 * jb];
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) acc[jb] = mfma_f64(a[jb], b0, acc[jb]);
      }
    } else {
      for (int k4 = 0; k4 < kc4; k4 += 4) {
        const double b0 = ap[k4 * LDA];
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb)
          if (jb >= jb0 && jb <= jb1) acc[jb] = mfma_f64(bp[k4 * LDB + 16 * jb], b0, acc[jb]);
      }
    }
  } else {
    for (int k = 0; k < kc4; ++k)
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb)
        if (jb >= jb0 && jb <= jb1)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[jb][r] += Bs[k * LDB + 16 * jb + lk + 4 * r] * As[k * LDA + 16 * wv + li];
  }
}


// ------------------------------------------------------------------------------------------------
// k_update2: the explicit-combo update kernel (contract: the comment above ComboDesc).  fp64 MFMA issue blocks on the matrix pipe
// (~100 cycles per v_mfma_f64_16x16x4_f64 per SIMD), so instructions placed BETWEEN the MFMAs of a k-step are
// nearly free.  The staging of the next chunk is therefore cut into KC/4 pieces and spread over the k-steps of
// the current chunk: piece i of chunk c+1 goes registers -> LDS (other buffer) and the same registers are at
// once re-loaded with piece i of chunk c+2 from global memory, one chunk of MFMA time ahead of its use.
// One register set, one barrier per chunk (two when the target mapping of the incoming chunk differs from what
// the LDS buffer holds: every thread first clears its own old cells).
template <bool MFMA>
__global__ __launch_bounds__(UPD_THREADS, SCILMM_UPD_WAVES) void k_update2(DevSym S, const UpdWork* __restrict__ work,
                                                 const ComboDesc* __restrict__ combos, double* __restrict__ L,
                                                 double* __restrict__ scratch) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Abuf = smem;                         // [2][KC*LDA]
  double* Bbuf = smem + 2 * KC * LDA;          // [2][KC*LDB]
  int32_t* rowlab = (int32_t*)(smem + 2 * KC * LDA + 2 * KC * LDB);  // [TM]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const UpdWork wk = work[blockIdx.x];
  const int32_t g = wk.tile;
  const int64_t cb = wk.cb, ce = wk.ce;
  if (cb >= ce) return;
  const int32_t s = S.tile_front[g];
  const int32_t ti = (int32_t)(g - S.tile_base[s]);
  const int32_t c0 = S.sn_start[s], w = S.sn_start[s + 1] - c0;
  const int32_t* rs = S.sn_rows + S.sn_rowptr[s];
  const int32_t m = (int32_t)(S.sn_rowptr[s + 1] - S.sn_rowptr[s]);
  const int32_t R0 = ti * TM;
  const int32_t nrow = min(TM, m - R0);
  const int ncb = (w + 15) >> 4;
  if (tid < TM) rowlab[tid] = tid < nrow ? rs[R0 + tid] : 0x7fffffff;
  for (int idx = tid; idx < 2 * KC * LDA + 2 * KC * LDB; idx += UPD_THREADS) smem[idx] = 0.0;
  d4 acc[NJB];
#pragma unroll
  for (int a = 0; a < NJB; ++a) acc[a] = (d4){0.0, 0.0, 0.0, 0.0};
  constexpr int KA = UPD_THREADS / TM, KB = UPD_THREADS / NB;
  constexpr int NPA = KC / KA, NPB = KC / KB, NS = KC / 4;  // pieces of A / B per thread, k-steps per chunk
  static_assert(NPA <= NS && NPB <= NS, "staging pieces must fit the k-steps of a chunk");
  const int t = tid % TM, kpa = tid / TM;
  const int q = tid % NB, kpb = tid / NB;
  // ---- three pipeline stages: L = being loaded into registers, W = being written to LDS, C = being multiplied
  struct Stage {
    int ip, jp, kc;              // this thread's tile row / target column (-1: none) and the chunk depth
    int la, lb;                  // LDS cell offsets: the row / column, or this thread's private pad cell
    int ip0, nt, jp0, nq;        // uniform mapping (for the same-cells test)
    int ilo, ihi, jb0, jb1;      // spans (for MFMA skipping)
    bool valid;
  };
  Stage SL{}, SW{}, SC{};
  int64_t cn = cb;  // cursor of the load stage
  int k0n = 0;
  ComboDesc dn = combos[cn];
  ComboDesc dnext = combos[min(cn + 1, ce - 1)];
  const double* gpa = nullptr;  // this thread's global read pointers for the load-stage chunk
  const double* gpb = nullptr;
  int64_t gmd = 0;
  auto locate_L = [&]() {
    SL.valid = true;
    SL.ip = -1;
    SL.jp = -1;
    if (t < dn.nt) {
      if (dn.ip0 >= 0) {
        SL.ip = dn.ip0 + t;
      } else {
        const int32_t lab = S.sn_rows[dn.rowoff + dn.ta + t];
        int lo = 0, hi = nrow;
        while (lo < hi) {
          int mid = (lo + hi) >> 1;
          if (rowlab[mid] < lab) lo = mid + 1; else hi = mid;
        }
        SL.ip = lo;
      }
    }
    if (q < dn.nq) SL.jp = (dn.jp0 >= 0) ? dn.jp0 + q : S.sn_rows[dn.rowoff + dn.p0 + q] - c0;
    SL.la = SL.ip >= 0 ? SL.ip : TM + (t & 15);
    SL.lb = SL.jp >= 0 ? SL.jp : NB + (q & 15);
    SL.ip0 = dn.ip0; SL.nt = dn.nt; SL.jp0 = dn.jp0; SL.nq = dn.nq;
    SL.ilo = dn.ilo; SL.ihi = dn.ihi; SL.jb0 = dn.jlo >> 4; SL.jb1 = min(dn.jhi >> 4, ncb - 1);
  };
  auto point_L = [&]() {
    SL.kc = min(KC, dn.wd - k0n);
    gmd = dn.md;
    const double* Pd = L + dn.loff + (int64_t)k0n * gmd;
    // threads without a row / column of this combo read a valid neighbour's element (the value is never used:
    // it lands in that thread's private LDS pad cell), so the hot loop needs no per-thread predicate
    gpa = Pd + dn.ta + (SL.ip >= 0 ? t : 0);
    gpb = Pd + dn.p0 + (SL.jp >= 0 ? q : 0);
  };
  auto advance_L = [&]() {
    // move the load cursor to the next chunk; invalidates SL at the end of the work item
    k0n += KC;
    if (k0n >= dn.wd) {
      ++cn;
      k0n = 0;
      if (cn >= ce) { SL.valid = false; return; }
      dn = dnext;
      dnext = combos[min(cn + 1, ce - 1)];
      locate_L();
    }
    point_L();
  };
  double ra[NPA], rb[NPB];
  // branch-free pieces: k is clamped into the chunk for the load, and rows beyond the chunk depth are written as
  // zeros (so every cell a thread owns in a buffer is rewritten by every chunk: no stale k rows)
  auto load_piece = [&](int i) {
    if (i < NPA) ra[i] = gpa[(int64_t)min(kpa + KA * i, SL.kc - 1) * gmd];
    if (i < NPB) rb[i] = gpb[(int64_t)min(kpb + KB * i, SL.kc - 1) * gmd];
  };
  auto write_piece = [&](int i, double* As, double* Bs) {
    if (i < NPA) As[(kpa + KA * i) * LDA + SW.la] = (kpa + KA * i < SW.kc) ? ra[i] : 0.0;
    if (i < NPB) Bs[(kpb + KB * i) * LDB + SW.lb] = (kpb + KB * i < SW.kc) ? rb[i] : 0.0;
  };
  // what each LDS buffer currently holds (per-thread cells + uniform mapping)
  int h_ip[2] = {-1, -1}, h_jp[2] = {-1, -1}, h_kc[2] = {0, 0};
  int u_ip0[2] = {-2, -2}, u_nt[2] = {0, 0}, u_jp0[2] = {-2, -2}, u_nq[2] = {0, 0};
  auto same_cells = [&](int b) -> bool {
    return SW.ip0 >= 0 && SW.ip0 == u_ip0[b] && SW.nt == u_nt[b] && SW.jp0 >= 0 && SW.jp0 == u_jp0[b] && SW.nq == u_nq[b];
  };
  auto clear_own = [&](int b) {
    double* As = Abuf + b * KC * LDA;
    double* Bs = Bbuf + b * KC * LDB;
    if (h_ip[b] >= 0) {
#pragma unroll
      for (int i = 0; i < NPA; ++i) As[(kpa + KA * i) * LDA + h_ip[b]] = 0.0;
    }
    if (h_jp[b] >= 0) {
#pragma unroll
      for (int i = 0; i < NPB; ++i) Bs[(kpb + KB * i) * LDB + h_jp[b]] = 0.0;
    }
  };
  auto record = [&](int b) {
    h_ip[b] = SW.ip; h_jp[b] = SW.jp; h_kc[b] = SW.kc;
    u_ip0[b] = SW.ip0; u_nt[b] = SW.nt; u_jp0[b] = SW.jp0; u_nq[b] = SW.nq;
  };
  __syncthreads();  // rowlab + zeroed buffers visible
  // ---- prologue: chunk 0 -> registers -> buffer 0; chunk 1 -> registers
  locate_L();
  point_L();
#pragma unroll
  for (int i = 0; i < NS; ++i) load_piece(i);
  SW = SL;
  advance_L();
#pragma unroll
  for (int i = 0; i < NS; ++i) write_piece(i, Abuf, Bbuf);
  record(0);
  if (SL.valid) {
#pragma unroll
    for (int i = 0; i < NS; ++i) load_piece(i);
  }
  SC = SW;
  SW = SL;
  if (SL.valid) advance_L();
  __syncthreads();
  int buf = 0;
  while (true) {
    double* Aw = Abuf + (buf ^ 1) * KC * LDA;
    double* Bw = Bbuf + (buf ^ 1) * KC * LDB;
    if (SW.valid && !same_cells(buf ^ 1)) {
      clear_own(buf ^ 1);
      __syncthreads();
    }
    const double* Ac = Abuf + buf * KC * LDA;
    const double* Bc = Bbuf + buf * KC * LDB;
    const int kc4 = (SC.kc + 3) & ~3;
    const bool mine = (16 * wv <= SC.ihi) && (16 * wv + 15 >= SC.ilo);
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      if (SW.valid) write_piece(i, Aw, Bw);
      if (SL.valid) load_piece(i);
      if (mine && 4 * i < kc4) {
        if (MFMA) {
          const double b0 = Ac[(4 * i + lk) * LDA + 16 * wv + li];
#pragma unroll
          for (int jb = 0; jb < NJB; ++jb)
            if (jb >= SC.jb0 && jb <= SC.jb1) acc[jb] = mfma_f64(Bc[(4 * i + lk) * LDB + 16 * jb + li], b0, acc[jb]);
        } else {
          for (int k = 4 * i; k < 4 * i + 4; ++k)
#pragma unroll
            for (int jb = 0; jb < NJB; ++jb)
              if (jb >= SC.jb0 && jb <= SC.jb1)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[jb][r] += Bc[k * LDB + 16 * jb + lk + 4 * r] * Ac[k * LDA + 16 * wv + li];
        }
      }
    }
    if (!SW.valid) break;
    record(buf ^ 1);
    SC = SW;
    SW = SL;
    if (SL.valid) advance_L();
    __syncthreads();
    buf ^= 1;
  }
  // epilogue: identical to k_update
  const int li = lane & 15, lr = lane >> 4;
  if (wk.slot < 0) {
    double* P = L + S.sn_loff[s];
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * jb + lr + 4 * r;
        const int i = 16 * wv + li;
        if (i < nrow && j < w) P[(int64_t)j * m + R0 + i] -= acc[jb][r];
      }
  } else {
    double* Q = scratch + (int64_t)wk.slot * (TM * NB);
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * jb + lr + 4 * r;
        const int i = 16 * wv + li;
        Q[j * TM + i] = acc[jb][r];
      }
  }
}


// ------------------------------------------------------------------------------------------------
// Dense-tail update (Symbolic::dense_first): both the target panel j and the descendant panels k0..k1-1 have every
// later column as a row, so the operands are plain column-major blocks found by arithmetic -- row r of the target is
// row (c0_j - c0_d + r) of descendant d -- and the kernels need no row lists, no combo descriptors, no binary searches,
// no clearing of LDS cells, no spans.  This is > 99.9 % of the factor flops at the 1M-individual config (170k-wide
// tail) and 82 % at 100k.
//   acc[256 x 128] = sum_d  L_d[rows of the two 128-row target tiles, :] * L_d[rows = target columns, :]^T
// One workgroup = TWO vertically adjacent 128-row target tiles that share the B operand (the descendant rows at the
// target's columns), one workgroup per CU so that the accumulators (64 doubles per lane) fit the register file.
// k_dense_b is the fp64 form, k_dense32 the fp32-product form; the register-staged and the both-operands-by-DMA
// predecessors live in csrc/tools/retired_kernels.hip.h (tuning harness only).
struct DenseWork {
  int32_t front;       // target front j (>= dense_first)
  int32_t ti0;         // first of the (one or two) target tiles
  int32_t ntiles;      // 1 or 2
  int32_t k0, k1;      // descendants dense_first + k0 .. dense_first + k1 - 1
  int32_t slot0, slot1;  // partial slab of each tile, or -1: subtract straight from the panel
  int32_t pad;
};
#ifndef SCILMM_DENSE_ABL
#define SCILMM_DENSE_ABL 0  // tuning-harness ablations (csrc/tools/dense_bench.hip only): 1 no epilogue, 2 no global loads, 3 no LDS stores
#endif
#ifndef SCILMM_DENSE_VEC
#define SCILMM_DENSE_VEC 1  // 16-byte staging accesses (0: the 8-byte form)
#endif
constexpr int DTR = 2 * TM;        // rows per dense work item
constexpr int KBA = 64;            // depth of a B buffer of the dense-tail kernel
constexpr int LDA2 = DTR + 16;     // == 16 mod 32 doubles: conflict-free b64 fragment reads

#ifdef SCILMM_DENSE_CLK
__device__ unsigned long long g_dense_clk[2];  // tuning harness only: summed wall-clock (100 MHz) / shader-clock ticks of wave 0
#endif

// a wave-uniform pointer / integer moved to scalar registers (lets loads use the SGPR-base + 32-bit lane offset form)
__device__ __forceinline__ int64_t uniform_i64(int64_t x) {
  const uint64_t v = (uint64_t)x;
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
// element at byte offset voff8 (32-bit, per lane) from a wave-uniform base: SGPR base + VGPR offset addressing
__device__ __forceinline__ double ld_off(const double* base, uint32_t voff8) {
  return *(const double*)((const char*)base + voff8);
}
__device__ __forceinline__ int uniform_int(int v) { return __builtin_amdgcn_readfirstlane(v); }
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* gl_vptr;

// ------------------------------------------------------------------------------------------------
// k_dense_b: the dense-tail update with ONLY the B operand in LDS.  A wave multiplies its own 32 rows and nobody else's,
// so its A fragments need no sharing: they are loaded straight from the panel into registers (16 consecutive rows = 128
// contiguous bytes per quarter wave), one 16-deep sub-chunk ahead of their use.  LDS holds B alone, 64 k-rows deep per
// buffer (2 x 64 x 144 doubles = 144 KB), filled by LDS-DMA (global_load_lds_dwordx4, gfx950: a wave-instruction moves
// 64 x 16 bytes from per-lane global addresses straight into 1 KiB of contiguous LDS = one k-row of the B image; k-rows
// past the end of a descendant come from a zero page): one barrier per 64 k, no A image, no staging registers.
// Both operand streams are software-pipelined inside the wave (round 3; k_dense_a in csrc/tools/retired_kernels.hip.h is
// the same kernel without that and gives the same bits: 56.9 -> 62.2 TFLOP/s alone on random operands, 62.5 -> 65.8 on
// 48-descendant items, profiles/r3_k_dense_b_isolated.txt):
//  * B fragments: the eight LDS reads of k-step t+1 are issued BETWEEN the MFMAs of k-step t (second register set), so a
//    wave never sits at an lgkmcnt wait with an empty matrix pipe; the barrier of a chunk moves in front of its LAST
//    k-step, whose fragments are already in registers: the first fragments of the next chunk are read, and the copy of
//    the chunk after next is started, while that k-step multiplies.
//  * A fragments: wave-uniform base (descendant panel + k * md in scalar registers) + one 32-bit lane offset per chunk
//    (the SGPR-base addressing form): no 64-bit address arithmetic per load.  k-steps past a short chunk's end re-read
//    its last 4-deep block (their B rows are the zero page's), at most 3 columns past the panel -- inside the slack the
//    factor storage keeps behind every panel for exactly this kind of over-read.
//  (Measured and dropped: a lane owning two ADJACENT rows, one 16-byte load per k -- 59.6 instead of 62.2 TFLOP/s.)
#ifndef SCILMM_DENSE_B_SG
#define SCILMM_DENSE_B_SG 1   // 1: interleave one LDS read with four MFMAs (sched_group_barrier); 0: leave it to the scheduler
#endif
__global__ __launch_bounds__(512, 1) void k_dense_b(DevSym S, int32_t dense_first, const DenseWork* __restrict__ work,
                                                    double* __restrict__ L, double* __restrict__ scratch,
                                                    const double* __restrict__ zeros) {
  static_assert(NB == 128 && DTR == 256, "k_dense_b: 8 waves x 32 rows, one B k-row per DMA instruction");
  extern __shared__ __attribute__((aligned(16))) double smem[];  // [2][KBA][LDB]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const DenseWork wk = work[blockIdx.x];
  const int32_t j = wk.front;
  const int32_t c0j = S.sn_start[j], wj = S.sn_start[j + 1] - c0j;
  const int32_t mj = S.n - c0j;
  const int32_t R0 = wk.ti0 * TM;
  const int32_t nrow = min(wk.ntiles * TM, mj - R0);
  // the two target rows of this lane inside the item (rows past the item's edge: row 0, never stored)
  const int32_t ia = 32 * wv + li, ib_ = ia + 16;
  const int32_t ra0 = R0 + (ia < nrow ? ia : 0);
  const int32_t ra1 = R0 + (ib_ < nrow ? ib_ : 0);
  const int32_t b_off = 2 * lane < wj ? 2 * lane : 0;
  const double* zsrc = zeros + 2 * lane;
  struct Chunk { const double* Pd; int32_t md; int kc; uint32_t v0, v1; };
  int32_t kd = wk.k0, kk0 = 0;
  // (the descendant's entries of the symbolic arrays are read once per descendant, not once per chunk: a descriptor built at the
  //  chunk boundary from fresh loads stalls every wave for their latency right before the last k-step's MFMAs)
  int32_t d_wd = 0, d_md = 0;
  const double* d_P0 = nullptr;  // column 0 of the current descendant's panel, at the target's first column's row
  auto next_chunk = [&]() {
    if (kk0 == 0) {
      const int32_t d = dense_first + kd;
      const int32_t c0d = S.sn_start[d];
      d_wd = __builtin_amdgcn_readfirstlane(S.sn_start[d + 1] - c0d);
      d_md = __builtin_amdgcn_readfirstlane(S.n - c0d);
      d_P0 = L + uniform_i64(S.sn_loff[d] + (c0j - c0d));
    }
    Chunk c;
    c.md = d_md;
    c.Pd = d_P0 + (int64_t)kk0 * d_md;
    c.kc = min(KBA, d_wd - kk0);
    c.v0 = (uint32_t)(lk * c.md + ra0) * 8u;  // byte offsets of this lane's rows in the k-column lk of a 4-deep block
    c.v1 = (uint32_t)(lk * c.md + ra1) * 8u;
    kk0 += KBA;
    if (kk0 >= d_wd) { kk0 = 0; ++kd; }
    return c;
  };
  auto issue_B = [&](const Chunk& c, int b) {
    double* Bs = smem + b * KBA * LDB;
    if (c.kc == KBA) {
      // full chunk (all but a descendant's last one): no depth test per k-row -- the generic form below costs a scalar compare,
      // a branch and a 64-bit address rebuild per copy instruction, between the chunk barrier and the last k-step's MFMAs
      const double* p = c.Pd + (int64_t)wv * c.md + b_off;
      const int64_t step = 8 * (int64_t)c.md;
#pragma unroll
      for (int i = 0; i < KBA / 8; ++i)
        __builtin_amdgcn_global_load_lds((gl_vptr)(p + i * step), (lds_vptr)(Bs + (wv + 8 * i) * LDB), 16, 0, 0);
      return;
    }
#pragma unroll
    for (int i = 0; i < KBA / 8; ++i) {
      const int kr = wv + 8 * i;
      __builtin_amdgcn_global_load_lds((gl_vptr)(kr < c.kc ? c.Pd + (int64_t)kr * c.md + b_off : zsrc), (lds_vptr)(Bs + kr * LDB), 16, 0, 0);
    }
  };
  auto load_A = [&](const Chunk& c, int s, double (&ra)[4][2]) {
    const int klast = (c.kc - 1) & ~3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double* sp = c.Pd + (int64_t)min(16 * s + 4 * q, klast) * c.md;  // wave-uniform
      ra[q][0] = ld_off(sp, c.v0);
      ra[q][1] = ld_off(sp, c.v1);
    }
  };
  d4 acc16[NJB][2];
#pragma unroll
  for (int a = 0; a < NJB; ++a) { acc16[a][0] = (d4){0.0, 0.0, 0.0, 0.0}; acc16[a][1] = (d4){0.0, 0.0, 0.0, 0.0}; }
  if (wk.k0 >= wk.k1) return;
  double rA[2][4][2];
  double bf[2][NJB];
  auto ldB = [&](const double* Bc, int k4, double (&b)[NJB]) {
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) b[jb] = Bc[(k4 + lk) * LDB + 16 * jb + li];
  };
  auto mma = [&](const double (&b)[NJB], double a0, double a1) {
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) {
      acc16[jb][0] = mfma_f64(b[jb], a0, acc16[jb][0]);
      acc16[jb][1] = mfma_f64(b[jb], a1, acc16[jb][1]);
    }
  };
  Chunk cur = next_chunk();
  issue_B(cur, 0);
  load_A(cur, 0, rA[0]);
  bool more = kd < wk.k1;
  Chunk nxt = cur;
  if (more) {
    nxt = next_chunk();
    issue_B(nxt, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  ldB(smem, 0, bf[0]);
  while (true) {
    const double* Bc = smem + buf * KBA * LDB;
    const double* Bn = smem + (buf ^ 1) * KBA * LDB;
    bool more2 = false;
    Chunk nn = nxt;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int s = t >> 2, q = t & 3;
      if (q == 0) {
        // the A fragments of the next sub-chunk (or of the next chunk's first one): one sub-chunk of MFMA time ahead
        if (s < 3) load_A(cur, s + 1, rA[(s + 1) & 1]);
        else if (more) load_A(nxt, 0, rA[0]);
      }
      if (t < 15) {
        ldB(Bc, 4 * (t + 1), bf[(t + 1) & 1]);
      } else if (more) {
        // chunk boundary: every wave has READ its last fragments of Bc (they are in registers) and the copy of the next
        // chunk has landed -- after this barrier Bn may be read and Bc overwritten
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ldB(Bn, 0, bf[0]);
        more2 = kd < wk.k1;
        if (more2) {
          nn = next_chunk();
          issue_B(nn, buf);
        }
      }
      mma(bf[t & 1], rA[s & 1][q][0], rA[s & 1][q][1]);
#if SCILMM_DENSE_B_SG
      if (t < 15) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one LDS read (two fragments) ...
          __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);  // ... behind four MFMAs
        }
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!more) break;
    cur = nxt;
    nxt = nn;
    more = more2;
    buf ^= 1;
  }
  double* P = L + S.sn_loff[j];
#pragma unroll
  for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ib == 0 ? ia : ib_, jc = 16 * jb + lk + 4 * r;
        const double v = acc16[jb][ib][r];
        const int h = i >> 7;
        const int32_t slot = h ? wk.slot1 : wk.slot0;
        if (slot < 0) {
          if (i < nrow && jc < wj) P[(int64_t)jc * mj + R0 + i] -= v;
        } else if (h < wk.ntiles) {
          scratch[(int64_t)slot * (TM * NB) + jc * TM + (i & (TM - 1))] = v;
        }
      }
}

// ------------------------------------------------------------------------------------------------
// k_dense32: the dense-tail update with the PRODUCTS on the fp32 matrix pipe and the SUMS in fp64 -- the
// "fp64 factor with fp32 MFMA fronts" of BASELINE configs[4] (opt-in: scilmm_set_front_precision(sym, 32)).
// Operands are rounded to fp32 when they are staged into LDS (the image is half as large), a 16-deep chunk is
// multiplied with v_mfma_f32_16x16x4_f32 into fp32 accumulators, and after every chunk those are folded into the
// fp64 accumulators that live across the whole item (so the fp32 error does not grow with K: each chunk contributes
// a relative error of ~1e-6 of its own magnitude, the fold and everything after it -- the subtraction from the
// panel, k_potrf, k_trsm, the solves -- is fp64).  The factor then carries a relative backward error of ~1e-7;
// scilmm_amd.factor.Factor repairs the solves by iterative refinement against the exact V (fp64 SpMM on the device).
// Same work items, slabs and epilogue contract as k_dense.  fp32 MFMA result layout (differs from the f64 form):
// lane l, register r holds D[M = 4 (l >> 4) + r][N = l & 15].
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int LDA2F = DTR + 16;   // floats; == 16 mod 64: the four k rows of a fragment read hit disjoint banks
constexpr int LDBF = NB + 16;

__global__ __launch_bounds__(512, 1) void k_dense32(DevSym S, int32_t dense_first, const DenseWork* __restrict__ work,
                                                    double* __restrict__ L, double* __restrict__ scratch) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  float* Abuf = (float*)smem;                 // [2][KC][LDA2F]
  float* Bbuf = Abuf + 2 * KC * LDA2F;        // [2][KC][LDBF]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const DenseWork wk = work[blockIdx.x];
  const int32_t j = wk.front;
  const int32_t c0j = S.sn_start[j], wj = S.sn_start[j + 1] - c0j;
  const int32_t mj = S.n - c0j;
  const int32_t R0 = wk.ti0 * TM;
  const int32_t nrow = min(wk.ntiles * TM, mj - R0);
  typedef double d2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  constexpr int NPA = KC / 4, NPB = KC / 8;
  const int pa = tid & (DTR / 2 - 1), ka = tid >> 7;
  const int pb = tid & (NB / 2 - 1), kb = tid >> 6;
  const int ra_row = 2 * pa < nrow ? 2 * pa : 0, rb_col = 2 * pb < wj ? 2 * pb : 0;  // see k_dense
  d2 ra[NPA], rb[NPB];
  int32_t kd = wk.k0, kk0 = 0;
  int kc_ld = 0;
  auto load_chunk = [&]() {
    const int32_t d = dense_first + kd;
    const int32_t c0d = S.sn_start[d], wd = S.sn_start[d + 1] - c0d;
    const int64_t md = S.n - c0d;
    const double* Pd = L + S.sn_loff[d] + (int64_t)kk0 * md + (c0j - c0d);
    kc_ld = min(KC, wd - kk0);
#pragma unroll
    for (int i = 0; i < NPA; ++i) __builtin_memcpy(&ra[i], Pd + (int64_t)min(ka + 4 * i, kc_ld - 1) * md + R0 + ra_row, 16);
#pragma unroll
    for (int i = 0; i < NPB; ++i) __builtin_memcpy(&rb[i], Pd + (int64_t)min(kb + 8 * i, kc_ld - 1) * md + rb_col, 16);
    kk0 += KC;
    if (kk0 >= wd) { kk0 = 0; ++kd; }
  };
  auto store_chunk = [&](int b) {
    float* As = Abuf + b * KC * LDA2F;
    float* Bs = Bbuf + b * KC * LDBF;
#pragma unroll
    for (int i = 0; i < NPA; ++i) {
      const bool on = ka + 4 * i < kc_ld;
      *(f2*)&As[(ka + 4 * i) * LDA2F + 2 * pa] = (f2){on ? (float)ra[i][0] : 0.0f, on ? (float)ra[i][1] : 0.0f};
    }
#pragma unroll
    for (int i = 0; i < NPB; ++i) {
      const bool on = kb + 8 * i < kc_ld;
      *(f2*)&Bs[(kb + 8 * i) * LDBF + 2 * pb] = (f2){on ? (float)rb[i][0] : 0.0f, on ? (float)rb[i][1] : 0.0f};
    }
  };
  // wave wv: rows [32 wv, 32 wv + 32) x all 128 columns; fp64 accumulators across the item, fp32 ones per chunk
  d4 acc[NJB][2];
#pragma unroll
  for (int a = 0; a < NJB; ++a) { acc[a][0] = (d4){0.0, 0.0, 0.0, 0.0}; acc[a][1] = (d4){0.0, 0.0, 0.0, 0.0}; }
  const int li = lane & 15, lk = lane >> 4;
  if (wk.k0 >= wk.k1) return;
  load_chunk();
  int kc_cur = kc_ld;
  store_chunk(0);
  __syncthreads();
  int buf = 0;
  while (true) {
    const bool more = kd < wk.k1;
    if (more) load_chunk();
    const float* Ac = Abuf + buf * KC * LDA2F;
    const float* Bc = Bbuf + buf * KC * LDBF;
    const int kc4 = (kc_cur + 3) & ~3;
    f4 c32[NJB][2];
#pragma unroll
    for (int a = 0; a < NJB; ++a) { c32[a][0] = (f4){0.f, 0.f, 0.f, 0.f}; c32[a][1] = (f4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 2
    for (int k4 = 0; k4 < kc4; k4 += 4) {
      const float a0 = Ac[(k4 + lk) * LDA2F + 32 * wv + li], a1 = Ac[(k4 + lk) * LDA2F + 32 * wv + 16 + li];
      float b[NJB];
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb) b[jb] = Bc[(k4 + lk) * LDBF + 16 * jb + li];
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb) {
        c32[jb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[jb], a0, c32[jb][0], 0, 0, 0);
        c32[jb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[jb], a1, c32[jb][1], 0, 0, 0);
      }
    }
    // fold the chunk's fp32 sums into the fp64 accumulators
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[jb][ib][r] += (double)c32[jb][ib][r];
    if (!more) break;
    store_chunk(buf ^ 1);
    kc_cur = kc_ld;
    __syncthreads();
    buf ^= 1;
  }
  double* P = L + S.sn_loff[j];
#pragma unroll
  for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 32 * wv + 16 * ib + li, jc = 16 * jb + 4 * lk + r;  // fp32 MFMA layout: M = 4 (l >> 4) + r
        const int h = i >> 7;
        const int32_t slot = h ? wk.slot1 : wk.slot0;
        if (slot < 0) {
          if (i < nrow && jc < wj) P[(int64_t)jc * mj + R0 + i] -= acc[jb][ib][r];
        } else if (h < wk.ntiles) {
          scratch[(int64_t)slot * (TM * NB) + jc * TM + (i & (TM - 1))] = acc[jb][ib][r];
        }
      }
}

// ------------------------------------------------------------------------------------------------
// k_dense_h: the fp32-product dense-tail update, round 3's final form.  The four earlier forms (k_dense32 above; three rewrites in
// csrc/tools/retired_kernels.hip.h) all run at 83 - 88 TFLOP/s, 0.55 of the fp32 matrix pipe, whatever their operand format --
// while a synthetic loop with the same operand feeds sustains 114 - 124 on a launch of the real shape (csrc/tools/mfma_f32_feed,
// mfma_f32_wave64; profiles/r3_mfma_f32_feed.txt).  What the real kernels have and the synthetic loop has not is REGISTER
// PRESSURE: a wave that owns 32 rows x 128 columns needs 128 registers of fp64 sums + 64 of fp32 sums + fragments + addressing,
// more than the 256 a wave has at two waves per SIMD; the spilled accumulators come back from scratch -- through the memory
// pipeline the 1 - 2 TB/s operand stream is using -- exactly where a chunk's fold needs them (k_dense_s without its fp64
// accumulators: 134 registers, 118 TFLOP/s instead of 86; with the whole register file for one wave per SIMD and 64 x 128 per
// wave the compiler still spills: 69).  Here a wave owns 32 rows x 64 COLUMNS -- the eight waves tile a 128-row x 128-column
// workgroup tile 4 x 2 -- so the sums take 64 + 32 registers and nothing spills.  Two workgroups serve one work item (its two
// 128-row tiles); HBM bytes per flop are unchanged (they depend on the columns per workgroup), the two column halves of a row
// group load the same A rows (the second load hits L1 / L2).
// Operands come from an fp32 SHADOW of the finished tail panels (k_shadow right after a panel's k_trsm, and behind the broadcast
// of a panel that arrives from another rank; same column-major layout and rank-local offsets, + 50 % tail storage): half the bytes of the fp64 panels per flop, no conversion in the loop.  Values = k_dense32's
// operands (one rounding of the finished fp64 entry); products on the fp32 matrix pipe; sums folded into fp64 every 256 k; the
// subtraction from the fp64 panel, k_potrf, k_trsm and the solves stay fp64.  Loop = k_dense_b's: A fragments (a lane's two
// adjacent rows, one 8-byte load per k) straight into registers one 16-deep sub-chunk ahead, B image (64 k-rows x 128 columns)
// by LDS-DMA 4 bytes per lane, fragments prefetched one k-step ahead, one barrier per 64 k; the item's descendants' symbolic
// entries sit in an LDS table.  Same work items, slabs and epilogue contract as k_dense_b / k_dense32.
__global__ __launch_bounds__(256) void k_shadow(const double* __restrict__ src, float* __restrict__ dst, int64_t cnt) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * 256) dst[i] = (float)src[i];
}

#ifndef SCILMM_DENSE_H_FOLD
#define SCILMM_DENSE_H_FOLD 4   // chunks of 64 k between two folds of the fp32 sums into the fp64 accumulators (a power of two)
#endif
#ifndef SCILMM_DENSE_H_WGS
#define SCILMM_DENSE_H_WGS 4    // waves per SIMD the register budget is cut for (4 = two workgroups per CU: 128 registers per wave)
#endif
constexpr int DH_MAXD = 256;  // descendants per item held in the LDS table (the plan's items have <= 64)
constexpr size_t dense_h_lds = sizeof(float) * 2 * KBA * LDBF + sizeof(int32_t) * (DH_MAXD + 2) + sizeof(int64_t) * DH_MAXD;

__global__ __launch_bounds__(512, SCILMM_DENSE_H_WGS) void k_dense_h(DevSym S, int32_t dense_first, const DenseWork* __restrict__ work,
                                                    double* __restrict__ L, const float* __restrict__ L32, int64_t base32,
                                                    double* __restrict__ scratch, const float* __restrict__ zeros32) {
  static_assert(NB == 128 && TM == 128, "k_dense_h: 4 x 2 waves of 32 rows x 64 columns");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  float* Bimg = (float*)smem;                            // [2][KBA][LDBF]
  int32_t* t_c0 = (int32_t*)(Bimg + 2 * KBA * LDBF);     // [DH_MAXD + 2]  first columns of the item's descendants
  int64_t* t_lo = (int64_t*)(t_c0 + DH_MAXD + 2);        // [DH_MAXD]      their panel offsets
  typedef float f2 __attribute__((ext_vector_type(2)));
  constexpr int NJH = NJB / 2;  // 16-column blocks per wave
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wv >> 1, ch = wv & 1;  // row group (32 rows) / column half (64 columns) of this wave
  const int li = lane & 15, lk = lane >> 4;
  const DenseWork wk = work[blockIdx.x >> 1];
  const int th = blockIdx.x & 1;        // which of the item's two 128-row tiles
  if (th >= wk.ntiles || wk.k0 >= wk.k1) return;
  const int32_t j = wk.front;
  const int32_t c0j = S.sn_start[j], wj = S.sn_start[j + 1] - c0j;
  const int32_t mj = S.n - c0j;
  const int32_t R0 = (wk.ti0 + th) * TM;
  const int32_t nrow = min(TM, mj - R0);
  const int32_t slot = th ? wk.slot1 : wk.slot0;
  const int32_t ndesc = min(wk.k1 - wk.k0, DH_MAXD);
  for (int i = tid; i <= ndesc; i += 512) t_c0[i] = S.sn_start[dense_first + wk.k0 + i];
  for (int i = tid; i < ndesc; i += 512) t_lo[i] = S.sn_loff[dense_first + wk.k0 + i];
  const int32_t ia = 32 * rg + 2 * li;                    // this lane's rows inside the tile: ia, ia + 1
  const int32_t ra = R0 + (ia < nrow ? ia : 0);           // (rows past the tile's edge: the pair 0, 1, never stored; a pair that
                                                          //  straddles the edge reads one entry past it into a row that is never stored)
  const int32_t bcol0 = lane < wj ? lane : 0, bcol1 = 64 + lane < wj ? 64 + lane : 0;
  __syncthreads();
  struct Chunk { const float* Pd; int32_t md; int kc; };
  int32_t kd = wk.k0, kk0 = 0;
  auto next_chunk = [&]() {
    const int32_t e = kd - wk.k0, d = dense_first + kd;
    const bool tab = e < DH_MAXD;
    const int32_t c0d = __builtin_amdgcn_readfirstlane(tab ? t_c0[e] : S.sn_start[d]);
    const int32_t wd = __builtin_amdgcn_readfirstlane(tab ? t_c0[e + 1] : S.sn_start[d + 1]) - c0d;
    Chunk c;
    c.md = S.n - c0d;
    c.Pd = L32 + (uniform_i64(tab ? t_lo[e] : S.sn_loff[d]) - base32 + (int64_t)kk0 * c.md + (c0j - c0d));
    c.kc = min(KBA, wd - kk0);
    kk0 += KBA;
    if (kk0 >= wd) { kk0 = 0; ++kd; }
    return c;
  };
  auto issue_B = [&](const Chunk& c, int b) {
    float* Bs = Bimg + b * KBA * LDBF;
#pragma unroll
    for (int i = 0; i < KBA / 8; ++i) {
      const int kr = wv + 8 * i;
      const float* row = c.Pd + (int64_t)kr * c.md;
      const bool on = kr < c.kc;
      __builtin_amdgcn_global_load_lds((gl_vptr)(on ? row + bcol0 : zeros32 + lane), (lds_vptr)(Bs + kr * LDBF), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((gl_vptr)(on ? row + bcol1 : zeros32 + lane), (lds_vptr)(Bs + kr * LDBF + 64), 4, 0, 0);
    }
  };
  auto load_A = [&](const Chunk& c, int s, f2 (&a)[4]) {
    const int klast = (c.kc - 1) & ~3;
    const uint32_t v = (uint32_t)(lk * c.md + ra) * 4u;  // byte offset of this lane's row pair in k-column lk of a 4-deep block
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float* sp = c.Pd + (int64_t)min(16 * s + 4 * q, klast) * c.md;  // wave-uniform
      a[q] = *(const f2*)((const char*)sp + v);
    }
  };
  d4 acc[NJH][2];
  f4 c32[NJH][2];
#pragma unroll
  for (int a = 0; a < NJH; ++a) {
    acc[a][0] = (d4){0.0, 0.0, 0.0, 0.0}; acc[a][1] = (d4){0.0, 0.0, 0.0, 0.0};
    c32[a][0] = (f4){0.f, 0.f, 0.f, 0.f}; c32[a][1] = (f4){0.f, 0.f, 0.f, 0.f};
  }
  f2 rA[2][4];
  float bf[2][NJH];
  int cidx = 0;  // chunks multiplied so far
  auto ldB = [&](const float* Bc, int k4, float (&b)[NJH]) {
#pragma unroll
    for (int jb = 0; jb < NJH; ++jb) b[jb] = Bc[(k4 + lk) * LDBF + 64 * ch + 16 * jb + li];
  };
  Chunk cur = next_chunk();
  issue_B(cur, 0);
  load_A(cur, 0, rA[0]);
  bool more = kd < wk.k1;
  Chunk nxt = cur;
  if (more) {
    nxt = next_chunk();
    issue_B(nxt, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  ldB(Bimg, 0, bf[0]);
  while (true) {
    const float* Bc = Bimg + buf * KBA * LDBF;
    const float* Bn = Bimg + (buf ^ 1) * KBA * LDBF;
    bool more2 = false;
    Chunk nn = nxt;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int s = t >> 2, q = t & 3;
      if (q == 0) {
        if (s < 3) load_A(cur, s + 1, rA[(s + 1) & 1]);
        else if (more) load_A(nxt, 0, rA[0]);
      }
      if (t < 15) {
        ldB(Bc, 4 * (t + 1), bf[(t + 1) & 1]);
      } else if (more) {
        // chunk boundary: every wave has READ its last fragments of Bc (they are in registers) and the copy of the next
        // chunk has landed -- after this barrier Bn may be read and Bc overwritten
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ldB(Bn, 0, bf[0]);
        more2 = kd < wk.k1;
        if (more2) {
          nn = next_chunk();
          issue_B(nn, buf);
        }
      }
      const f2 a = rA[s & 1][q];
#pragma unroll
      for (int jb = 0; jb < NJH; ++jb) {
        c32[jb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[t & 1][jb], a[0], c32[jb][0], 0, 0, 0);
        c32[jb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[t & 1][jb], a[1], c32[jb][1], 0, 0, 0);
      }
#ifndef SCILMM_DENSE_H_NOFOLD  // (fold experiment of csrc/tools/dense_bench2: fp32 sums over the whole item, WRONG error model)
      if (t == 15 && (cidx & (SCILMM_DENSE_H_FOLD - 1)) == SCILMM_DENSE_H_FOLD - 1) {
        // fold the fp32 sums into the fp64 accumulators every SCILMM_DENSE_H_FOLD chunks of 64 k.  Between two folds the fp64
        // accumulators are dead weight: at the 128-register budget of two workgroups per CU the compiler parks part of them in
        // scratch and fetches them here, so the fold period sets the price (alone, 1M-shaped launch: 64 / 128 / 256 / 512 k =
        // 72 / 103.7 / 110.9 / 114.1 TFLOP/s, no fold at all 117; error of a 6144-deep item against fp64 at entries of 2e3:
        // 9e-5 / 2.1e-4 / 4.6e-4 / 6.9e-4).  At one workgroup per CU (156 registers, nothing spilled) the period hardly
        // matters (94.5 ... 98.0) and staggering the two waves of a SIMD by eight k-steps does not help (92.6).
#pragma unroll
        for (int jb = 0; jb < NJH; ++jb)
#pragma unroll
          for (int ib = 0; ib < 2; ++ib) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[jb][ib][r] += (double)c32[jb][ib][r];
            c32[jb][ib] = (f4){0.f, 0.f, 0.f, 0.f};
          }
      }
#endif
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!more) break;
    cur = nxt;
    nxt = nn;
    more = more2;
    buf ^= 1;
    ++cidx;
  }
#ifndef SCILMM_DENSE_H_NOFOLD
#pragma unroll
  for (int jb = 0; jb < NJH; ++jb)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[jb][ib][r] += (double)c32[jb][ib][r];
#endif
  double* P = L + S.sn_loff[j];
#pragma unroll
  for (int jb = 0; jb < NJH; ++jb)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ia + ib, jc = 64 * ch + 16 * jb + 4 * lk + r;  // fp32 MFMA layout: M = 4 (l >> 4) + r
#ifdef SCILMM_DENSE_H_NOFOLD
        const double v = (double)c32[jb][ib][r];
#else
        const double v = acc[jb][ib][r];
#endif
        if (slot < 0) {
          if (i < nrow && jc < wj) P[(int64_t)jc * mj + R0 + i] -= v;
        } else {
          scratch[(int64_t)slot * (TM * NB) + jc * TM + i] = v;
        }
      }
}

// ------------------------------------------------------------------------------------------------
// k_outside: the contribution of a PRELUDE front to the dense tail, computed in the descendant's own coordinates.
// The rows of descendant d that lie in the tail (rows t0 .. m_d of its panel) are contiguous in the panel, so the
// products  U = L_d[t0:, :] * L_d[t0:, :]^T  are plain dense 128 x 128 x w_d block products with no padding at all
// -- where the target-coordinate kernels spend a whole 128 x 128 x 16 chunk on a piece that typically is 42 x 32
// (22 % of their MFMA slots useful; 17 % of a 300k / 1M factorization).  Only the DESTINATION is scattered: entry
// (i, j) of a block belongs to row label r_i, column label r_j of the tail, whose address is arithmetic because every
// tail panel has every later column as a row.  Several descendants hit the same cells, so the subtraction is an fp64
// atomic add (global_atomic_add_f64, no return): the sum order -- hence the last bits -- is not reproducible from run
// to run; SCILMM_DETERMINISTIC=1 keeps these updates on the target-coordinate path (fixed order).
// One workgroup (4 waves, wave = 32 rows x 128 columns) per lower block pair (bi >= bj) of one descendant; the
// launch runs while nothing else touches the tail panels (after the last prelude level, before the first tail level).
struct OutsideWork {
  int32_t d;        // descendant front (below the dense tail)
  int32_t t0;       // first row of its panel that lies in the tail
  int32_t bi, bj;   // 128-row blocks of those rows: target rows / target columns
};

// SCATTER: 0 = atomic subtraction (the product), 1 = nothing, 2 = plain stores (timing ablations of diagnostic builds)
template <bool MFMA, int SCATTER = 0>
__global__ __launch_bounds__(256) void k_outside(DevSym S, int32_t dense_first, const OutsideWork* __restrict__ work,
                                                 const int32_t* __restrict__ tail_front,  // tail column (label - c0_tail) -> front
                                                 const uint8_t* __restrict__ keep_front,  // multi-GPU: fronts this rank computes
                                                 const int32_t* __restrict__ grp_next,    // next descendant with the SAME tail rows, or -1
                                                 const int32_t* __restrict__ grp_t0,      // its first tail row
                                                 double* __restrict__ L) {
  __shared__ __attribute__((aligned(16))) double As[KCS * LDA];  // [k][target-row entry]
  __shared__ __attribute__((aligned(16))) double Bs[KCS * LDB];  // [k][target-column entry]
  __shared__ int32_t lab_i[TM], lab_j[NB];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const OutsideWork wk = work[blockIdx.x];
  int32_t ni, nj;
  {
    // the destination labels: the leader's tail rows (every member of its group has the very same ones)
    const int32_t d = wk.d;
    const int32_t md = (int32_t)(S.sn_rowptr[d + 1] - S.sn_rowptr[d]);
    const int32_t* rd = S.sn_rows + S.sn_rowptr[d];
    const int32_t ri0 = wk.t0 + TM * wk.bi, rj0 = wk.t0 + NB * wk.bj;  // first panel row of the two blocks
    ni = min(TM, md - ri0);
    nj = min(NB, md - rj0);
    if (tid < TM) lab_i[tid] = tid < ni ? rd[ri0 + tid] : -1;
    else if (tid < TM + NB) lab_j[tid - TM] = (tid - TM) < nj ? rd[rj0 + tid - TM] : -1;
  }
  d4 acc[NJB][2];
#pragma unroll
  for (int a = 0; a < NJB; ++a) { acc[a][0] = (d4){0.0, 0.0, 0.0, 0.0}; acc[a][1] = (d4){0.0, 0.0, 0.0, 0.0}; }
  // staging: thread (x = tid & 127, kk = tid >> 7) carries k = kk, kk + 2, ... of row x of both blocks
  const int x = tid & 127, kk = tid >> 7;
  bool first_chunk = true;
  // (round 4) descendants with IDENTICAL tail rows -- the 128-column blocks a wide supernode was cut into -- hit the very same
  // cells: their products are summed in the registers, K = the group's columns, and scattered ONCE (a quarter fewer atomic bytes
  // at the 100k config: 61 549 -> 45 309 block pairs)
  for (int32_t d = wk.d, t0 = wk.t0; d >= 0;) {
    const int32_t wd = S.sn_start[d + 1] - S.sn_start[d];
    const int32_t md = (int32_t)(S.sn_rowptr[d + 1] - S.sn_rowptr[d]);
    const double* Pd = L + S.sn_loff[d];
    const int32_t ri0 = t0 + TM * wk.bi, rj0 = t0 + NB * wk.bj;
    const double* pa = Pd + ri0 + min(x, ni - 1);
    const double* pb = Pd + rj0 + min(x, nj - 1);
    for (int32_t k0 = 0; k0 < wd; k0 += KCS) {
      const int kc = min(KCS, wd - k0);
      const int kc4 = (kc + 3) & ~3;
      double va[KCS / 2], vb[KCS / 2];
#pragma unroll
      for (int i = 0; i < KCS / 2; ++i) {
        const int64_t kq = k0 + min(kk + 2 * i, kc - 1);
        va[i] = pa[kq * md];
        vb[i] = pb[kq * md];
      }
      if (!first_chunk) __syncthreads();  // the previous chunk has been consumed
      first_chunk = false;
#pragma unroll
      for (int i = 0; i < KCS / 2; ++i) {
        const int k = kk + 2 * i;
        if (k < kc4) {
          As[k * LDA + x] = (x < ni && k < kc) ? va[i] : 0.0;
          Bs[k * LDB + x] = (x < nj && k < kc) ? vb[i] : 0.0;
        }
      }
      __syncthreads();
      if (32 * wv < ni) tile_mma<MFMA>(As, Bs, kc4, NJB, lane, wv, acc);
    }
    d = grp_next[d];
    if (d >= 0) t0 = grp_t0[d];
  }
  // scatter: acc[jb][ib][r] = U(i, j) with i = 32 wv + 16 ib + (lane & 15), j = 16 jb + (lane >> 4) + 4 r
  const int li = lane & 15, lr = lane >> 4;
  const int32_t c0_tail = S.sn_start[dense_first];
#pragma unroll
  for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jloc = 16 * jb + lr + 4 * r;
      const int32_t cj = lab_j[jloc];
      if (cj < 0) continue;
      const int32_t f = tail_front[cj - c0_tail];
      if (!keep_front[f]) continue;
      const int32_t c0f = S.sn_start[f];
      double* col = L + S.sn_loff[f] + (int64_t)(cj - c0f) * (S.n - c0f) - c0f;  // + row label = the cell
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) {
        const int iloc = 32 * wv + 16 * ib + li;
        const int32_t ci = lab_i[iloc];
        if (SCATTER == 0) {
          if (ci >= cj) unsafeAtomicAdd(col + ci, -acc[jb][ib][r]);  // lower triangle only (ci = -1 for padding rows)
        } else if (SCATTER == 1) {
          if (ci >= cj && acc[jb][ib][r] == 123.456) col[ci] = 0.0;
        } else {
          if (ci >= cj) col[ci] = -acc[jb][ib][r];
        }
      }
    }
}


// ------------------------------------------------------------------------------------------------
// Cell-wise path for the small update pairs (a few rows x a few columns of a narrow descendant): the host
// groups every contributed target cell by address; one thread owns one target cell and subtracts the dot
// products  sum_k L_d[t,k] L_d[q,k]  of all its contributions in a fixed order.  No LDS, no barriers, no
// atomics, bitwise reproducible.  Integer/latency-bound gather work, kept off the MFMA pipeline.
__device__ __forceinline__ double cell_dot(const double* __restrict__ L, int64_t st, int64_t sq, int64_t md, int wd) {
  // latency-bound gather: eight independent element pairs in flight per batch, fixed summation order
  const double* pt = L + st;
  const double* pq = L + sq;
  double acc = 0.0;
  int k = 0;
  for (; k + 8 <= wd; k += 8) {
    double a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = pt[(int64_t)(k + u) * md];
      b[u] = pq[(int64_t)(k + u) * md];
    }
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int u = 0; u < 8; u += 2) {
      s0 += a[u] * b[u];
      s1 += a[u + 1] * b[u + 1];
    }
    acc += s0 + s1;
  }
  if (k < wd) {
    double a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int kk = min(k + u, wd - 1);  // clamped: the loads stay unconditional and independent
      a[u] = pt[(int64_t)kk * md];
      b[u] = pq[(int64_t)kk * md];
    }
    double s0 = 0.0;
#pragma unroll
    for (int u = 0; u < 8; ++u) s0 += (k + u < wd) ? a[u] * b[u] : 0.0;
    acc += s0;
  }
  return acc;
}

struct CellSrc { int64_t st, sq; int32_t md, wd; };

// groups [first, first+n_short) : one thread each;  groups [first+n_short, first+count) : one wave each
__global__ __launch_bounds__(256) void k_sparse_cells(int64_t first, int64_t n_short, int64_t count,
                                                      const int64_t* __restrict__ uniq_dst,
                                                      const int64_t* __restrict__ grp_ptr,
                                                      const int64_t* __restrict__ src_t,
                                                      const int64_t* __restrict__ src_q,
                                                      const int32_t* __restrict__ src_md,
                                                      const int32_t* __restrict__ src_wd, double* __restrict__ L) {
  const int64_t short_blocks = (n_short + 255) / 256;
  if ((int64_t)blockIdx.x < short_blocks) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= n_short) return;
    const int64_t g = first + u;
    const int64_t c0 = grp_ptr[g], c1 = grp_ptr[g + 1];
    double acc = 0.0;
    // the descriptor of the next contribution is fetched while the current dot product runs
    CellSrc cur{src_t[c0], src_q[c0], src_md[c0], src_wd[c0]};
    for (int64_t c = c0; c < c1; ++c) {
      const int64_t cn = min(c + 1, c1 - 1);
      const CellSrc nxt{src_t[cn], src_q[cn], src_md[cn], src_wd[cn]};
      acc += cell_dot(L, cur.st, cur.sq, cur.md, cur.wd);
      cur = nxt;
    }
    L[uniq_dst[g]] -= acc;
  } else {
    const int lane = threadIdx.x & 63;
    const int64_t u = n_short + ((int64_t)blockIdx.x - short_blocks) * 4 + (threadIdx.x >> 6);
    if (u >= count) return;
    const int64_t g = first + u;
    double acc = 0.0;
    for (int64_t c = grp_ptr[g] + lane; c < grp_ptr[g + 1]; c += 64) acc += cell_dot(L, src_t[c], src_q[c], src_md[c], src_wd[c]);
    // fixed-shape butterfly: the same summation tree on every run
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) L[uniq_dst[g]] -= acc;
  }
}

// ------------------------------------------------------------------------------------------------
// Fold the split-K partial products of the update kernel into the panel:  P[tile] -= sum_seg part[seg].
// One workgroup per (tile, 16-column group); thread = tile row.  Fixed summation order (reproducible).
__global__ __launch_bounds__(128) void k_reduce(DevSym S, const int32_t* __restrict__ red_tiles,
                                                const int32_t* __restrict__ tile_pslot,
                                                const int32_t* __restrict__ tile_pnseg, const double* __restrict__ scratch,
                                                double* __restrict__ L) {
  const int32_t g = red_tiles[blockIdx.x / (NB / 4)];
  const int jg = (blockIdx.x % (NB / 4)) * 4;
  const int32_t s = S.tile_front[g];
  const int32_t ti = (int32_t)(g - S.tile_base[s]);
  const int32_t w = S.sn_start[s + 1] - S.sn_start[s];
  const int32_t m = (int32_t)(S.sn_rowptr[s + 1] - S.sn_rowptr[s]);
  const int32_t R0 = ti * TM;
  const int t = threadIdx.x;
  if (R0 + t >= m || jg >= w) return;
  const int32_t ps = tile_pslot[g], pn = tile_pnseg[g];
  double* P = L + S.sn_loff[s] + R0 + t + (int64_t)jg * m;
  const double* sp = scratch + (int64_t)ps * (TM * NB) + jg * TM + t;
  const int jend = min(4, w - jg);
  double v[4], e[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int u = 0; u < 4; ++u) v[u] = (u < jend) ? P[(int64_t)u * m] : 0.0;
  // two interleaved accumulator sets keep 8 loads in flight; the order of additions is fixed
  int sg = 0;
  for (; sg + 2 <= pn; sg += 2) {
    const double* q0 = sp + (int64_t)sg * (TM * NB);
    const double* q1 = q0 + TM * NB;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u < jend) {
        v[u] -= q0[u * TM];
        e[u] -= q1[u * TM];
      }
  }
  if (sg < pn) {
    const double* q0 = sp + (int64_t)sg * (TM * NB);
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (u < jend) v[u] -= q0[u * TM];
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (u < jend) P[(int64_t)u * m] = v[u] + e[u];
}

// ------------------------------------------------------------------------------------------------
// Dense Cholesky of the w x w (w <= NB) diagonal block of each front of a level plus its explicit
// inverse (every later triangular solve with this block becomes an MFMA GEMM) and sum(log diag).
// Blocked by 16: the 16 x 16 diagonal blocks are factored and inverted by ONE wave in registers (pivot column
// broadcast with v_readlane, factor and inverse steps interleaved), one block column AHEAD of the trailing
// update the other three waves are finishing; panel solve and trailing update are MFMA products out of the
// folded LDS triangle; the off-diagonal blocks of the inverse stay in registers (block column per wave).
// value of x in lane `src` (wave-uniform src) as a scalar broadcast: two v_readlane_b32, no LDS round trip
__device__ __forceinline__ double bc_lane(double x, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(x), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src);
  return __hiloint2double(hi, lo);
}

#ifdef SCILMM_POTRF_PROF
__device__ unsigned long long g_potrf_prof[16];
#define PPROF(i) do { if (w == NB && tid == 0) { unsigned long long t_ = wall_clock64(); atomicAdd(&g_potrf_prof[i], t_ - tprev_); tprev_ = t_; } } while (0)
#else
#define PPROF(i) do {} while (0)
#endif
// (round 3) The split-K partial slabs of the level's LATE update are folded in HERE, on load, and in k_trsm -- not by a
// k_reduce launch in between: one launch and one round trip of the panel less on the per-level chain of the main stream
// (104 us of 430 us per level at the 100k config).  A tile's slabs are summed in slab order: fixed, reproducible.
__global__ __launch_bounds__(256) void k_potrf(DevSym S, const int32_t* __restrict__ fronts, double* __restrict__ L,
                                               double* __restrict__ invD, double* __restrict__ logd,
                                               int32_t* __restrict__ status, const int32_t* __restrict__ tile_pslot,
                                               const int32_t* __restrict__ tile_pnseg, const double* __restrict__ slabs) {
  // LDS budget: the look-ahead keeps two 74 KB update workgroups resident on every CU, so a kernel of the main
  // stream can only start where ONE of them has retired: <= 85 KB.  Only the lower triangle of L is kept, folded
  // into NB/2 rows of NB+1 doubles (row i >= NB/2 holds L[i][0..i]; the rest of that row holds row NB-1-i), plus
  // the eight 16 x 16 diagonal inverses: 83.5 KB.  The off-diagonal blocks of the inverse never touch LDS (phase 2).
  constexpr int LDF = NB + 1, HB = NB / 2, XS = 16 * 17;
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* F = smem;                 // [HB][LDF] folded lower triangle
  double* Xd = smem + HB * LDF;     // [NJB][16][17]  Xd[kb][n][k] = X(16 kb + n, 16 kb + k), zero above the diagonal
  double* piv = Xd + NJB * XS;      // [32] pivot column of L / pivot row of X of the diagonal-block step in flight (wave 0)
#define FA(i, k) (((i) >= HB) ? ((i) - HB) * LDF + (k) : (HB - 1 - (i)) * LDF + (NB - (i)) + (k))
  __builtin_amdgcn_s_setprio(3);  // latency chain of the main stream: issue ahead of the co-resident update waves
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int32_t s = fronts[blockIdx.x];
  const int32_t c0 = S.sn_start[s], w = S.sn_start[s + 1] - c0;
  const int32_t m = (int32_t)(S.sn_rowptr[s + 1] - S.sn_rowptr[s]);
  double* P = L + S.sn_loff[s];
  double* I = invD + S.inv_off[s];
  const int nb = (w + 15) >> 4, W = nb << 4;  // padded with an identity block
  // late partial slabs of the front's first tile (it holds the whole diagonal block)
  const int64_t g0 = S.tile_base[s];
  const int32_t pn = tile_pnseg[g0];
  const double* sl0 = slabs + (int64_t)tile_pslot[g0] * (TM * NB);
#ifdef SCILMM_POTRF_PROF
  unsigned long long tprev_ = wall_clock64();
  if (w == NB && tid == 0) atomicAdd(&g_potrf_prof[15], 1ull);
#endif
  // lower triangle of the block -> LDS
  if (w == NB) {
    // full-width block (every block of a dense chain): constant strides, 32 loads in flight per thread
    constexpr int PT = NB * NB / 256 / 2;
#pragma unroll
    for (int hpass = 0; hpass < 2; ++hpass) {
      double v[PT];
#pragma unroll
      for (int u = 0; u < PT; ++u) {
        const int idx = tid + 256 * (u + PT * hpass);
        const int k = idx / NB, i = idx % NB;
        v[u] = P[(int64_t)k * m + max(i, k)];  // clamped into the lower triangle: unconditional loads
      }
      for (int sg = 0; sg < pn; ++sg) {
        const double* q = sl0 + (int64_t)sg * (TM * NB);
#pragma unroll
        for (int u = 0; u < PT; ++u) {
          const int idx = tid + 256 * (u + PT * hpass);
          const int k = idx / NB, i = idx % NB;
          v[u] -= q[k * TM + max(i, k)];
        }
      }
#pragma unroll
      for (int u = 0; u < PT; ++u) {
        const int idx = tid + 256 * (u + PT * hpass);
        const int k = idx / NB, i = idx % NB;
        if (i >= k) F[FA(i, k)] = v[u];
      }
    }
  } else {
    for (int base = tid; base < W * W; base += 256 * 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + 256 * u;
        const int k = idx / W, i = idx - k * W;
        v[u] = (i == k) ? 1.0 : 0.0;
        if (idx < W * W && i < w && k < w && i >= k) {
          v[u] = P[(int64_t)k * m + i];
          for (int sg = 0; sg < pn; ++sg) v[u] -= sl0[(int64_t)sg * (TM * NB) + k * TM + i];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + 256 * u;
        const int k = idx / W, i = idx - k * W;
        if (idx < W * W && i >= k) F[FA(i, k)] = v[u];
      }
    }
  }
  __syncthreads();
  PPROF(0);
  const int li = lane & 15, lk = lane >> 4;
  // One 16 x 16 diagonal block (factor + inverse) is the work of a single wave and the longest step of a block
  // column, so it runs one column AHEAD: while wave 0 factors block kb+1 (its column was updated first), the
  // other three waves finish the trailing update of column kb.
  auto diag16 = [&](int kb) {
    const int o = kb << 4;
    double* Xk = Xd + kb * XS;
    // ---- 16 x 16 diagonal block (factor + inverse), right-looking, on ALL 64 lanes of the wave: lane (r, q) = (lane & 15,
    //      lane >> 4) holds A(r, c) and X(c, r) for the four columns / rows c = q + 4 t.  Step j: the quarter that holds
    //      column j publishes it (and row j of X) through 32 doubles of LDS -- one write -> read round trip per step --
    //      and every lane scales and applies it to its own four entries.  (Round 2 kept a whole row per lane on 16 lanes
    //      and broadcast the pivot column with 2 x 15 v_readlane per step: 1300 readlanes, SGPR spills through
    //      v_writelane, 10 us per block -- 80 of the kernel's 113 us; profiles/r3_potrf_phases.txt.)  Same products and
    //      sums per entry as before: the factor's bits do not change.
    const int r = lane & 15, q = lane >> 4;
    double a[4], x[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = q + 4 * t;
      a[t] = (c <= r) ? F[FA(o + r, o + c)] : 0.0;
      x[t] = (c == r) ? 1.0 : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int qj = j & 3, tj = j >> 2;
      if (q == qj) {
        piv[r] = a[tj];        // A(r, j): the pivot column before scaling (rows r < j: don't-care)
        piv[16 + r] = x[tj];   // X(j, r) before scaling
      }
      __builtin_amdgcn_wave_barrier();  // (one wave: its LDS operations execute in order; this only pins the code order)
      double dj = piv[j];
      if (!(dj > 0.0) || !(dj < 1.0e300)) {
        if (lane == 0) atomicMin(status, c0 + o + j);
        dj = 1.0;
      }
      double y = rsqrt(dj);
      y = y * (1.5 - 0.5 * dj * y * y);  // one Newton step: full double precision
      const double lr = (r >= j) ? piv[r] * y : 0.0;  // L(r, j); the diagonal comes out as dj / sqrt(dj)
      const double xj = piv[16 + r] * y;              // X(j, r), final (zero above the diagonal: r > j)
      double lc[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) lc[t] = piv[q + 4 * t] * y;  // L(c, j) of this lane's columns
      __builtin_amdgcn_wave_barrier();
      if (q == qj) {
        a[tj] = lr;
        x[tj] = xj;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int c = q + 4 * t;
        if (c > j) {
          a[t] -= lr * lc[t];   // A(r,c) -= L(r,j) L(c,j)  (used for r >= c)
          x[t] -= lc[t] * xj;   // X(c,r) -= L(c,j) X(j,r)
        }
      }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int c = q + 4 * t;
      if (c <= r) F[FA(o + r, o + c)] = a[t];  // L(o+r, o+c)
      Xk[c * 17 + r] = x[t];                   // X(o+c, o+r); zero for c < r
      if (c >= r && o + c < w && o + r < w) I[(int64_t)(o + r) * w + o + c] = x[t];
    }
  };
  auto pair_update = [&](int o, int ib, int kk) {
    // C_{ib,kk} -= B_ib * B_kk^T (MFMA), lower block pairs
    const int r0 = o + 16 + 16 * ib, q0 = o + 16 + 16 * kk;
    d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int k4 = 0; k4 < 16; k4 += 4) acc = mfma_f64(F[FA(r0 + li, o + k4 + lk)], F[FA(q0 + li, o + k4 + lk)], acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = r0 + lk + 4 * r, col = q0 + li;
      if (row >= col) F[FA(row, col)] -= acc[r];
    }
  };
  if (wv == 0) diag16(0);
  __syncthreads();
  PPROF(1);
  for (int kb = 0; kb + 1 < nb; ++kb) {
    const int o = kb << 4;
    const double* Xk = Xd + kb * XS;
    const int nbr = (W - o - 16) >> 4;
    // ---- panel below: B_ib = A_ib * Dinv^T, one wave per 16-row block, MFMA, in place
    //      D[m][n] = sum_k A[m][k] Dinv[n][k];  Dinv[n][k] = X(o+n, o+k) = Xk[n][k]
    for (int ib = wv; ib < nbr; ib += 4) {
      const int r0 = o + 16 + 16 * ib;
      d4 acc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int k4 = 0; k4 < 16; k4 += 4) acc = mfma_f64(F[FA(r0 + li, o + k4 + lk)], Xk[li * 17 + k4 + lk], acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) F[FA(r0 + lk + 4 * r, o + li)] = acc[r];
    }
    __syncthreads();
    PPROF(2);
    // ---- trailing update, first the next block column (kk = 0) ...
    for (int ib = wv; ib < nbr; ib += 4) pair_update(o, ib, 0);
    __syncthreads();
    // ---- ... then wave 0 factors the next diagonal block while waves 1..3 update the other columns
    if (wv == 0) {
      diag16(kb + 1);
    } else {
      int pidx = 0;
      for (int ib = 1; ib < nbr; ++ib)
        for (int kk = 1; kk <= ib; ++kk, ++pidx)
          if (pidx % 3 == wv - 1) pair_update(o, ib, kk);
    }
    __syncthreads();
    PPROF(3);
  }
  // ---- inverse, off-diagonal blocks, one block COLUMN per wave (columns j and nb-1-j for balance):
  //      X_ij = -X_ii * sum_{kb=j}^{i-1} L_{i,kb} X_{kb,j}.  An MFMA result block D[(l>>4)+4r][l&15] is, lane by
  //      lane, exactly the B operand of the next product (k-step s takes acc[s]), so the finished blocks of a
  //      column stay in registers: no LDS traffic, no barriers; they go straight to invD.
  for (int j = 0; j < nb; ++j) {
    const int owner = (j < (nb + 1) / 2) ? (j & 3) : ((nb - 1 - j) & 3);  // long columns paired with short ones
    if (owner != wv) continue;
    d4 xb[NJB];
#pragma unroll
    for (int r = 0; r < 4; ++r) xb[0][r] = Xd[j * XS + (lk + 4 * r) * 17 + li];  // X_jj in result layout
#pragma unroll
    for (int di = 1; di < NJB; ++di) {
      const int i = j + di;
      if (i < nb) {
        d4 sacc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int dk = 0; dk < NJB; ++dk)
          if (dk < di) {
            const int kb = j + dk;
#pragma unroll
            for (int sstep = 0; sstep < 4; ++sstep)
              sacc = mfma_f64(F[FA(16 * i + li, 16 * kb + 4 * sstep + lk)], xb[dk][sstep], sacc);
          }
        d4 yacc = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int sstep = 0; sstep < 4; ++sstep) yacc = mfma_f64(Xd[i * XS + li * 17 + 4 * sstep + lk], sacc[sstep], yacc);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          xb[di][r] = -yacc[r];
          const int row = 16 * i + lk + 4 * r, col = 16 * j + li;
          if (row < w && col < w) I[(int64_t)col * w + row] = -yacc[r];
        }
      }
    }
  }
  PPROF(4);
  {
    // sum(log diag): one log per thread, wave reduction, fixed summation order
    __shared__ double lgp[4];
    double lg = 0.0;
    for (int jj = tid; jj < w; jj += 256) lg += log(F[FA(jj, jj)]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) lg += __shfl_down(lg, off, 64);
    if (lane == 0) lgp[wv] = lg;
    __syncthreads();
    if (tid == 0) logd[s] = (lgp[0] + lgp[1]) + (lgp[2] + lgp[3]);
  }
  if (w == NB) {
#pragma unroll 8
    for (int idx = tid; idx < NB * NB; idx += 256) {
      const int k = idx / NB, i = idx % NB;
      P[(int64_t)k * m + i] = (i >= k) ? F[FA(i, k)] : 0.0;
    }
  } else {
    for (int idx = tid; idx < w * w; idx += 256) {
      const int k = idx / w, i = idx - k * w;
      P[(int64_t)k * m + i] = (i >= k) ? F[FA(i, k)] : 0.0;
    }
  }
  PPROF(5);
#undef FA
}

// ------------------------------------------------------------------------------------------------
// Sub-diagonal panel solve as a GEMM with the inverse diagonal block:
//   P[i, :] <- P[i, :] * invL^T   for the rows i >= w of a 128-row tile.   D[M=j][N=i].
template <bool MFMA>
__global__ __launch_bounds__(256) void k_trsm(DevSym S, const int32_t* __restrict__ tiles, double* __restrict__ L,
                                              const double* __restrict__ invD, const int32_t* __restrict__ tile_pslot,
                                              const int32_t* __restrict__ tile_pnseg, const double* __restrict__ slabs) {
  __shared__ __attribute__((aligned(16))) double As[KCS * LDA];
  __shared__ __attribute__((aligned(16))) double Bs[KCS * LDB];
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int32_t g = tiles[blockIdx.x];
  const int32_t s = S.tile_front[g];
  const int32_t ti = (int32_t)(g - S.tile_base[s]);
  const int32_t c0 = S.sn_start[s], w = S.sn_start[s + 1] - c0;
  const int32_t m = (int32_t)(S.sn_rowptr[s + 1] - S.sn_rowptr[s]);
  const int32_t R0 = ti * TM;
  const int32_t nrow = min(TM, m - R0);
  if (R0 + nrow <= w) return;  // tile lies entirely inside the diagonal block
  const int ncb = (w + 15) >> 4;
  double* P = L + S.sn_loff[s];
  const double* I = invD + S.inv_off[s];
  d4 acc[NJB][2];
#pragma unroll
  for (int a = 0; a < NJB; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  // K is streamed in chunks of KCS through single LDS images; the global loads of chunk c+1 are issued into
  // registers before the MFMAs of chunk c (the launch is latency-bound: one tile per workgroup, ~90 per level)
  constexpr int PA = KCS / 2, PB = KCS / (256 / NB);  // values per thread and chunk: A row t, B column q
  const int t = tid & 127, ka = tid >> 7;
  const int q = tid % NB, kb = tid / NB;
  const bool ha = t < nrow, hb = q < w;
  const double* pa = P + R0 + (ha ? t : 0);
  const double* pb = I + (hb ? q : 0);
  double ra[PA], rb[PB];
  // late partial slabs of this tile (folded here instead of by a k_reduce launch: see k_potrf)
  const int32_t pn = tile_pnseg[g];
  const double* sl = slabs + (int64_t)tile_pslot[g] * (TM * NB) + (ha ? t : 0);
  auto fetch = [&](int k0) {
    const int kc = min(KCS, w - k0);
#pragma unroll
    for (int i = 0; i < PA; ++i) ra[i] = pa[(int64_t)(k0 + min(ka + 2 * i, kc - 1)) * m];
    for (int sg = 0; sg < pn; ++sg) {
      const double* q = sl + (int64_t)sg * (TM * NB);
#pragma unroll
      for (int i = 0; i < PA; ++i) ra[i] -= q[(k0 + min(ka + 2 * i, kc - 1)) * TM];
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) rb[i] = pb[(int64_t)(k0 + min(kb + (256 / NB) * i, kc - 1)) * w];
  };
  auto stage = [&](int k0) {
    const int kc = min(KCS, w - k0);
    const int kc4 = (kc + 3) & ~3;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int k = ka + 2 * i;
      if (k < kc4) As[k * LDA + t] = (ha && k < kc) ? ra[i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int k = kb + (256 / NB) * i;
      if (k < kc4) Bs[k * LDB + q] = (hb && k < kc) ? rb[i] : 0.0;  // Aop[j][k] = invL[j][k] -> Bs[k][j]
    }
  };
  fetch(0);
  for (int32_t k0 = 0; k0 < w; k0 += KCS) {
    const int kc = min(KCS, w - k0);
    const int kc4 = (kc + 3) & ~3;
    if (k0 > 0) __syncthreads();  // the previous chunk has been consumed
    stage(k0);
    if (k0 + KCS < w) fetch(k0 + KCS);
    __syncthreads();
    if (32 * wv < nrow) tile_mma<MFMA>(As, Bs, kc4, ncb, lane, wv, acc);
  }
  const int li = lane & 15, lr = lane >> 4;
#pragma unroll
  for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * jb + lr + 4 * r;
        const int i = 32 * wv + 16 * ib + li;
        if (i < nrow && R0 + i >= w && j < w) P[(int64_t)j * m + R0 + i] = acc[jb][ib][r];
      }
}


// ------------------------------------------------------------------------------------------------
// k_trsm_lite (round 4): the same panel solve, shaped to START AT ONCE beside the resident update kernel.  k_trsm<true> needs 98
// VGPRs + 256 AGPRs and 36 KB of LDS per workgroup: beside k_dense_b (210 VGPRs x 2 waves per SIMD, 135 KB of LDS per CU) none of
// its waves fits, so every level's trsm waited for dense items to retire -- 15 ms of the 52 ms factorization of the 100k config
// (profiles/r4_level_timeline_100k.csv: 0.11 ms per level for 15 us of arithmetic), the largest piece of the per-level chain.
// Here a workgroup owns 32 rows x all columns of a tile, a wave 32 rows x 32 columns (16 accumulator doubles per lane); K is
// streamed through LDS in 16-deep chunks like k_trsm (21 KB: fits beside k_dense_b's 135 KB), the next chunk's loads in flight
// during the MFMAs, <= 80 VGPRs (amdgpu_waves_per_eu(6)): one wave per SIMD fits into the registers k_dense_b leaves free.
// (A first form without LDS -- fragments straight from global memory, one 4-deep k-step at a time -- did start beside the
// update kernel but took 0.18 ms by itself: 32 dependent round trips at the loaded memory latency of a busy chip.)
// Triangular skipping: column block jb only needs k < 16 (jb + 1).  The late partial slabs are folded on load as in k_trsm.
constexpr int TL_ROWS = 32;
constexpr int TL_LDA = TL_ROWS + 16;  // [k][row] image of the 32 panel rows ((ld * 8 B) == 128 mod 256: conflict-free b64 fragment reads)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8)))
void k_trsm_lite(DevSym S, const int32_t* __restrict__ tiles, double* __restrict__ L, const double* __restrict__ invD,
                 const int32_t* __restrict__ tile_pslot, const int32_t* __restrict__ tile_pnseg, const double* __restrict__ slabs) {
  __shared__ __attribute__((aligned(16))) double As[KCS * TL_LDA];
  __shared__ __attribute__((aligned(16))) double Bs[KCS * LDB];
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int32_t g = tiles[blockIdx.x >> 2];
  const int sub = blockIdx.x & 3;                       // 32-row slab of the 128-row tile
  const int32_t s = S.tile_front[g];
  const int32_t ti = (int32_t)(g - S.tile_base[s]);
  const int32_t c0 = S.sn_start[s], w = S.sn_start[s + 1] - c0;
  const int32_t m = (int32_t)(S.sn_rowptr[s + 1] - S.sn_rowptr[s]);
  const int32_t R0 = ti * TM + TL_ROWS * sub;
  const int32_t nrow = min(TL_ROWS, m - R0);
  if (nrow <= 0 || R0 + nrow <= w) return;              // (uniform over the workgroup) nothing below the diagonal block here
  double* P = L + S.sn_loff[s];
  const double* I = invD + S.inv_off[s];
  const int32_t pn = tile_pnseg[g];
  const int j0 = 32 * wv;                                // this wave's 32 output columns
  d4 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  // staging roles: A: thread -> row t = tid & 31, k = (tid >> 5) + 8 i (i < 2); B: column q = tid & 127, k = (tid >> 7) + 2 i (i < 8)
  constexpr int PA = KCS / 8, PB = KCS / 2;
  const int t = tid & 31, ka = tid >> 5;
  const int q = tid & 127, kb = tid >> 7;
  const bool ha = t < nrow, hb = q < w;
  const double* pa = P + R0 + (ha ? t : 0);
  const double* pb = I + (hb ? q : 0);
  const double* sl = slabs + (int64_t)tile_pslot[g] * (TM * NB) + TL_ROWS * sub + (ha ? t : 0);
  double ra[PA], rb[PB];
  auto fetch = [&](int k0) {
    const int kc = min(KCS, w - k0);
#pragma unroll
    for (int i = 0; i < PA; ++i) ra[i] = pa[(int64_t)(k0 + min(ka + 8 * i, kc - 1)) * m];
    for (int sg = 0; sg < pn; ++sg) {
      const double* qs = sl + (int64_t)sg * (TM * NB);
#pragma unroll
      for (int i = 0; i < PA; ++i) ra[i] -= qs[(k0 + min(ka + 8 * i, kc - 1)) * TM];
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) rb[i] = pb[(int64_t)(k0 + min(kb + 2 * i, kc - 1)) * w];
  };
  auto stage = [&](int k0) {
    const int kc = min(KCS, w - k0);
    const int kc4 = (kc + 3) & ~3;
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int k = ka + 8 * i;
      if (k < kc4) As[k * TL_LDA + t] = (ha && k < kc) ? ra[i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int k = kb + 2 * i;
      if (k < kc4) Bs[k * LDB + q] = (hb && k < kc) ? rb[i] : 0.0;   // Aop[j][k] = invL[j][k] -> Bs[k][j]
    }
  };
  fetch(0);
  for (int32_t k0 = 0; k0 < w; k0 += KCS) {
    const int kc = min(KCS, w - k0);
    const int kc4 = (kc + 3) & ~3;
    if (k0 > 0) __syncthreads();  // the previous chunk has been consumed
    stage(k0);
    if (k0 + KCS < w) fetch(k0 + KCS);
    __syncthreads();
    if (j0 < w && k0 < j0 + 32) {  // invL[j][k] = 0 for k > j: this wave's columns end at j0 + 31
      const double* ap = Bs + lk * LDB + j0 + li;
      const double* bp = As + lk * TL_LDA + li;
      for (int k4 = 0; k4 < kc4; k4 += 4) {
        const double a0 = ap[k4 * LDB], a1 = ap[k4 * LDB + 16];
        const double b0 = bp[k4 * TL_LDA], b1 = bp[k4 * TL_LDA + 16];
        acc[0][0] = mfma_f64(a0, b0, acc[0][0]);
        acc[0][1] = mfma_f64(a0, b1, acc[0][1]);
        acc[1][0] = mfma_f64(a1, b0, acc[1][0]);
        acc[1][1] = mfma_f64(a1, b1, acc[1][1]);
      }
    }
  }
  // (every global read of the 32 rows happened before the loop's last barrier: the columns may be overwritten in place)
  if (j0 < w) {
#pragma unroll
    for (int jbb = 0; jbb < 2; ++jbb)
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int j = j0 + 16 * jbb + lk + 4 * r;
          const int i = 16 * ib + li;
          if (i < nrow && R0 + i >= w && j < w) P[(int64_t)j * m + R0 + i] = acc[jbb][ib][r];
        }
  }
}


// ------------------------------------------------------------------------------------------------
// Right-hand-side kernels.  rp = padded column count (multiple of 16, <= RPMAX); ldy = LDS leading
// dimension of [k][c] images (== 16 mod 32 so that b64 reads of Bop[k][c] are conflict-free).

__global__ void k_perm_in(int32_t n, int32_t r, int32_t rp, int32_t cbeg, const int32_t* __restrict__ perm,
                          const double* __restrict__ B, double* __restrict__ W) {
  // W[p][c] = B[perm[p]][cbeg + c]   (perm == nullptr: identity, only pads the columns)
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * rp) return;
  int32_t p = (int32_t)(idx / rp), c = (int32_t)(idx - (int64_t)p * rp);
  const int32_t src = perm ? perm[p] : p;
  W[idx] = (cbeg + c < r) ? B[(int64_t)src * r + cbeg + c] : 0.0;
}

__global__ void k_perm_out(int32_t n, int32_t r, int32_t rp, int32_t cbeg, const int32_t* __restrict__ perm,
                           const double* __restrict__ Xp, double* __restrict__ X) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * rp) return;
  int32_t p = (int32_t)(idx / rp), c = (int32_t)(idx - (int64_t)p * rp);
  if (cbeg + c < r) X[(int64_t)perm[p] * r + cbeg + c] = Xp[idx];
}

// D[M][N=c] accumulate helper for RHS kernels: one M-tile (16 rows) x ncn N-tiles (16 cols each).
// Aop[M=row][k] read from a k-major image A_lds[k*lda + row0 + (l&15)], Bop[k][c] from Y_lds[k*ldy + c].
template <bool MFMA, int NCT>
__device__ __forceinline__ void rhs_mma(const double* __restrict__ A_lds, int lda, int row0,
                                        const double* __restrict__ Y_lds, int ldy, int kn4, int ncn, int lane,
                                        d4 (&acc)[NCT]) {
  const int li = lane & 15, lk = lane >> 4;
  if (MFMA) {
    for (int k4 = 0; k4 < kn4; k4 += 4) {
      const double a = A_lds[(k4 + lk) * lda + row0 + li];
#pragma unroll
      for (int cn = 0; cn < NCT; ++cn)
        if (cn < ncn) acc[cn] = mfma_f64(a, Y_lds[(k4 + lk) * ldy + 16 * cn + li], acc[cn]);
    }
  } else {
    for (int k = 0; k < kn4; ++k)
#pragma unroll
      for (int cn = 0; cn < NCT; ++cn)
        if (cn < ncn)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[cn][r] += A_lds[k * lda + row0 + lk + 4 * r] * Y_lds[k * ldy + 16 * cn + li];
  }
}

// Right-hand-side kernels work on a window of CW = 32 columns selected by blockIdx.y (gridDim.y =
// ceil(rp / CW)): four times the workgroups per level, a quarter of the MFMA chain and of the staging per
// workgroup, 24 KB LDS images.  Global RHS rows keep the full stride rp.
#ifndef SCILMM_CW
#define SCILMM_CW 32
#endif
constexpr int CW = SCILMM_CW;  // RHS columns per workgroup
constexpr int LDW = 48;        // LDS leading dimension of [k][c] images (== 16 mod 32, >= CW)
constexpr int NCT = CW / 16;   // 16-column MFMA tiles per window

// Diagonal-block step of both sweeps, one workgroup per (front of the level, column window):
//   TRANS = false:  Xout[c0:c1] = invL_s   * Yin[c0:c1]     (forward)
//   TRANS = true :  Xout[c0:c1] = invL_s^T * Yin[c0:c1]     (backward, in place)
// invL is triangular, so row block jb only needs k < 16 (jb+1) (forward) or k >= 16 jb (backward).
template <bool MFMA, bool TRANS>
__global__ __launch_bounds__(256) void k_diag_solve(DevSym S, const int32_t* __restrict__ fronts,
                                                    const double* __restrict__ invD, const double* Yin, double* Xout,
                                                    int32_t rp) {
  __shared__ __attribute__((aligned(16))) double Ys[NB * LDW];  // y_s, [k][c]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int c_lo = blockIdx.y * CW;
  const int rpl = min(CW, rp - c_lo);
  if (rpl <= 0) return;
  const int32_t s = fronts[blockIdx.x];
  const int32_t c0 = S.sn_start[s], w = S.sn_start[s + 1] - c0;
  const int ncn = rpl >> 4;
  const int w4 = (w + 3) & ~3;
  const double* I = invD + S.inv_off[s];
  {
    const int c = tid & 63;
    if (c < LDW)
      for (int k = tid >> 6; k < w4; k += 4)
        Ys[k * LDW + c] = (k < w && c < rpl) ? Yin[(int64_t)(c0 + k) * rp + c_lo + c] : 0.0;
  }
  __syncthreads();
  const int li = lane & 15, lk = lane >> 4;
  // row blocks of 16 are dealt round-robin to the four waves; the A operand op(invL)[j][k] (w x w,
  // column-major, L2-resident) is read straight into the MFMA fragment: lane (li, lk) needs element
  // (16 jb + li, k4 + lk)
  for (int jb = wv; 16 * jb < w; jb += 4) {
    d4 xa[NCT];
#pragma unroll
    for (int cn = 0; cn < NCT; ++cn) xa[cn] = (d4){0.0, 0.0, 0.0, 0.0};
    const int kbeg = TRANS ? 16 * jb : 0;
    const int kend = TRANS ? w4 : min(w4, 16 * (jb + 1));
    const int j = 16 * jb + li;
    if (MFMA) {
      for (int k4 = kbeg; k4 < kend; k4 += 4) {
        const int k = k4 + lk;
        double a = 0.0;
        if (j < w && k < w) a = TRANS ? I[(int64_t)j * w + k] : I[(int64_t)k * w + j];
#pragma unroll
        for (int cn = 0; cn < NCT; ++cn)
          if (cn < ncn) xa[cn] = mfma_f64(a, Ys[k * LDW + 16 * cn + li], xa[cn]);
      }
    } else {
      for (int k = kbeg; k < kend && k < w; ++k)
#pragma unroll
        for (int cn = 0; cn < NCT; ++cn)
          if (cn < ncn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int jr = 16 * jb + lk + 4 * r;
              const double a = (jr < w) ? (TRANS ? I[(int64_t)jr * w + k] : I[(int64_t)k * w + jr]) : 0.0;
              xa[cn][r] += a * Ys[k * LDW + 16 * cn + li];
            }
    }
#pragma unroll
    for (int cn = 0; cn < NCT; ++cn)
      if (cn < ncn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int jr = 16 * jb + lk + 4 * r;
          if (jr < w) Xout[(int64_t)(c0 + jr) * rp + c_lo + 16 * cn + li] = xa[cn][r];
        }
  }
}

// Forward sweep of one level, push step.  Workgroup = (front s, 128-row tile, column window):
//   W[rows of the tile] -= L21[tile rows, :] * x_s        with x_s = Xin[c0:c1] (from k_diag_solve).
// ATOMIC: fronts of one level may share target rows -> fp64 hardware atomics; a level with a single front
// (every level of a dense chain) uses plain read-modify-write.
// MODE 0: forward solve.  MODE 1: multiply (Z += L[:, s] * R_s, including the diagonal block; always atomic).
template <bool MFMA, int MODE, bool ATOMIC>
__global__ __launch_bounds__(256) void k_fwd(DevSym S, const int32_t* __restrict__ tiles, const double* __restrict__ L,
                                             const double* Xin, double* W, int32_t rp) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Ys = smem;             // [NB][LDW]  x_s (or R_s), [k][c]
  double* As = smem + NB * LDW;  // [KCS][LDA]  panel chunk, [k][row]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int c_lo = blockIdx.y * CW;
  const int rpl = min(CW, rp - c_lo);
  if (rpl <= 0) return;
  const int32_t g = tiles[blockIdx.x];
  const int32_t s = S.tile_front[g];
  const int32_t ti = (int32_t)(g - S.tile_base[s]);
  const int32_t c0 = S.sn_start[s], w = S.sn_start[s + 1] - c0;
  const int32_t* rs = S.sn_rows + S.sn_rowptr[s];
  const int32_t m = (int32_t)(S.sn_rowptr[s + 1] - S.sn_rowptr[s]);
  const int32_t R0 = ti * TM;
  const int32_t nrow = min(TM, m - R0);
  if (MODE == 0 && R0 + nrow <= w) return;
  const int ncn = rpl >> 4;
  const int w4 = (w + 3) & ~3;
  const double* P = L + S.sn_loff[s];
  {
    const int c = tid & 63;
    if (c < LDW)
      for (int k = tid >> 6; k < w4; k += 4)
        Ys[k * LDW + c] = (k < w && c < rpl) ? Xin[(int64_t)(c0 + k) * rp + c_lo + c] : 0.0;
  }
  // acc[i][c] = sum_k P[R0+i][k] * x[k][c]; wave wv owns rows [32 wv, 32 wv + 32)
  d4 a0[NCT], a1[NCT];
#pragma unroll
  for (int cn = 0; cn < NCT; ++cn) {
    a0[cn] = (d4){0.0, 0.0, 0.0, 0.0};
    a1[cn] = (d4){0.0, 0.0, 0.0, 0.0};
  }
  // panel chunk c+1 is fetched into registers before the MFMAs of chunk c (same staging as k_trsm)
  constexpr int PA = KCS / 2;
  const int ta_ = tid & 127, ka_ = tid >> 7;
  const bool ha_ = ta_ < nrow;
  const double* pa_ = P + R0 + (ha_ ? ta_ : 0);
  double ra_[PA];
  auto fetch = [&](int k0) {
    const int kc = min(KCS, w - k0);
#pragma unroll
    for (int i = 0; i < PA; ++i) ra_[i] = pa_[(int64_t)(k0 + min(ka_ + 2 * i, kc - 1)) * m];
  };
  fetch(0);
  for (int32_t k0 = 0; k0 < w; k0 += KCS) {
    const int kc = min(KCS, w - k0);
    const int kc4 = (kc + 3) & ~3;
    if (k0 > 0) __syncthreads();
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int k = ka_ + 2 * i;
      if (k < kc4) As[k * LDA + ta_] = (ha_ && k < kc) ? ra_[i] : 0.0;
    }
    if (k0 + KCS < w) fetch(k0 + KCS);
    __syncthreads();
    if (32 * wv < nrow) {
      rhs_mma<MFMA, NCT>(As, LDA, 32 * wv, Ys + k0 * LDW, LDW, kc4, ncn, lane, a0);
      if (32 * wv + 16 < nrow) rhs_mma<MFMA, NCT>(As, LDA, 32 * wv + 16, Ys + k0 * LDW, LDW, kc4, ncn, lane, a1);
    }
  }
  const int li = lane & 15, lr = lane >> 4;
#pragma unroll
  for (int cn = 0; cn < NCT; ++cn)
    if (cn < ncn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int i = 32 * wv + 16 * h + lr + 4 * r;
          if (i < nrow && (MODE == 1 || R0 + i >= w)) {
            const double v = h == 0 ? a0[cn][r] : a1[cn][r];
            double* dst = &W[(int64_t)rs[R0 + i] * rp + c_lo + 16 * cn + li];
            if (ATOMIC) unsafeAtomicAdd(dst, MODE == 0 ? -v : v);
            else *dst += (MODE == 0 ? -v : v);
          }
        }
      }
}

// Backward sweep, step (b): for every update pair (target s in this level, descendant d)
//   X[cols of d] -= L_d[p0:p1, :]^T * X[rows p0..p1 of d]      (those rows are columns of s: final)
// Two fronts of one level never share a descendant (they would be on one root path), so the
// read-modify-write of X[cols of d] is exclusive: plain loads/stores, bitwise reproducible.
template <bool MFMA>
__global__ __launch_bounds__(256) void k_bwd_push(DevSym S, const int32_t* __restrict__ pairs,
                                                  const int64_t* __restrict__ grp_ptr, const double* __restrict__ L,
                                                  double* __restrict__ X, int32_t rp, const int32_t* __restrict__ grp_slot,
                                                  double* __restrict__ partial) {
  // grp_ptr == nullptr: one update pair per workgroup.  Otherwise `pairs` holds triples (descendant, p0, p1) and
  // workgroup b folds the row ranges [grp_ptr[b], grp_ptr[b+1]) -- all of the SAME descendant d -- into one
  // read-modify-write of X[cols of d] (used after the chain sweep, where the targets of many levels are final at
  // once: the rows of d that belong to consecutive chain blocks are merged into long ranges by the host).
  __shared__ __attribute__((aligned(16))) double Ps[NB * LDP];  // Ps[k*LDP + q] = L_d[p0+q0+q][k]
  __shared__ __attribute__((aligned(16))) double Xg[32 * LDW];  // gathered X rows, [q][c]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int c_lo = blockIdx.y * CW;
  const int rpl = min(CW, rp - c_lo);
  if (rpl <= 0) return;
  const int64_t g0 = grp_ptr ? grp_ptr[blockIdx.x] : (int64_t)blockIdx.x;
  const int64_t g1 = grp_ptr ? grp_ptr[blockIdx.x + 1] : (int64_t)blockIdx.x + 1;
  const int32_t d = grp_ptr ? pairs[3 * g0] : S.upd_src[pairs[g0]];
  const int32_t pslot = grp_slot ? grp_slot[blockIdx.x] : -1;
  const int32_t* rd = S.sn_rows + S.sn_rowptr[d];
  const int32_t md = (int32_t)(S.sn_rowptr[d + 1] - S.sn_rowptr[d]);
  const int32_t cd = S.sn_start[d], wd = S.sn_start[d + 1] - cd;
  const double* Pd = L + S.sn_loff[d];
  const int ncn = rpl >> 4;
  constexpr int NH = (NJB + 3) / 4;  // row blocks (of the descendant's columns) per wave
  d4 acc[NH][NCT];
#pragma unroll
  for (int h = 0; h < NH; ++h)
#pragma unroll
    for (int cn = 0; cn < NCT; ++cn) acc[h][cn] = (d4){0.0, 0.0, 0.0, 0.0};
  for (int64_t gi = g0; gi < g1; ++gi) {
  const int32_t p0 = grp_ptr ? pairs[3 * gi + 1] : S.upd_p0[pairs[gi]];
  const int32_t p1 = grp_ptr ? pairs[3 * gi + 2] : S.upd_p1[pairs[gi]];
  for (int32_t q0 = p0; q0 < p1; q0 += 32) {
    const int qn = min(32, p1 - q0);
    const int qn4 = (qn + 3) & ~3;
    if (q0 > p0 || gi > g0) __syncthreads();
    {
      const int q = tid & 31;
      for (int k = tid >> 5; k < NB; k += 8)
        Ps[k * LDP + q] = (q < qn && k < wd) ? Pd[(int64_t)k * md + q0 + q] : 0.0;
      const int c = tid & 63;
      if (c < LDW)
        for (int q2 = tid >> 6; q2 < qn4; q2 += 4)
          Xg[q2 * LDW + c] = (q2 < qn && c < rpl) ? X[(int64_t)rd[q0 + q2] * rp + c_lo + c] : 0.0;
    }
    __syncthreads();
    // D[M=k][N=c] += sum_q Ps[k][q] * Xg[q][c];  Aop[k][q] read as Ps[(16 kb + l&15)*LDP + q4 + (l>>4)]
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int kb = wv + 4 * h;
      if (16 * kb < wd) {
        if (MFMA) {
          for (int q4 = 0; q4 < qn4; q4 += 4) {
            const double a = Ps[(16 * kb + li) * LDP + q4 + lk];
#pragma unroll
            for (int cn = 0; cn < NCT; ++cn)
              if (cn < ncn) acc[h][cn] = mfma_f64(a, Xg[(q4 + lk) * LDW + 16 * cn + li], acc[h][cn]);
          }
        } else {
          for (int q = 0; q < qn4; ++q)
#pragma unroll
            for (int cn = 0; cn < NCT; ++cn)
              if (cn < ncn)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  acc[h][cn][r] += Ps[(16 * kb + lk + 4 * r) * LDP + q] * Xg[q * LDW + 16 * cn + li];
        }
      }
    }
  }
  }
  {
    const int li = lane & 15, lr = lane >> 4;
#pragma unroll
    for (int h = 0; h < NH; ++h) {
      const int kb = wv + 4 * h;
      if (16 * kb < wd) {
#pragma unroll
        for (int cn = 0; cn < NCT; ++cn)
          if (cn < ncn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int k = 16 * kb + lr + 4 * r;
              if (k < wd) {
                // a row slice of a long group leaves its partial sum for k_push_fold (fixed summation order)
                if (pslot >= 0) partial[((int64_t)pslot * NB + k) * rp + c_lo + 16 * cn + li] = acc[h][cn][r];
                else X[(int64_t)(cd + k) * rp + c_lo + 16 * cn + li] -= acc[h][cn][r];
              }
            }
      }
    }
  }
}


// X[cols of d] -= sum of the partial sums the row slices of a long group left (slices in order)
__global__ __launch_bounds__(256) void k_push_fold(DevSym S, const int32_t* __restrict__ fold, const double* __restrict__ partial,
                                                   double* __restrict__ X, int32_t rp) {
  const int32_t d = fold[3 * blockIdx.x], slot0 = fold[3 * blockIdx.x + 1], ns = fold[3 * blockIdx.x + 2];
  const int c_lo = blockIdx.y * CW;
  const int rpl = min(CW, rp - c_lo);
  const int32_t cd = S.sn_start[d], wd = S.sn_start[d + 1] - cd;
  const int cc = threadIdx.x % CW;
  if (cc >= rpl) return;
  for (int k = threadIdx.x / CW; k < wd; k += 256 / CW) {
    double sum = 0.0;
    for (int sl = 0; sl < ns; ++sl) sum += partial[((int64_t)(slot0 + sl) * NB + k) * rp + c_lo + cc];
    X[(int64_t)(cd + k) * rp + c_lo + cc] -= sum;
  }
}

// ------------------------------------------------------------------------------------------------
// Dense-chain sweeps.  The last levels of the elimination tree are a chain of single fronts (the blocks of
// the trailing dense clique): level-by-level kernels spend ~170 us per block there, almost all of it latency.
// These two kernels run the whole chain in ONE launch each: workgroup (chain block i, RHS window c) PULLS the
// contributions of the other chain blocks as soon as they are final,
//     forward :  x_i = invL_i   (w_i - sum_{j<i} L_ij   x_j)      L_ij  = rows [p0,p0+nq) of panel j
//     backward:  x_i = invL_i^T (y_i - sum_{t>i} L_ti^T x_t)      L_ti  = rows [p0,p0+nq) of panel i
// and publishes x_i through a flag (flag value = epoch of this launch, so flags are never reset).  The XCDs'
// L2s are not coherent with each other, and device-scope fences (L2 write-back / invalidate) cost tens of
// microseconds per step; instead only the communicated data is accessed coherently: x windows and flags are
// written with agent-scope (write-through) stores and read with agent-scope (cache-bypassing) loads, the
// producer drains its stores (s_waitcnt vmcnt(0)) before the barrier that precedes the flag store, and the
// panels of L keep using ordinary cached loads.
// A workgroup takes its chain position from an atomic TICKET drawn when it starts (not from blockIdx: HIP does not
// promise a dispatch order across XCDs / priorities / concurrent kernels).  It only ever waits for positions with a
// lower ticket, whose workgroups have therefore already started and cannot be starved by it: no co-residency
// requirement (decoupled look-back argument); the wait is bounded as well (err flag), so every wave always exits.
// The L fragments are read straight from global memory into the MFMA A operand BEFORE the wait: only the
// 128 x 32 x-window of the block just finished is on the critical path.
struct ChainPair {
  int32_t other;  // chain position of the other block (descendant j forward, target t backward)
  int32_t p0, nq; // rows [p0, p0+nq) of the descendant panel ...
  int32_t jp0;    // ... are columns jp0.. of the target block when >= 0 (contiguous: every pair of a dense chain)
  int32_t map;    // jp0 < 0, forward: offset into the column -> row map (NB entries, -1 = no such row)
};


__device__ __forceinline__ bool chain_wait(const int32_t* flag, int32_t epoch, int32_t* err, int behind) {
  // one thread spins; returns false on timeout / earlier error (the caller then leaves quietly).  `behind` = how
  // many more blocks this workgroup has to wait for after this one: only the workgroups next in line poll
  // tightly, the others mostly sleep (a few hundred pollers on two dozen flag lines slow every hop down).
  // Timeout = no workgroup of the sweep has published anything for ~1 s of wall clock (err[1] counts published
  // windows; a healthy sweep publishes one every few microseconds), so a long sweep is never cut short and a
  // stuck one releases its CUs after a second.
  int spins = 0;
  int32_t seen = __hip_atomic_load(err + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  unsigned long long t_seen = wall_clock64();  // 100 MHz constant-rate counter
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) {
    if (behind == 0) __builtin_amdgcn_s_sleep(1);
    else
      for (int z = 0; z < min(behind, 8); ++z) __builtin_amdgcn_s_sleep(127);
    if ((++spins & 255) == 0) {
      if (__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
      const int32_t now = __hip_atomic_load(err + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long t = wall_clock64();
      if (now != seen) {
        seen = now;
        t_seen = t;
      } else if (t - t_seen > 100000000ull) {
        __hip_atomic_store(err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return false;
      }
    }
  }
  return true;
}

#ifndef SCILMM_CHAIN_WAVES
#define SCILMM_CHAIN_WAVES 2
#endif
#ifdef SCILMM_CHAIN_PROF
__device__ unsigned long long g_chain_prof[8 * 4096];  // per front (window 0, forward sweep): eight timestamps
#define CPROF(slot) do { if (c == 0 && tid == 0 && i < 4096 && !BWD) g_chain_prof[8 * i + (slot)] = wall_clock64(); } while (0)
#else
#define CPROF(slot) do {} while (0)
#endif
// NCTW = 16-column tiles per RHS window: 2 (32 columns, two workgroups per CU: the latency-bound short chains), 4 (64
// columns, one workgroup per CU) or 7 (112 columns: every fused right-hand side of an evaluation in ONE window) for long
// chains, where every window re-reads the whole dense tail (1M config: 4 -> 1 passes over 127 GB per sweep) and the
// fixed cost of a pair (flag, x window, two barriers) is spread over 2 - 3.5 times the MFMAs.
template <bool MFMA, bool BWD, int NCTW>
__global__ __launch_bounds__(512, NCTW == 2 ? SCILMM_CHAIN_WAVES : 1) void k_chain(DevSym S, int32_t T, const int32_t* __restrict__ chain,
                                               const int32_t* __restrict__ pair_ptr, const ChainPair* __restrict__ pairs,
                                               const int32_t* __restrict__ colmap, const double* __restrict__ L,
                                               const double* __restrict__ invD, const double* W, double* X, int32_t rp,
                                               int32_t ncw, int32_t* flags, int32_t epoch, int32_t* err, int32_t* ticket) {
  constexpr int CW = 16 * NCTW, LDW = NCTW == 2 ? 48 : NCTW == 4 ? 80 : 112, NCT = NCTW;  // (LDW == 16 mod 32, >= CW: shadows the file-wide window constants)
  static_assert(NCTW == 2 || NCTW == 4 || NCTW == 7, "k_chain: 32-, 64- or 112-column windows");
  __shared__ __attribute__((aligned(16))) double Ys[NB * LDW];  // x window of the other block, then w_i: [k][c]
  __shared__ int s_ok, s_ready, s_ticket;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  if (tid == 0) s_ticket = atomicAdd(ticket, 1);  // chain position = order of ARRIVAL on the device
  __syncthreads();
  const int32_t lin = s_ticket;
  const int32_t ord = lin / ncw, c = lin - ord * ncw;
  const int32_t i = BWD ? T - 1 - ord : ord;
  const int c_lo = c * CW;
  const int rpl = min(CW, rp - c_lo);
  const int ncn = rpl >> 4;
  const int32_t s = chain[i];
  const int32_t c0 = S.sn_start[s], w = S.sn_start[s + 1] - c0;
  const int32_t m = (int32_t)(S.sn_rowptr[s + 1] - S.sn_rowptr[s]);
  const double* I = invD + S.inv_off[s];
  constexpr int NK = NB / 4;
  // wave wv owns row block wv of the result (rows / columns 16 wv .. 16 wv + 15 of block i)
  const int jrow = 16 * wv + li;  // A-operand row of this lane
  const int32_t e0 = pair_ptr[i], e1 = pair_ptr[i + 1];
  // ---- how many leading pairs are already final?  one parallel look at their flags instead of one round trip each
  if (wv == 0) {
    int ready = 0;
    bool open = true;
    for (int32_t base = e0; base < e1 && open; base += 64) {
      bool r = false;
      if (base + lane < e1)
        r = __hip_atomic_load(flags + (int64_t)pairs[base + lane].other * ncw + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= epoch;
      const unsigned long long b = __ballot(r);
      const int lead = (~b == 0ull) ? 64 : __builtin_ctzll(~b);
      ready += lead;
      open = lead == 64;
    }
    if (lane == 0) s_ready = ready;
  }
  // ---- everything of the diagonal step that does not depend on other blocks: op(invL_i) fragment and own rows
  const int w4 = (w + 3) & ~3;
  const int kbeg = BWD ? 16 * wv : 0;
  const int kend = BWD ? w4 : min(w4, 16 * (wv + 1));
  double iv[NK];
  // Fragment loads use a wave-uniform base (SGPR pair) per k-step plus ONE 32-bit lane offset, whole k-steps
  // are loaded or skipped by a uniform test, and lanes past the last column / row are discarded afterwards (the
  // arrays carry slack for that): no per-load address registers, no per-lane branches.
  auto load_iv = [&]() {
    const int jc = jrow < w ? jrow : 0;
    const uint32_t voff = (uint32_t)(BWD ? jc * w + lk : lk * w + jc);
    const int klast = uniform_int(max(w4 - 4, 0)), wu = uniform_int(w);
    const double* Ib = invD + uniform_i64(S.inv_off[s]);
#pragma unroll
    for (int u = 0; u < NK; ++u) {  // unconditional, independent loads (a k-step past the block re-reads the last one)
      const int ku = min(4 * u, klast);
      iv[u] = ld_off(Ib + (BWD ? ku : ku * wu), voff * 8u);
    }
  };  // (lanes / k-steps outside the triangle are masked where iv is used)
  const double* Yin = BWD ? (const double*)X : W;
  double yv[NCT][4];
  auto load_yv = [&]() {
#pragma unroll
    for (int cn = 0; cn < NCT; ++cn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jr = 16 * wv + lk + 4 * r;
        yv[cn][r] = (jr < w && cn < ncn) ? Yin[(int64_t)(c0 + jr) * rp + c_lo + 16 * cn + li] : 0.0;
      }
  };
  if (NCTW == 2) load_yv();  // (wide windows: after the pairs, like the inverse fragment)
  d4 acc[NCT];
#pragma unroll
  for (int cn = 0; cn < NCT; ++cn) acc[cn] = (d4){0.0, 0.0, 0.0, 0.0};
  __syncthreads();
  const int32_t nready = s_ready;
  bool ok = true;
  // L fragment of pair e -> registers (independent of every flag): issued one pair ahead of its use
  auto load_frag = [&](int32_t e, double (&av)[NK]) {
    const ChainPair pr = pairs[e];
    if (!BWD) {
      // A[jrow][k] = L_ij[jrow][k] = P_j[k * md + p0 + q(jrow)],  k < w_j
      const int32_t so = chain[pr.other];
      const int32_t wo = S.sn_start[so + 1] - S.sn_start[so];
      const int64_t md = S.sn_rowptr[so + 1] - S.sn_rowptr[so];
      int q = jrow - pr.jp0;
      if (pr.jp0 < 0) q = colmap[pr.map + jrow];
      const bool rowok = q >= 0 && q < pr.nq;
      const double* Pj = L + uniform_i64(S.sn_loff[so] + pr.p0);
      const int64_t mdu = uniform_int((int)md);
      const uint32_t voff = (uint32_t)(lk * (int)md + (rowok ? q : 0));
      const int klast = uniform_int(max(((wo + 3) & ~3) - 4, 0));
#pragma unroll
      for (int u = 0; u < NK; ++u) av[u] = ld_off(Pj + (int64_t)min(4 * u, klast) * mdu, voff * 8u);
    } else {
      // A[jrow][q] = L_ti[q][jrow] = P_i[jrow * m + p0 + q],  q < nq
      const double* Pi = L + uniform_i64(S.sn_loff[s] + pr.p0);
      const uint32_t voff = (uint32_t)((jrow < w ? jrow : 0) * m + lk);
      const int klast = uniform_int(max(((pr.nq + 3) & ~3) - 4, 0));
#pragma unroll
      for (int u = 0; u < NK; ++u) av[u] = ld_off(Pi + min(4 * u, klast), voff * 8u);
    }
  };
  auto consume = [&](int32_t e, double (&av)[NK]) {
    const ChainPair pr = pairs[e];
    const int32_t so = chain[pr.other];
    const int32_t co = S.sn_start[so], wo = S.sn_start[so + 1] - co;
    const int kn = BWD ? ((pr.nq + 3) & ~3) : ((wo + 3) & ~3);  // depth of the product (rounded up to 4)
    // lanes whose row / k lies outside the pair were loaded from a clamped address: mask them at the point of use
    bool rowok;
    if (!BWD) {
      int q = jrow - pr.jp0;
      if (pr.jp0 < 0) q = colmap[pr.map + jrow];
      rowok = q >= 0 && q < pr.nq;
    } else {
      rowok = jrow < w;
    }
    const int kvalid = BWD ? pr.nq : wo;
    if (NCTW == 2 && e == e1 - 1) load_iv();  // (wide windows: the registers are full, the fragment is fetched after the pairs)
    if (e - e0 >= nready) {
      if (tid == 0) s_ok = chain_wait(flags + (int64_t)pr.other * ncw + c, epoch, err, e1 - 1 - e) ? 1 : 0;
      if (e == e1 - 1) CPROF(0);  // flag of the newest block observed
      __syncthreads();
      ok = s_ok != 0;
    }
    if (ok) {
      // x rows: forward = all columns of block j; backward = the rows of this pair (columns of block t).
      // Plain (cached) loads are safe: nothing read x of that block on this CU / XCD before its flag was seen,
      // and a 128-byte line never holds data of two producers (windows are 256-byte aligned).
      const int nx = BWD ? pr.nq : wo;
#pragma unroll
      for (int cbase = 0; cbase < LDW; cbase += 64) {  // (windows wider than 64 columns: two passes)
        const int cc = cbase + (tid & 63);
        if (cc < LDW) {
          // all rows of this thread are requested before the first one is written to LDS (the loop form waited
          // for every load in turn: 16 exposed latencies on the critical path of the sweep)
          double xv[NB / 8];
#pragma unroll
          for (int u = 0; u < NB / 8; ++u) {
            const int k = (tid >> 6) + 8 * u;
            xv[u] = 0.0;
            if (k < nx && cc < rpl) {
              int64_t xr;
              if (!BWD) xr = co + k;
              else xr = (pr.jp0 >= 0) ? co + pr.jp0 + k : S.sn_rows[S.sn_rowptr[s] + pr.p0 + k];
              xv[u] = X[xr * rp + c_lo + cc];
            }
          }
#pragma unroll
          for (int u = 0; u < NB / 8; ++u) {
            const int k = (tid >> 6) + 8 * u;
            if (k < kn) Ys[k * LDW + cc] = xv[u];
          }
        }
      }
    }
    __syncthreads();
    if (e == e1 - 1) CPROF(1);  // x window staged
    if (ok) {
      if (MFMA) {
#pragma unroll
        for (int u = 0; u < NK; ++u)
          if (4 * u < kn) {
#pragma unroll
            for (int cn = 0; cn < NCT; ++cn)
              if (cn < ncn)
                acc[cn] = mfma_f64((rowok && 4 * u + lk < kvalid) ? av[u] : 0.0, Ys[(4 * u + lk) * LDW + 16 * cn + li], acc[cn]);
          }
      } else {
        // scalar restatement in the same accumulator layout D[(l>>4)+4r][l&15]: A is re-read by (row, k)
        for (int k = 0; k < kn; ++k) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int jr = 16 * wv + lk + 4 * r;
            double a = 0.0;
            if (!BWD) {
              const int64_t md = S.sn_rowptr[so + 1] - S.sn_rowptr[so];
              int q = jr - pr.jp0;
              if (pr.jp0 < 0) q = colmap[pr.map + jr];
              if (q >= 0 && q < pr.nq && k < wo) a = L[S.sn_loff[so] + (int64_t)k * md + pr.p0 + q];
            } else {
              if (jr < w && k < pr.nq) a = L[S.sn_loff[s] + (int64_t)jr * m + pr.p0 + k];
            }
#pragma unroll
            for (int cn = 0; cn < NCT; ++cn)
              if (cn < ncn) acc[cn][r] += a * Ys[k * LDW + 16 * cn + li];
          }
        }
      }
    }
    __syncthreads();  // Ys is reused by the next pair
    if (e == e1 - 1) CPROF(2);  // last pair multiplied
  };
  if (NCTW == 2 && e0 == e1) load_iv();
  for (int32_t e = e0; e < e1 && ok; ++e) {
    double av[NK];
    load_frag(e, av);
    consume(e, av);
  }
  if (NCTW != 2) {
    load_iv();
    load_yv();
  }
  // ---- diagonal step: v = (W or X)[block i] - acc  ->  Ys,   x_i = op(invL_i) v
  if (ok) {
#pragma unroll
    for (int cn = 0; cn < NCT; ++cn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jr = 16 * wv + lk + 4 * r;
        if (jr < w4) Ys[jr * LDW + 16 * cn + li] = (jr < w && cn < ncn) ? yv[cn][r] - acc[cn][r] : 0.0;
      }
  }
  __syncthreads();
  CPROF(3);  // w_i in LDS
  if (ok && 16 * wv < w) {
    d4 xa[NCT];
#pragma unroll
    for (int cn = 0; cn < NCT; ++cn) xa[cn] = (d4){0.0, 0.0, 0.0, 0.0};
    if (MFMA) {
#pragma unroll
      for (int u = 0; u < NK; ++u)
        if (4 * u >= kbeg && 4 * u < kend) {
#pragma unroll
          for (int cn = 0; cn < NCT; ++cn)
            if (cn < ncn)
              xa[cn] = mfma_f64((jrow < w && 4 * u + lk < w) ? iv[u] : 0.0, Ys[(4 * u + lk) * LDW + 16 * cn + li], xa[cn]);
        }
    } else {
      for (int k = kbeg; k < kend && k < w; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int jr = 16 * wv + lk + 4 * r;
          const double a = (jr < w) ? (BWD ? I[(int64_t)jr * w + k] : I[(int64_t)k * w + jr]) : 0.0;
#pragma unroll
          for (int cn = 0; cn < NCT; ++cn)
            if (cn < ncn) xa[cn][r] += a * Ys[k * LDW + 16 * cn + li];
        }
    }
#pragma unroll
    for (int cn = 0; cn < NCT; ++cn)
      if (cn < ncn)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int jr = 16 * wv + lk + 4 * r;
          if (jr < w)
            __hip_atomic_store(&X[(int64_t)(c0 + jr) * rp + c_lo + 16 * cn + li], xa[cn][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
  }
  CPROF(4);  // wave 0: diagonal product done, stores issued
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's write-through stores have completed
  __syncthreads();
  CPROF(5);  // all waves drained
  if (tid == 0 && ok) {
    __hip_atomic_store(flags + (int64_t)i * ncw + c, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(err + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // progress beacon for the waiters' timeout
  }
}

// ------------------------------------------------------------------------------------------------
// out[c] = sum over pattern entries a_ij U[i][c] U[j][c] (x2 off the diagonal): the SpMM+reduce of the
// stochastic gradient (reference compute_gradients, SparseCholesky.py:65).  Each wave walks a contiguous
// range of pattern slots; lanes own RHS columns c and c+64.  HBM/L2-bound gather of U rows.
__global__ __launch_bounds__(256) void k_quad(DevSym S, int64_t nnz, int64_t slots_per_wave,
                                              const double* __restrict__ vals, const double* __restrict__ U,
                                              int32_t rp, double* __restrict__ partial) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t e0 = wave * slots_per_wave;
  const int64_t e1 = min(nnz, e0 + slots_per_wave);
  double t0 = 0.0, t1 = 0.0;
  if (e0 < e1) {
    // column containing slot e0
    int32_t lo = 0, hi = S.n;
    while (lo < hi) {
      int32_t mid = (lo + hi) >> 1;
      if (S.pat_colptr[mid + 1] <= e0) lo = mid + 1; else hi = mid;
    }
    int32_t j = lo;
    int64_t cend = S.pat_colptr[j + 1];
    const bool h0 = lane < rp, h1 = lane + 64 < rp;
    double uj0 = h0 ? U[(int64_t)j * rp + lane] : 0.0, uj1 = h1 ? U[(int64_t)j * rp + lane + 64] : 0.0;
    double a0 = 0.0, a1 = 0.0;
    for (int64_t e = e0; e < e1; ++e) {
      if (e >= cend) {
        t0 += uj0 * a0;
        t1 += uj1 * a1;
        a0 = a1 = 0.0;
        while (e >= cend) { ++j; cend = S.pat_colptr[j + 1]; }
        uj0 = h0 ? U[(int64_t)j * rp + lane] : 0.0;
        uj1 = h1 ? U[(int64_t)j * rp + lane + 64] : 0.0;
      }
      const int32_t i = S.pat_row[e];
      const double a = (i == j) ? vals[e] : 2.0 * vals[e];
      if (h0) a0 += a * U[(int64_t)i * rp + lane];
      if (h1) a1 += a * U[(int64_t)i * rp + lane + 64];
    }
    t0 += uj0 * a0;
    t1 += uj1 * a1;
  }
  partial[wave * RPMAX + lane] = t0;
  partial[wave * RPMAX + lane + 64] = t1;
}

__global__ void k_quad_reduce(int64_t nwaves, const double* __restrict__ partial, int32_t rp, double* __restrict__ out,
                              double scale, int accumulate) {
  const int c = threadIdx.x;
  if (c >= rp) return;
  double s = 0.0;
  for (int64_t w = 0; w < nwaves; ++w) s += partial[w * RPMAX + c];
  out[c] = (accumulate ? out[c] : 0.0) + scale * s;
}

// diagonal matrix: out[c] = sum_j d_j U[j][c]^2 ; one workgroup, fixed summation order
__global__ __launch_bounds__(256) void k_quad_diag(int32_t n, const double* __restrict__ dvals,
                                                   const double* __restrict__ U, int32_t rp, double* __restrict__ partial) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t nw = (int64_t)gridDim.x * 4;
  double t0 = 0.0, t1 = 0.0;
  for (int64_t j = wave; j < n; j += nw) {
    const double dj = dvals[j];
    if (lane < rp) { double u = U[j * rp + lane]; t0 += dj * u * u; }
    if (lane + 64 < rp) { double u = U[j * rp + lane + 64]; t1 += dj * u * u; }
  }
  partial[wave * RPMAX + lane] = t0;
  partial[wave * RPMAX + lane + 64] = t1;
}

// Y = A X for few columns (symmetric half stored): used by the REML trace term and the Hessian.
__global__ void k_spmm(DevSym S, int64_t nnz, const double* __restrict__ vals, const double* __restrict__ X, int32_t rp,
                       int32_t r, double* __restrict__ Y) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nnz) return;
  int32_t lo = 0, hi = S.n;
  while (lo < hi) {
    int32_t mid = (lo + hi) >> 1;
    if (S.pat_colptr[mid + 1] <= e) lo = mid + 1; else hi = mid;
  }
  const int32_t j = lo, i = S.pat_row[e];
  const double a = vals[e];
  if (a == 0.0) return;
  for (int c = 0; c < r; ++c) {
    unsafeAtomicAdd(&Y[(int64_t)i * rp + c], a * X[(int64_t)j * rp + c]);
    if (i != j) unsafeAtomicAdd(&Y[(int64_t)j * rp + c], a * X[(int64_t)i * rp + c]);
  }
}

// Y += A X with the lanes of a wave owning the right-hand-side columns (c, c + 64) -- k_quad's walk over a contiguous range of
// pattern slots: per stored entry a_ij (i >= j) the row X[i] is read ONCE (both halves of the wave-wide row are coalesced), the
// column's own sum Y[j] += a_ij X[i] is kept in registers and flushed when the column ends, and Y[i] += a_ij X[j] is one
// coalesced wave-wide atomic add per half -- instead of k_spmm's thread per entry looping over the columns with scalar atomics
// (300k config, 103 columns: 1.82 s -> see profiles).  Y must be zero (or hold the sum to add to) on entry.
__global__ __launch_bounds__(256) void k_spmm_w(DevSym S, int64_t nnz, int64_t slots_per_wave, const double* __restrict__ vals,
                                                const double* __restrict__ X, int32_t rp, double* __restrict__ Y) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t e0 = wave * slots_per_wave;
  const int64_t e1 = min(nnz, e0 + slots_per_wave);
  if (e0 >= e1) return;
  int32_t lo = 0, hi = S.n;
  while (lo < hi) {
    int32_t mid = (lo + hi) >> 1;
    if (S.pat_colptr[mid + 1] <= e0) lo = mid + 1; else hi = mid;
  }
  int32_t j = lo;
  int64_t cend = S.pat_colptr[j + 1];
  const bool h0 = lane < rp, h1 = lane + 64 < rp;
  double xj0 = h0 ? X[(int64_t)j * rp + lane] : 0.0, xj1 = h1 ? X[(int64_t)j * rp + lane + 64] : 0.0;
  double a0 = 0.0, a1 = 0.0;
  auto flush = [&]() {
    if (h0 && a0 != 0.0) unsafeAtomicAdd(&Y[(int64_t)j * rp + lane], a0);
    if (h1 && a1 != 0.0) unsafeAtomicAdd(&Y[(int64_t)j * rp + lane + 64], a1);
    a0 = a1 = 0.0;
  };
  for (int64_t e = e0; e < e1; ++e) {
    if (e >= cend) {
      flush();
      while (e >= cend) { ++j; cend = S.pat_colptr[j + 1]; }
      xj0 = h0 ? X[(int64_t)j * rp + lane] : 0.0;
      xj1 = h1 ? X[(int64_t)j * rp + lane + 64] : 0.0;
    }
    const int32_t i = S.pat_row[e];
    const double a = vals[e];
    if (a == 0.0) continue;
    if (h0) a0 += a * X[(int64_t)i * rp + lane];
    if (h1) a1 += a * X[(int64_t)i * rp + lane + 64];
    if (i != j) {
      if (h0) unsafeAtomicAdd(&Y[(int64_t)i * rp + lane], a * xj0);
      if (h1) unsafeAtomicAdd(&Y[(int64_t)i * rp + lane + 64], a * xj1);
    }
  }
  flush();
}

__global__ void k_spmm_diag(int32_t n, const double* __restrict__ dvals, const double* __restrict__ X, int32_t rp,
                            double* __restrict__ Y) {
  int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n * rp) return;
  Y[idx] += dvals[idx / rp] * X[idx];
}

// ------------------------------------------------------------------------------------------------
// Haseman-Elston moments (reference HE, scilmm/SparseCholesky.py:192-246: S_ij = sum(A_i o A_j) - diag_i . diag_j,
// q_i = y'A_i y - diag_i . y^2): the only matrix-sized work is the Frobenius inner product of two value arrays that
// already sit in HBM in pattern-slot order.  One streaming pass (HBM-bound, 16 B per slot), block partials in a
// fixed order; the host adds them up.  part[b] = sum over the block's slots of v1 * v2.
__global__ __launch_bounds__(256) void k_dot_slots(int64_t nnz, const double* __restrict__ v1, const double* __restrict__ v2,
                                                   double* __restrict__ part) {
  __shared__ double red[4];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double s0 = 0.0, s1 = 0.0;
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; e + stride < nnz; e += 2 * stride) {
    s0 += v1[e] * v2[e];
    s1 += v1[e + stride] * v2[e + stride];
  }
  if (e < nnz) s0 += v1[e] * v2[e];
  double s = s0 + s1;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// diagonal entries: slot pat_colptr[j] is the diagonal of permuted column j.  part[b] = sum_j d1_j * d2_j
__global__ __launch_bounds__(256) void k_dot_diag(int32_t n, const int64_t* __restrict__ pat_colptr, const double* __restrict__ v1,
                                                  const double* __restrict__ v2, int diag1, int diag2, double* __restrict__ part) {
  __shared__ double red[4];
  double s = 0.0;
  for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (int64_t)gridDim.x * blockDim.x) {
    const double a = diag1 ? v1[j] : v1[pat_colptr[j]];  // diagonal-only matrices are stored as one value per row
    const double b = diag2 ? v2[j] : v2[pat_colptr[j]];
    s += a * b;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------
// Selected inverse on the supernodal factor (Takahashi / Erisman-Tinney recursion): every stored entry
// of L is replaced, IN PLACE, by the entry of Z = (V[P][:,P])^-1 at the same position.  For front s with columns C
// (w of them), rows below R (u = m - w) and stored panel [L11; L21], processed from the LAST level down:
//     Y    = L21 L11^-1                       (k_sinv_y: the trsm product with the inverse NOT transposed; kept in Ybuf)
//     Z_RC = - Z_RR Y                         (k_sinv_w: gather-GEMM; R x R entries of Z live in the ancestors' panels)
//     Z_CC = L11^-T L11^-1 - Y^T Z_RC         (k_sinv_cc0 writes the first term, k_sinv_cc adds the second per row tile)
// R is a clique of the filled graph, so every Z(r1, r2), r1, r2 in R, is a stored entry of an ancestor: the entry
// (hi, lo) sits in the panel of the front that owns column lo, at the row of hi -- found by arithmetic in the dense tail
// (rows = all later columns) and by bisection in a prelude front's row list.  Twice the factorization's flops; the
// factor is consumed (it has to be refactorized before the next solve).  tr(V^-1 A_k) is then one pass over A_k's
// pattern slots: sum_e c_e vals_k[e] Z[asm_dst[e]] (k_sinv_trace), c_e = 2 off the diagonal.
struct SinvOwner {  // where the entries Z(., lo) with lo in one front live
  int64_t loff;       // panel offset
  const int32_t* rows;  // row list of the front (prelude fronts: searched)
  int32_t c0, m;      // first column, panel rows
  int32_t tail;       // rows = c0 .. n-1: position by arithmetic
};
__device__ __forceinline__ SinvOwner sinv_owner(const DevSym& S, const int32_t* __restrict__ col_front, int32_t dense_first, int32_t col) {
  const int32_t a = col_front[col];
  SinvOwner o;
  o.loff = S.sn_loff[a];
  o.rows = S.sn_rows + S.sn_rowptr[a];
  o.c0 = S.sn_start[a];
  o.m = (int32_t)(S.sn_rowptr[a + 1] - S.sn_rowptr[a]);
  o.tail = a >= dense_first;
  return o;
}
// offset of Z(hi, lo) (hi >= lo) in the panel of lo's owner
__device__ __forceinline__ int64_t sinv_offset(const SinvOwner& o, int32_t hi, int32_t lo) {
  int32_t pos;
  if (o.tail) {
    pos = hi - o.c0;
  } else {
    int32_t a = 0, b = o.m;
    while (a < b) {  // first position whose row label is >= hi (hi is in the list: R is a clique)
      const int32_t mid = (a + b) >> 1;
      if (o.rows[mid] < hi) a = mid + 1; else b = mid;
    }
    pos = a;
  }
  return o.loff + (int64_t)(lo - o.c0) * o.m + pos;
}

// Y tile = L21 tile * L11^-1 (128 x w x w), written column-major (leading dimension u) into Ybuf[yoff[front]]
template <bool MFMA>
__global__ __launch_bounds__(256) void k_sinv_y(DevSym S, const int32_t* __restrict__ tiles, const double* __restrict__ L,
                                                const double* __restrict__ invD, double* __restrict__ Ybuf,
                                                const int64_t* __restrict__ yoff, int32_t dense_first) {
  __shared__ __attribute__((aligned(16))) double As[KCS * LDA];
  __shared__ __attribute__((aligned(16))) double Bs[KCS * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int32_t g = tiles[blockIdx.x];
  const int32_t s = S.tile_front[g];
  const int32_t ti = (int32_t)(g - S.tile_base[s]);
  const int32_t c0 = S.sn_start[s], w = S.sn_start[s + 1] - c0;
  const int32_t m = (int32_t)(S.sn_rowptr[s + 1] - S.sn_rowptr[s]);
  const int32_t R0 = ti * TM;
  const int32_t nrow = min(TM, m - R0);
  if (R0 + nrow <= w) return;
  const int32_t u = m - w;
  const int ncb = (w + 15) >> 4;
  const double* P = L + S.sn_loff[s];
  const double* I = invD + S.inv_off[s];
  double* Y = Ybuf + yoff[s];
  d4 acc[NJB][2];
#pragma unroll
  for (int a = 0; a < NJB; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  const int t = tid & 127, ka = tid >> 7;
  const int q = tid % NB, kb = tid / NB;
  const bool ha = t < nrow, hb = q < w;
  for (int32_t k0 = 0; k0 < w; k0 += KCS) {
    const int kc = min(KCS, w - k0);
    const int kc4 = (kc + 3) & ~3;
    if (k0 > 0) __syncthreads();
#pragma unroll
    for (int i = 0; i < KCS / 2; ++i) {
      const int k = ka + 2 * i;
      if (k < kc4) As[k * LDA + t] = (ha && k < kc) ? P[(int64_t)(k0 + k) * m + R0 + t] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < KCS / (256 / NB); ++i) {
      const int k = kb + (256 / NB) * i;
      // Bop[k][j] = invL[k][j] (NOT transposed; k_trsm stages invL[j][k]); invL is stored column-major w x w
      if (k < kc4) Bs[k * LDB + q] = (hb && k < kc) ? I[(int64_t)q * w + k0 + k] : 0.0;
    }
    __syncthreads();
    if (32 * wv < nrow) tile_mma<MFMA>(As, Bs, kc4, ncb, lane, wv, acc);
  }
  const int li = lane & 15, lr = lane >> 4;
#pragma unroll
  for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * jb + lr + 4 * r;
        const int i = 32 * wv + 16 * ib + li;
        if (i < nrow && R0 + i >= w && j < w) {
          // dense-tail fronts keep Y TRANSPOSED ([u][128]: the k-rows k_sinv_tail's LDS-DMA copies), the others column-major
          if (s >= dense_first) Y[(int64_t)(R0 + i - w) * NB + j] = acc[jb][ib][r];
          else Y[(int64_t)j * u + (R0 + i - w)] = acc[jb][ib][r];
        }
      }
}

// Z_RC tile = - sum_k Z(r_i, r_k) Y[k, :]  over all rows k of R, the Z entries gathered from the ancestors' panels
template <bool MFMA>
__global__ __launch_bounds__(256) void k_sinv_w(DevSym S, const int32_t* __restrict__ tiles, double* L,
                                                const double* __restrict__ Ybuf, const int64_t* __restrict__ yoff,
                                                const int32_t* __restrict__ col_front, int32_t dense_first) {
  __shared__ __attribute__((aligned(16))) double As[KCS * LDA];  // [k][i] = Z(r_i, r_k)
  __shared__ __attribute__((aligned(16))) double Bs[KCS * LDB];  // [k][j] = Y[k][j]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int32_t g = tiles[blockIdx.x];
  const int32_t s = S.tile_front[g];
  const int32_t ti = (int32_t)(g - S.tile_base[s]);
  const int32_t c0 = S.sn_start[s], w = S.sn_start[s + 1] - c0;
  const int32_t* rs = S.sn_rows + S.sn_rowptr[s];
  const int32_t m = (int32_t)(S.sn_rowptr[s + 1] - S.sn_rowptr[s]);
  const int32_t R0 = ti * TM;
  const int32_t nrow = min(TM, m - R0);
  if (R0 + nrow <= w) return;
  const int32_t u = m - w;
  const int ncb = (w + 15) >> 4;
  double* P = L + S.sn_loff[s];
  const double* Y = Ybuf + yoff[s];
  d4 acc[NJB][2];
#pragma unroll
  for (int a = 0; a < NJB; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  constexpr int PA = KCS / 2, PB = KCS / (256 / NB);
  const int t = tid & 127, ka = tid >> 7;
  const int q = tid % NB, kb = tid / NB;
  const bool ha = t < nrow && R0 + t >= w, hb = q < w;
  const int32_t gi = ha ? rs[R0 + t] : 0;
  const SinvOwner oi = sinv_owner(S, col_front, dense_first, gi);  // used where r_i < r_k: the entry lives with r_i's owner
  double ra[PA], rb[PB];
  auto fetch = [&](int k0) {
    const int kc = min(KCS, u - k0);
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int k = ka + 2 * i;
      double v = 0.0;
      if (ha && k < kc) {
        const int32_t gk = rs[w + k0 + k];
        if (gi >= gk) {
          const SinvOwner ok = sinv_owner(S, col_front, dense_first, gk);
          v = L[sinv_offset(ok, gi, gk)];
        } else {
          v = L[sinv_offset(oi, gk, gi)];
        }
      }
      ra[i] = v;
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int k = kb + (256 / NB) * i;
      rb[i] = (hb && k < kc) ? Y[(int64_t)q * u + k0 + k] : 0.0;
    }
  };
  fetch(0);
  for (int32_t k0 = 0; k0 < u; k0 += KCS) {
    const int kc = min(KCS, u - k0);
    const int kc4 = (kc + 3) & ~3;
    if (k0 > 0) __syncthreads();
#pragma unroll
    for (int i = 0; i < PA; ++i) {
      const int k = ka + 2 * i;
      if (k < kc4) As[k * LDA + t] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < PB; ++i) {
      const int k = kb + (256 / NB) * i;
      if (k < kc4) Bs[k * LDB + q] = rb[i];
    }
    if (k0 + KCS < u) fetch(k0 + KCS);
    __syncthreads();
    if (32 * wv < nrow) tile_mma<MFMA>(As, Bs, kc4, ncb, lane, wv, acc);
  }
  const int li = lane & 15, lr = lane >> 4;
#pragma unroll
  for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 16 * jb + lr + 4 * r;
        const int i = 32 * wv + 16 * ib + li;
        if (i < nrow && R0 + i >= w && j < w) P[(int64_t)j * m + R0 + i] = -acc[jb][ib][r];
      }
}

// Z_RC of a DENSE-TAIL front through the dense-tail kernel's machinery (k_dense_b): the R x R part of Z is the trailing
// dense matrix, so nothing is searched.  Item = (front s, 256 rows of R, a range of later tail fronts K as the reduction
// dimension):   acc[256 x w] += sum_{k in cols(K)} Z(g_i, g_k) Y[k, :].
//  * B operand: Y^T rows (k-row = 128 contiguous doubles in the front's scratch: k_sinv_y writes Y TRANSPOSED for tail
//    fronts), by LDS-DMA, 64 deep, double-buffered -- exactly k_dense_b's B stream;
//  * A operand: Z(g_i, g_k) straight from the panels into registers.  g_i >= g_k: the entry sits in the panel of K at row
//    g_i (k_dense_b's access: 16 consecutive rows = 128 contiguous bytes); g_i < g_k: it sits in the panel that owns column
//    g_i, at row g_k -- "transposed": a lane reads along the reduction dimension (4 consecutive k = 32 contiguous bytes,
//    the rest of the 128-byte line is used by the next k-steps out of L1).  Chosen per element, so the blocks on the
//    diagonal need no special case.
//  * the sum is ADDED to the panel with fp64 atomics (the front's rows were zeroed after Y was taken): several K ranges
//    of one row tile run as separate workgroups, which is what keeps a launch at >= 1000 items for every front.
struct SinvWork {
  int32_t front;   // tail front s
  int32_t q;       // rows [256 q, 256 q + 256) of R
  int32_t ka, kb;  // source fronts [ka, kb) (absolute front ids, all > s)
};
__global__ __launch_bounds__(512, 1) void k_sinv_tail(DevSym S, int32_t dense_first, const SinvWork* __restrict__ work, double* L,
                                                      const double* __restrict__ Ybuf, const int64_t* __restrict__ yoff,
                                                      const int32_t* __restrict__ col_front, const double* __restrict__ zeros) {
  extern __shared__ __attribute__((aligned(16))) double smem[];  // [2][KBA][LDB]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const SinvWork wk = work[blockIdx.x];
  const int32_t s = wk.front;
  const int32_t c0 = S.sn_start[s], w = S.sn_start[s + 1] - c0;
  const int32_t m = S.n - c0, u = m - w;
  const int32_t c1 = c0 + w;                 // first global column of R
  const int32_t r0 = 256 * wk.q;             // first row of the item inside R
  const int32_t nrow = min(256, u - r0);
  const int32_t ia = 32 * wv + li, ib_ = ia + 16;
  const int32_t gia = c1 + r0 + (ia < nrow ? ia : 0), gib = c1 + r0 + (ib_ < nrow ? ib_ : 0);  // global labels of this lane's rows
  // "transposed" bases: element index of Z(g_k, g_i) = baseT + g_k
  int64_t bTa, bTb;
  {
    const int32_t oa = col_front[gia], ob = col_front[gib];
    const int32_t ca = S.sn_start[oa], cb = S.sn_start[ob];
    bTa = S.sn_loff[oa] + (int64_t)(gia - ca) * (S.n - ca) - ca;
    bTb = S.sn_loff[ob] + (int64_t)(gib - cb) * (S.n - cb) - cb;
  }
  const double* YT = Ybuf + yoff[s];         // [u][128]
  const int32_t b_off = 2 * lane < NB ? 2 * lane : 0;
  const double* zsrc = zeros + 2 * lane;
  struct Chunk { int64_t uN; int32_t md, kc, g0; };  // uN: element index of Z(0-th row label, first k) minus the row label: N address = uN + (k - g0) md + g_i
  int32_t kd = wk.ka, kk0 = 0;
  auto next_chunk = [&]() {
    const int32_t cK = S.sn_start[kd], wK = S.sn_start[kd + 1] - cK;
    Chunk c;
    c.md = __builtin_amdgcn_readfirstlane(S.n - cK);
    c.g0 = __builtin_amdgcn_readfirstlane(cK + kk0);
    c.uN = uniform_i64(S.sn_loff[kd] + (int64_t)kk0 * c.md - cK);
    c.kc = __builtin_amdgcn_readfirstlane(min(KBA, wK - kk0));
    kk0 += KBA;
    if (kk0 >= wK) { kk0 = 0; ++kd; }
    return c;
  };
  auto issue_B = [&](const Chunk& c, int b) {
    double* Bs = smem + b * KBA * LDB;
    const double* src = YT + (int64_t)(c.g0 - c1) * NB + b_off;
#pragma unroll
    for (int i = 0; i < KBA / 8; ++i) {
      const int kr = wv + 8 * i;
      __builtin_amdgcn_global_load_lds((gl_vptr)(kr < c.kc ? src + (int64_t)kr * NB : zsrc), (lds_vptr)(Bs + kr * LDB), 16, 0, 0);
    }
  };
  auto load_A = [&](const Chunk& c, int sc, double (&ra)[4][2]) {
    const int klast = (c.kc - 1) & ~3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kq = min(16 * sc + 4 * q, klast) + lk;      // k inside the chunk (lanes past a short chunk: B is zero there)
      const int32_t gk = c.g0 + min(kq, c.kc - 1);
      const int64_t eN = c.uN + (int64_t)(gk - c.g0) * c.md;
      ra[q][0] = L[gia >= gk ? eN + gia : bTa + gk];
      ra[q][1] = L[gib >= gk ? eN + gib : bTb + gk];
    }
  };
  d4 acc16[NJB][2];
#pragma unroll
  for (int a = 0; a < NJB; ++a) { acc16[a][0] = (d4){0.0, 0.0, 0.0, 0.0}; acc16[a][1] = (d4){0.0, 0.0, 0.0, 0.0}; }
  if (wk.ka >= wk.kb || nrow <= 0) return;
  double rA[2][4][2];
  double bf[2][NJB];
  auto ldB = [&](const double* Bc, int k4, double (&b)[NJB]) {
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) b[jb] = Bc[(k4 + lk) * LDB + 16 * jb + li];
  };
  auto mma = [&](const double (&b)[NJB], double a0, double a1) {
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) {
      acc16[jb][0] = mfma_f64(b[jb], a0, acc16[jb][0]);
      acc16[jb][1] = mfma_f64(b[jb], a1, acc16[jb][1]);
    }
  };
  Chunk cur = next_chunk();
  issue_B(cur, 0);
  load_A(cur, 0, rA[0]);
  bool more = kd < wk.kb;
  Chunk nxt = cur;
  if (more) {
    nxt = next_chunk();
    issue_B(nxt, 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  ldB(smem, 0, bf[0]);
  while (true) {
    const double* Bc = smem + buf * KBA * LDB;
    const double* Bn = smem + (buf ^ 1) * KBA * LDB;
    bool more2 = false;
    Chunk nn = nxt;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int sc = t >> 2, q = t & 3;
      if (q == 0) {
        if (sc < 3) load_A(cur, sc + 1, rA[(sc + 1) & 1]);
        else if (more) load_A(nxt, 0, rA[0]);
      }
      if (t < 15) {
        ldB(Bc, 4 * (t + 1), bf[(t + 1) & 1]);
      } else if (more) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ldB(Bn, 0, bf[0]);
        more2 = kd < wk.kb;
        if (more2) {
          nn = next_chunk();
          issue_B(nn, buf);
        }
      }
      mma(bf[t & 1], rA[sc & 1][q][0], rA[sc & 1][q][1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!more) break;
    cur = nxt;
    nxt = nn;
    more = more2;
    buf ^= 1;
  }
  double* P = L + S.sn_loff[s];
#pragma unroll
  for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ib == 0 ? ia : ib_, jc = 16 * jb + lk + 4 * r;
        if (i < nrow && jc < w) unsafeAtomicAdd(&P[(int64_t)jc * m + w + r0 + i], -acc16[jb][ib][r]);
      }
}

// rows [w, m) of every column of the panels of `fronts` <- 0 (before the atomic accumulation of Z_RC); one workgroup per
// (front, column)
__global__ __launch_bounds__(256) void k_sinv_zero(DevSym S, const int32_t* __restrict__ fronts, double* __restrict__ L) {
  const int32_t s = fronts[blockIdx.x];
  const int32_t w = S.sn_start[s + 1] - S.sn_start[s];
  const int32_t m = (int32_t)(S.sn_rowptr[s + 1] - S.sn_rowptr[s]);
  const int32_t j = blockIdx.y;
  if (j >= w) return;
  double* col = L + S.sn_loff[s] + (int64_t)j * m;
  for (int32_t i = w + threadIdx.x; i < m; i += 256) col[i] = 0.0;
}

// diagonal block <- L11^-T L11^-1 (lower triangle; the strict upper part stays zero), one workgroup per front
__global__ __launch_bounds__(256) void k_sinv_cc0(DevSym S, const int32_t* __restrict__ fronts, double* __restrict__ L,
                                                  const double* __restrict__ invD) {
  const int32_t s = fronts[blockIdx.x];
  const int32_t w = S.sn_start[s + 1] - S.sn_start[s];
  const int32_t m = (int32_t)(S.sn_rowptr[s + 1] - S.sn_rowptr[s]);
  double* P = L + S.sn_loff[s];
  const double* I = invD + S.inv_off[s];  // X = L11^-1, column-major, lower triangular
  for (int idx = threadIdx.x; idx < w * w; idx += 256) {
    const int b = idx / w, a = idx - b * w;  // entry (row a, column b), a >= b
    if (a < b) continue;
    double sum = 0.0;
    for (int k = a; k < w; ++k) sum += I[(int64_t)a * w + k] * I[(int64_t)b * w + k];  // sum_k X[k][a] X[k][b], X[k][a] = 0 for k < a
    P[(int64_t)b * m + a] = sum;
  }
}

// diagonal block -= Y_I^T Z_RC,I for one 128-row tile I of R (fp64 atomics: the tiles of a front add up in any order)
template <bool MFMA>
__global__ __launch_bounds__(256) void k_sinv_cc(DevSym S, const int32_t* __restrict__ tiles, double* L,
                                                 const double* __restrict__ Ybuf, const int64_t* __restrict__ yoff,
                                                 int32_t dense_first) {
  __shared__ __attribute__((aligned(16))) double As[KCS * LDA];  // [k = row of the tile][a] = Y[row][a]
  __shared__ __attribute__((aligned(16))) double Bs[KCS * LDB];  // [k][b] = Z_RC[row][b]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int32_t g = tiles[blockIdx.x];
  const int32_t s = S.tile_front[g];
  const int32_t ti = (int32_t)(g - S.tile_base[s]);
  const int32_t w = S.sn_start[s + 1] - S.sn_start[s];
  const int32_t m = (int32_t)(S.sn_rowptr[s + 1] - S.sn_rowptr[s]);
  const int32_t R0 = ti * TM;
  const int32_t nrow = min(TM, m - R0);
  if (R0 + nrow <= w) return;
  const int32_t u = m - w;
  const int ncb = (w + 15) >> 4;
  double* P = L + S.sn_loff[s];
  const double* Y = Ybuf + yoff[s];
  d4 acc[NJB][2];
#pragma unroll
  for (int a = 0; a < NJB; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = (d4){0.0, 0.0, 0.0, 0.0};
  const int cq = tid & 127, kh = tid >> 7;  // column a (or b) and row parity inside a 16-row chunk
  for (int32_t i0 = 0; i0 < nrow; i0 += KCS) {
    if (i0 > 0) __syncthreads();
#pragma unroll
    for (int x = 0; x < KCS / 2; ++x) {
      const int k = kh + 2 * x;
      const int32_t row = R0 + i0 + k;  // panel row
      const bool on = i0 + k < nrow && row >= w && cq < w;
      As[k * LDA + cq] = on ? (s >= dense_first ? Y[(int64_t)(row - w) * NB + cq] : Y[(int64_t)cq * u + (row - w)]) : 0.0;
      Bs[k * LDB + cq] = on ? P[(int64_t)cq * m + row] : 0.0;
    }
    __syncthreads();
    if (32 * wv < w) tile_mma<MFMA>(As, Bs, KCS, ncb, lane, wv, acc);  // D[b][a] += sum_k Z_RC[k][b] Y[k][a]
  }
  const int li = lane & 15, lr = lane >> 4;
#pragma unroll
  for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int b = 16 * jb + lr + 4 * r;      // column of the diagonal block
        const int a = 32 * wv + 16 * ib + li;    // row
        if (a < w && b <= a) unsafeAtomicAdd(&P[(int64_t)b * m + a], -acc[jb][ib][r]);
      }
}

// part[block] = sum over this block's slots of vals[e] * Z[asm_dst[e]] (all slots once); the caller doubles it and takes
// the diagonal slots (k_sinv_trace_diag) off once
__global__ __launch_bounds__(256) void k_sinv_trace(int64_t nnz, const int64_t* __restrict__ asm_dst, const double* __restrict__ vals,
                                                    const double* __restrict__ Z, double* __restrict__ part) {
  double acc = 0.0;
  for (int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x; e < nnz; e += (int64_t)gridDim.x * 256) acc += vals[e] * Z[asm_dst[e]];
  __shared__ double red[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// diag_vals: per permuted column j the value of the matrix on the diagonal: vals[pat_colptr[j]] (general matrix; the
// diagonal slot comes first in its column) or vals[j] (diagonal-only matrix)
__global__ __launch_bounds__(256) void k_sinv_trace_diag(int32_t n, const int64_t* __restrict__ pat_colptr, const int64_t* __restrict__ diag_dst,
                                                         const double* __restrict__ vals, int32_t is_diag, const double* __restrict__ Z,
                                                         double* __restrict__ part) {
  double acc = 0.0;
  for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < n; j += (int64_t)gridDim.x * 256)
    acc += (is_diag ? vals[j] : vals[pat_colptr[j]]) * Z[diag_dst[j]];
  __shared__ double red[4];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------
// IBD (numerator relationship) values ON THE DEVICE, straight into the value slots of the engine (SURVEY 8f rank 1;
// reference scilmm/Matrices/Numerator.py:5-38 computes A = L D L^T in interpreted Python).  Tabular recursion on the
// known pattern (pairs with a common ancestor): for individuals i > j in pedigree order (parents before children, so i
// is never an ancestor of j)
//     A[i, j] = 1/2 (A[f_i, j] + A[m_i, j]),      A[i, i] = 1 + 1/2 A[f_i, m_i],      unknown parent: 0,
// where the entries on the right are looked up (bisection in a sorted pattern column, like csrc/dominance.hip) among the
// slots already computed.  An entry depends only on entries whose generation SUM gen(a) + gen(b) is smaller, so the slots
// are sorted by that sum once (one radix-sort pass) and each sum is one launch: ~2 x generations launches, every slot
// written exactly once, no atomics.  Values are dyadic rationals: halving and adding them is exact.
__global__ __launch_bounds__(256) void k_ibd_keys(int32_t n, const int64_t* __restrict__ colptr, const int32_t* __restrict__ prow,
                                                  const int32_t* __restrict__ perm, const int32_t* __restrict__ gen,
                                                  uint8_t* __restrict__ key, uint32_t* __restrict__ slot) {
  // one wave per pattern column c (permuted labels): key = generation sum of the pair, slot = identity
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t c = wave; c < n; c += nw) {
    const int32_t gc = gen[perm[c]];
    for (int64_t e = colptr[c] + lane; e < colptr[c + 1]; e += 64) {
      key[e] = (uint8_t)(gc + gen[perm[prow[e]]]);
      slot[e] = (uint32_t)e;
    }
  }
}

__global__ void k_ibd_bounds(int64_t nnz, const uint8_t* __restrict__ skey, int64_t* __restrict__ pass_ptr) {
  // pass_ptr[k] = first position of key k in the sorted key array (entries of absent keys are fixed up on the host)
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nnz) return;
  if (t == 0 || skey[t] != skey[t - 1]) pass_ptr[skey[t]] = t;
}

__device__ __forceinline__ double ibd_look(int32_t p, int32_t q, const int64_t* __restrict__ colptr, const int32_t* __restrict__ prow,
                                           const int32_t* __restrict__ iperm, const double* __restrict__ vals) {
  if (p < 0 || q < 0) return 0.0;
  const int32_t P = iperm[p], Q = iperm[q];
  const int32_t c = min(P, Q), r = max(P, Q);
  int64_t lo = colptr[c], hi = colptr[c + 1];
  while (lo < hi) {  // first slot of column c whose row is >= r (rows ascending, diagonal first)
    const int64_t mid = (lo + hi) >> 1;
    if (prow[mid] < r) lo = mid + 1; else hi = mid;
  }
  return (lo < colptr[c + 1] && prow[lo] == r) ? vals[lo] : 0.0;  // no stored entry: no common ancestor, A = 0
}

__global__ __launch_bounds__(256) void k_ibd_pass(int64_t cnt, const uint32_t* __restrict__ slots, int32_t n,
                                                  const int64_t* __restrict__ colptr, const int32_t* __restrict__ prow,
                                                  const int32_t* __restrict__ perm, const int32_t* __restrict__ iperm,
                                                  const int32_t* __restrict__ par, double* vals) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= cnt) return;
  const int64_t e = slots[t];
  int32_t lo = 0, hi = n;  // column of slot e: last c with colptr[c] <= e
  while (hi - lo > 1) {
    const int32_t mid = (lo + hi) >> 1;
    if (colptr[mid] <= e) lo = mid; else hi = mid;
  }
  const int32_t a = perm[lo], b = perm[prow[e]];
  const int32_t i = max(a, b), j = min(a, b);
  const int32_t f = par[2 * i], m = par[2 * i + 1];
  double v;
  if (i == j) v = 1.0 + 0.5 * ibd_look(f, m, colptr, prow, iperm, vals);
  else v = 0.5 * (ibd_look(f, j, colptr, prow, iperm, vals) + ibd_look(m, j, colptr, prow, iperm, vals));
  vals[e] = v;
}

// Dominance relationship values ON THE DEVICE in the engine's slot order, from the IBD values already resident in HBM
// (reference scilmm/Matrices/Dominance.py:12-43; the CSR-layout form is csrc/dominance.hip):
//     D[a, b] = 1/4 (A[f_a, f_b] A[m_a, m_b] + A[f_a, m_b] A[m_a, f_b]),  D[a, a] = 1,  unknown parent: 0,
// one wave per pattern column (permuted labels), the four look-ups by bisection in the permuted pattern columns like
// k_ibd_pass.  Products and the sum rounded one by one (no fma contraction) in the reference's order: the values equal
// NumPy's bit for bit, and (a, b) / (b, a) give the same bits, so the lower-triangle slot holds either.
__global__ __launch_bounds__(256) void k_dom_slots(int32_t n, const int64_t* __restrict__ colptr, const int32_t* __restrict__ prow,
                                                   const int32_t* __restrict__ perm, const int32_t* __restrict__ iperm,
                                                   const int32_t* __restrict__ par, const double* __restrict__ A, double* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t c = wave; c < n; c += nw) {
    const int32_t a = perm[c];
    const int32_t fa = par[2 * (int64_t)a], ma = par[2 * (int64_t)a + 1];
    for (int64_t e = colptr[c] + lane; e < colptr[c + 1]; e += 64) {
      const int32_t b = perm[prow[e]];
      double v = 1.0;
      if (a != b) {
        const int32_t fb = par[2 * (int64_t)b], mb = par[2 * (int64_t)b + 1];
        const double p1 = __dmul_rn(ibd_look(fa, fb, colptr, prow, iperm, A), ibd_look(ma, mb, colptr, prow, iperm, A));
        const double p2 = __dmul_rn(ibd_look(fa, mb, colptr, prow, iperm, A), ibd_look(ma, fb, colptr, prow, iperm, A));
        v = __dmul_rn(0.25, __dadd_rn(p1, p2));
      }
      out[e] = v;
    }
  }
}

}  // namespace scilmm
