// RETIRED kernel variants -- tuning-harness material only (csrc/tools/dense_bench.hip); nothing in libscilmm_hip.so
// includes this file.  Each of them was measured against the kernel that replaced it on the hot path (DESIGN.md
// section 4: same speed or slower) and produces the same results:
//   k_update        -> k_update2   (staging interleaved with the MFMA k-steps)
//   k_update3       -> k_update2   (the v_mfma_f64_4x4x4 form: 67.5 vs 66.2 ms at 100k)
//   k_dense<MF>     -> k_dense_a   (register-staged dense-tail update, both f64 MFMA forms)
//   k_dense_g       -> k_dense_a   (both operands by LDS-DMA)
//   k_dense_a       -> k_dense_b   (A from registers, B by LDS-DMA, without the in-wave software pipeline of round 3)
//   k_update_compact               (compact-coordinate update: 66 -> 91..96 ms at 100k)
//   k_trsm4         -> k_trsm      (the 4x4x4 form)
// Include AFTER ../kernels.hip.h.
#pragma once
#include "../kernels.hip.h"

namespace scilmm {

// ABL (diagnostic builds only, selected by SCILMM_ABLATE): 0 = real kernel, 1 = no MFMAs, 2 = no global loads.
template <bool MFMA, int ABL = 0>
__global__ __launch_bounds__(UPD_THREADS, SCILMM_UPD_WAVES) void k_update(DevSym S, const UpdWork* __restrict__ work,
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
  // eight waves: wave wv owns target rows [16 wv, 16 wv + 16) and all NB columns (NJB accumulator tiles)
  d4 acc[NJB];
#pragma unroll
  for (int a = 0; a < NJB; ++a) acc[a] = (d4){0.0, 0.0, 0.0, 0.0};
  // this thread's fixed roles in the staging: A row t with k phase kpa (of KA); B column q with k phase kpb (of KB)
  constexpr int KA = UPD_THREADS / TM, KB = UPD_THREADS / NB;
  const int t = tid % TM, kpa = tid / TM;
  const int q = tid % NB, kpb = tid / NB;
  // per-buffer record of what this thread wrote (so that it can clear exactly that)
  int w_ip[2] = {-1, -1}, w_jp[2] = {-1, -1}, w_kc[2] = {0, 0};
  // "next chunk" cursor
  int64_t cn = cb;
  int k0n = 0;
  ComboDesc dn = combos[cn];
  ComboDesc dnext = combos[min(cn + 1, ce - 1)];  // descriptor of the following combo, fetched one combo ahead
  int ipn = -1, jpn = -1;
  auto locate = [&]() {
    // tile position of this thread's descendant row / target column for combo dn
    ipn = -1;
    jpn = -1;
    if (t < dn.nt) {
      if (dn.ip0 >= 0) {
        ipn = dn.ip0 + t;
      } else {
        const int32_t lab = S.sn_rows[dn.rowoff + dn.ta + t];
        int lo = 0, hi = nrow;
        while (lo < hi) {
          int mid = (lo + hi) >> 1;
          if (rowlab[mid] < lab) lo = mid + 1; else hi = mid;
        }
        ipn = lo;
      }
    }
    if (q < dn.nq) jpn = (dn.jp0 >= 0) ? dn.jp0 + q : S.sn_rows[dn.rowoff + dn.p0 + q] - c0;
  };
  double ra[KC / KA], rb[KC / KB];
  int kcn = 0;
  auto prefetch = [&]() {
    kcn = min(KC, dn.wd - k0n);
    const double* Pd = L + dn.loff + (int64_t)k0n * dn.md;
    const int64_t md = dn.md;
    if (ABL == 2) {
#pragma unroll
      for (int i = 0; i < KC / KA; ++i) ra[i] = 0.0;
#pragma unroll
      for (int i = 0; i < KC / KB; ++i) rb[i] = 0.0;
      return;
    }
    if (ipn >= 0) {
      const double* pa = Pd + (int64_t)kpa * md + dn.ta + t;
      if (kcn == KC) {
#pragma unroll
        for (int i = 0; i < KC / KA; ++i) ra[i] = pa[(int64_t)(KA * i) * md];
      } else {
#pragma unroll
        for (int i = 0; i < KC / KA; ++i) ra[i] = (kpa + KA * i < kcn) ? pa[(int64_t)(KA * i) * md] : 0.0;
      }
    }
    if (jpn >= 0) {
      const double* pb = Pd + (int64_t)kpb * md + dn.p0 + q;
      if (kcn == KC) {
#pragma unroll
        for (int i = 0; i < KC / KB; ++i) rb[i] = pb[(int64_t)(KB * i) * md];
      } else {
#pragma unroll
        for (int i = 0; i < KC / KB; ++i) rb[i] = (kpb + KB * i < kcn) ? pb[(int64_t)(KB * i) * md] : 0.0;
      }
    }
  };
  // uniform (per-workgroup) record of the mapping last staged into each buffer
  int u_ip0[2] = {-2, -2}, u_nt[2] = {0, 0}, u_jp0[2] = {-2, -2}, u_nq[2] = {0, 0}, u_kc[2] = {0, 0};
  int s_ilo[2] = {0, 0}, s_ihi[2] = {TM - 1, TM - 1}, s_jb0[2] = {0, 0}, s_jb1[2] = {NJB - 1, NJB - 1};
  auto stage = [&](int b) {
    double* As = Abuf + b * KC * LDA;
    double* Bs = Bbuf + b * KC * LDB;
    // The cells a thread writes belong to the descendant row/column it carries, so when the mapping of
    // the incoming chunk differs from what the buffer holds, every thread first clears its own old cells
    // and a barrier separates the clears from the new writes (another thread may now own that cell).
    // Consecutive chunks of the dense chains share one mapping: no clear, no extra barrier.
    const bool same = dn.ip0 >= 0 && dn.ip0 == u_ip0[b] && dn.nt == u_nt[b] && dn.jp0 >= 0 && dn.jp0 == u_jp0[b] &&
                      dn.nq == u_nq[b] && kcn >= u_kc[b];
    if (!same) {
      if (w_ip[b] >= 0) {
#pragma unroll
        for (int i = 0; i < KC / KA; ++i) {
          const int k = kpa + KA * i;
          if (k < w_kc[b]) As[k * LDA + w_ip[b]] = 0.0;
        }
      }
      if (w_jp[b] >= 0) {
#pragma unroll
        for (int i = 0; i < KC / KB; ++i) {
          const int k = kpb + KB * i;
          if (k < w_kc[b]) Bs[k * LDB + w_jp[b]] = 0.0;
        }
      }
      __syncthreads();
    }
    if (ipn >= 0) {
      double* wa = As + kpa * LDA + ipn;
      if (kcn == KC) {
#pragma unroll
        for (int i = 0; i < KC / KA; ++i) wa[KA * i * LDA] = ra[i];
      } else {
#pragma unroll
        for (int i = 0; i < KC / KA; ++i)
          if (kpa + KA * i < kcn) wa[KA * i * LDA] = ra[i];
      }
    }
    if (jpn >= 0) {
      double* wb = Bs + kpb * LDB + jpn;
      if (kcn == KC) {
#pragma unroll
        for (int i = 0; i < KC / KB; ++i) wb[KB * i * LDB] = rb[i];
      } else {
#pragma unroll
        for (int i = 0; i < KC / KB; ++i)
          if (kpb + KB * i < kcn) wb[KB * i * LDB] = rb[i];
      }
    }
    w_ip[b] = ipn;
    w_jp[b] = jpn;
    w_kc[b] = kcn;
    u_ip0[b] = dn.ip0;
    u_nt[b] = dn.nt;
    u_jp0[b] = dn.jp0;
    u_nq[b] = dn.nq;
    u_kc[b] = kcn;
    s_ilo[b] = dn.ilo;
    s_ihi[b] = dn.ihi;
    s_jb0[b] = dn.jlo >> 4;
    s_jb1[b] = min(dn.jhi >> 4, ncb - 1);
  };
  auto advance = [&]() -> bool {
    // move the cursor to the chunk after (cn, k0n); returns false at the end
    k0n += KC;
    if (k0n >= dn.wd) {
      ++cn;
      k0n = 0;
      if (cn >= ce) return false;
      dn = dnext;
      dnext = combos[min(cn + 1, ce - 1)];
      locate();
    }
    return true;
  };
  __syncthreads();  // rowlab + zeroed buffers visible
  locate();
  prefetch();
  stage(0);
  int kc4_cur = (kcn + 3) & ~3;
  bool more = advance();
  __syncthreads();
  int buf = 0;
  while (true) {
    if (more) prefetch();  // global loads of the next chunk in flight during the MFMAs
    if (ABL != 1 && 16 * wv <= s_ihi[buf] && 16 * wv + 15 >= s_ilo[buf])
      tile_mma8<MFMA>(Abuf + buf * KC * LDA, Bbuf + buf * KC * LDB, kc4_cur, s_jb0[buf], s_jb1[buf], lane, wv, acc);
    if (!more) break;
    stage(buf ^ 1);
    kc4_cur = (kcn + 3) & ~3;
    more = advance();
    __syncthreads();
    buf ^= 1;
  }
  // epilogue: D[M=j][N=i] -> panel(R0+i, j) (or the partial slot); 16 consecutive lanes = 128 contiguous bytes
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
// k_update3: k_update2 with the matrix pipe driven by v_mfma_f64_4x4x4f64 (four independent 4x4x4 blocks per
// instruction) instead of v_mfma_f64_16x16x4_f64.  Measured on this GPU (csrc/tools/mfma_sweep.hip,
// profiles/r2_mfma_sweep.txt): the 16x16x4 form issues every ~105 cycles per SIMD (49 TFLOP/s chip-wide, whatever the
// number of busy CUs), the 4x4x4 form every ~17 cycles for a quarter of the flops = 75 TFLOP/s, 95 % of the
// 78.6 TFLOP/s datasheet figure.  The price is operand bandwidth (64 + 64 operand values per 256 MACs instead of
// per 1024), paid from registers: per k-step a wave reads its 4 row operands (rows replicated over the 4 blocks:
// an LDS broadcast) and 8 column operands once and issues 8 x 4 MFMAs on them -- 12 LDS reads per 32 MFMAs.
// Lane roles of the instruction as probed on gfx950 (csrc/tools/mfma_probe.hip, profiles/r2_mfma_probe.txt):
//   A operand: lane = 16 k + 4 block + i      B operand: lane = 16 k + 4 block + j      D: lane = 16 i + 4 block + j
// Used here with the COLUMN operand as A, replicated over the blocks (4 target columns x 4 k), and the ROW operand
// as B (rows 4 block + j: 16 tile rows x 4 k), so one instruction updates a 16-row x 4-column piece of the tile and
// result lane l holds row (l & 15), column (l >> 4) of it: 16 consecutive lanes = 16 consecutive rows of one panel
// column (128 contiguous bytes in the epilogue, like the 16x16x4 form).  Wave wv owns target columns
// [16 wv, 16 wv + 16) x all 128 rows as 8 x 4 such pieces.
template <bool MFMA>
__global__ __launch_bounds__(UPD_THREADS, SCILMM_UPD_WAVES) void k_update3(DevSym S, const UpdWork* __restrict__ work,
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
  // acc4[p][q]: this lane's element of the piece (rows 16 p .., columns 16 wv + 4 q ..): one double per MFMA
  constexpr int NRB = TM / 16;
  double acc4[NRB][4];
#pragma unroll
  for (int a = 0; a < NRB; ++a)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) acc4[a][q4] = 0.0;
  const int l15 = lane & 15, l3 = lane & 3, lk4 = lane >> 4;
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
    const bool mine = wv >= SC.jb0 && wv <= SC.jb1;     // this wave's 16 columns lie inside the chunk's column span
    const int plo = SC.ilo >> 4, phi = SC.ihi >> 4;     // 16-row groups inside its row span
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      if (SW.valid) write_piece(i, Aw, Bw);
      if (SL.valid) load_piece(i);
      if (mine && 4 * i < kc4) {
        if (MFMA) {
          double cv[4];
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) cv[q4] = Bc[(4 * i + lk4) * LDB + 16 * wv + 4 * q4 + l3];
#pragma unroll
          for (int pr = 0; pr < NRB; ++pr)
            if (pr >= plo && pr <= phi) {
              const double rv = Ac[(4 * i + lk4) * LDA + 16 * pr + l15];
#pragma unroll
              for (int q4 = 0; q4 < 4; ++q4) acc4[pr][q4] = __builtin_amdgcn_mfma_f64_4x4x4f64(cv[q4], rv, acc4[pr][q4], 0, 0, 0);
            }
        } else {
          for (int k = 4 * i; k < 4 * i + 4; ++k)
#pragma unroll
            for (int pr = 0; pr < NRB; ++pr)
              if (pr >= plo && pr <= phi)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) acc4[pr][q4] += Bc[k * LDB + 16 * wv + 4 * q4 + lk4] * Ac[k * LDA + 16 * pr + l15];
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
  // epilogue: piece (p, q), lane l -> tile row 16 p + (l & 15), target column 16 wv + 4 q + (l >> 4)
  if (wk.slot < 0) {
    double* P = L + S.sn_loff[s];
#pragma unroll
    for (int pr = 0; pr < NRB; ++pr)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int j = 16 * wv + 4 * q4 + lk4;
        const int i = 16 * pr + l15;
        if (i < nrow && j < w) P[(int64_t)j * m + R0 + i] -= acc4[pr][q4];
      }
  } else {
    double* Q = scratch + (int64_t)wk.slot * (TM * NB);
#pragma unroll
    for (int pr = 0; pr < NRB; ++pr)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int j = 16 * wv + 4 * q4 + lk4;
        const int i = 16 * pr + l15;
        Q[j * TM + i] = acc4[pr][q4];
      }
  }
}

template <int MF, bool MFMA>
__global__ __launch_bounds__(512, 1) void k_dense(DevSym S, int32_t dense_first, const DenseWork* __restrict__ work,
                                                  double* __restrict__ L, double* __restrict__ scratch) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Abuf = smem;                      // [2][KC][LDA2]
  double* Bbuf = smem + 2 * KC * LDA2;      // [2][KC][LDB]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
#ifdef SCILMM_DENSE_CLK
  const unsigned long long clk_w0 = wall_clock64(), clk_c0 = clock64();
#endif
  const DenseWork wk = work[blockIdx.x];
  const int32_t j = wk.front;
  const int32_t c0j = S.sn_start[j], wj = S.sn_start[j + 1] - c0j;
  const int32_t mj = S.n - c0j;             // dense tail: the rows of front j are the labels c0j .. n-1
  const int32_t R0 = wk.ti0 * TM;
  const int32_t nrow = min(wk.ntiles * TM, mj - R0);
#if SCILMM_DENSE_VEC
  // staging roles, 16 bytes per access: A row PAIR pa (rows 2 pa, 2 pa + 1) with k phase ka (of 4), B column pair pb
  // with k phase kb (of 8).  Rows / columns of a panel column are contiguous, so a pair is one 16-byte global load
  // (8-byte aligned: the leading dimension may be odd) and one ds_write_b128: half the load and store instructions.
  typedef double d2 __attribute__((ext_vector_type(2)));
  constexpr int NPA = KC / 4, NPB = KC / 8;
  const int pa = tid & (DTR / 2 - 1), ka = tid >> 7;
  const int pb = tid & (NB / 2 - 1), kb = tid >> 6;
  // a pair that starts inside the range is loaded where it is (at an odd edge its second element is the first one past
  // the range: valid memory -- the next rows of the column, the next column, or the slack behind L -- and its LDS cell
  // is never read back into a stored result); pairs beyond the range re-read pair 0
  const int ra_row = 2 * pa < nrow ? 2 * pa : 0, rb_col = 2 * pb < wj ? 2 * pb : 0;
  d2 ra[NPA], rb[NPB];
  int32_t kd = wk.k0;   // descendant cursor of the chunk being loaded
  int32_t kk0 = 0;      // first column of that chunk inside the descendant
  int kc_ld = 0;        // depth of the chunk held in ra / rb
  auto load_chunk = [&]() {
    const int32_t d = dense_first + kd;
    const int32_t c0d = S.sn_start[d], wd = S.sn_start[d + 1] - c0d;
    const int64_t md = S.n - c0d;
    const double* Pd = L + S.sn_loff[d] + (int64_t)kk0 * md + (c0j - c0d);
    kc_ld = min(KC, wd - kk0);
    if (SCILMM_DENSE_ABL != 2 || (kd == wk.k0 && kk0 == 0)) {
#pragma unroll
      for (int i = 0; i < NPA; ++i) {
        const double* q = Pd + (int64_t)min(ka + 4 * i, kc_ld - 1) * md + R0 + ra_row;
        __builtin_memcpy(&ra[i], q, 16);
      }
#pragma unroll
      for (int i = 0; i < NPB; ++i) {
        const double* q = Pd + (int64_t)min(kb + 8 * i, kc_ld - 1) * md + rb_col;
        __builtin_memcpy(&rb[i], q, 16);
      }
    }
    kk0 += KC;
    if (kk0 >= wd) { kk0 = 0; ++kd; }
  };
  auto store_chunk = [&](int b) {
    double* As = Abuf + b * KC * LDA2;
    double* Bs = Bbuf + b * KC * LDB;
    const d2 zero = (d2){0.0, 0.0};
#pragma unroll
    for (int i = 0; i < NPA; ++i) *(d2*)&As[(ka + 4 * i) * LDA2 + 2 * pa] = (ka + 4 * i < kc_ld) ? ra[i] : zero;
#pragma unroll
    for (int i = 0; i < NPB; ++i) *(d2*)&Bs[(kb + 8 * i) * LDB + 2 * pb] = (kb + 8 * i < kc_ld) ? rb[i] : zero;
  };
#else
  // staging roles: A row ta with k phase ka (of 2), B column tb with k phase kb (of 4)
  constexpr int NPA = KC / 2, NPB = KC / 4;
  const int ta = tid & (DTR - 1), ka = tid >> 8;
  const int tb = tid & (NB - 1), kb = tid >> 7;
  const int ra_row = min(ta, nrow - 1), rb_col = min(tb, wj - 1);  // clamped: every load is unconditional
  double ra[NPA], rb[NPB];
  int32_t kd = wk.k0;   // descendant cursor of the chunk being loaded
  int32_t kk0 = 0;      // first column of that chunk inside the descendant
  int kc_ld = 0;        // depth of the chunk held in ra / rb
  auto load_chunk = [&]() {
    const int32_t d = dense_first + kd;
    const int32_t c0d = S.sn_start[d], wd = S.sn_start[d + 1] - c0d;
    const int64_t md = S.n - c0d;
    const double* Pd = L + S.sn_loff[d] + (int64_t)kk0 * md + (c0j - c0d);
    kc_ld = min(KC, wd - kk0);
    if (SCILMM_DENSE_ABL != 2 || (kd == wk.k0 && kk0 == 0)) {
#pragma unroll
      for (int i = 0; i < NPA; ++i) ra[i] = Pd[(int64_t)min(ka + 2 * i, kc_ld - 1) * md + R0 + ra_row];
#pragma unroll
      for (int i = 0; i < NPB; ++i) rb[i] = Pd[(int64_t)min(kb + 4 * i, kc_ld - 1) * md + rb_col];
    }
    kk0 += KC;
    if (kk0 >= wd) { kk0 = 0; ++kd; }
  };
  auto store_chunk = [&](int b) {
    double* As = Abuf + b * KC * LDA2;
    double* Bs = Bbuf + b * KC * LDB;
#pragma unroll
    for (int i = 0; i < NPA; ++i) As[(ka + 2 * i) * LDA2 + ta] = (ka + 2 * i < kc_ld) ? ra[i] : 0.0;
#pragma unroll
    for (int i = 0; i < NPB; ++i) Bs[(kb + 4 * i) * LDB + tb] = (kb + 4 * i < kc_ld) ? rb[i] : 0.0;
  };
#endif
  // accumulators (64 doubles per lane in both forms)
  d4 acc16[NJB][2];
  double acc4[DTR / 16][4];
  if (MF == 16) {
#pragma unroll
    for (int a = 0; a < NJB; ++a) { acc16[a][0] = (d4){0.0, 0.0, 0.0, 0.0}; acc16[a][1] = (d4){0.0, 0.0, 0.0, 0.0}; }
  } else {
#pragma unroll
    for (int a = 0; a < DTR / 16; ++a)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) acc4[a][q4] = 0.0;
  }
  const int li = lane & 15, lk = lane >> 4, l3 = lane & 3;
  if (wk.k0 >= wk.k1) return;
  load_chunk();
  int kc_cur = kc_ld;
  store_chunk(0);
  __syncthreads();
  int buf = 0;
  while (true) {
    const bool more = kd < wk.k1;
    if (more) load_chunk();  // global loads of the next chunk in flight during the MFMAs of this one
    const double* Ac = Abuf + buf * KC * LDA2;
    const double* Bc = Bbuf + buf * KC * LDB;
    const int kc4 = (kc_cur + 3) & ~3;
    if (MFMA && MF == 16) {
      // wave wv: rows [32 wv, 32 wv + 32) x all 128 columns;  D[M = column][N = row]
#pragma unroll 2
      for (int k4 = 0; k4 < kc4; k4 += 4) {
        const double a0 = Ac[(k4 + lk) * LDA2 + 32 * wv + li], a1 = Ac[(k4 + lk) * LDA2 + 32 * wv + 16 + li];
        double b[NJB];
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) b[jb] = Bc[(k4 + lk) * LDB + 16 * jb + li];
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) {
          acc16[jb][0] = mfma_f64(b[jb], a0, acc16[jb][0]);
          acc16[jb][1] = mfma_f64(b[jb], a1, acc16[jb][1]);
        }
      }
    } else if (MFMA) {
      // wave wv: columns [16 wv, 16 wv + 16) x all 256 rows as 16 x 4 pieces of 16 rows x 4 columns
#pragma unroll 2
      for (int k4 = 0; k4 < kc4; k4 += 4) {
        double cv[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) cv[q4] = Bc[(k4 + lk) * LDB + 16 * wv + 4 * q4 + l3];
#pragma unroll
        for (int pr = 0; pr < DTR / 16; ++pr) {
          const double rv = Ac[(k4 + lk) * LDA2 + 16 * pr + li];
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) acc4[pr][q4] = __builtin_amdgcn_mfma_f64_4x4x4f64(cv[q4], rv, acc4[pr][q4], 0, 0, 0);
        }
      }
    } else {
      // scalar restatement in the layout of the 4x4x4 form (debug path)
      for (int k = 0; k < kc4; ++k)
#pragma unroll
        for (int pr = 0; pr < DTR / 16; ++pr)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) acc4[pr][q4] += Bc[k * LDB + 16 * wv + 4 * q4 + lk] * Ac[k * LDA2 + 16 * pr + li];
    }
    if (!more) break;
    if (SCILMM_DENSE_ABL != 3) store_chunk(buf ^ 1);
    kc_cur = kc_ld;
    __syncthreads();
    buf ^= 1;
  }
  if (SCILMM_DENSE_ABL == 1) {  // keep the accumulators alive without the stores
    double sacc = 0.0;
    if (MF == 16) {
#pragma unroll
      for (int a = 0; a < NJB; ++a) sacc += acc16[a][0][0] + acc16[a][0][1] + acc16[a][0][2] + acc16[a][0][3] + acc16[a][1][0] + acc16[a][1][1] + acc16[a][1][2] + acc16[a][1][3];
    } else {
#pragma unroll
      for (int a = 0; a < DTR / 16; ++a) sacc += acc4[a][0] + acc4[a][1] + acc4[a][2] + acc4[a][3];
    }
    if (sacc == 123.456) scratch[0] = sacc;
    return;
  }
#ifdef SCILMM_DENSE_CLK
  if (tid == 0) {
    atomicAdd(&g_dense_clk[0], wall_clock64() - clk_w0);
    atomicAdd(&g_dense_clk[1], clock64() - clk_c0);
  }
#endif
  // epilogue: tile h = 0 / 1 (rows [128 h, 128 h + 128) of the item) -> panel or its partial slab
  double* P = L + S.sn_loff[j];
  auto put = [&](int i, int jc, double v) {
    const int h = i >> 7;
    const int32_t slot = h ? wk.slot1 : wk.slot0;
    if (slot < 0) {
      if (i < nrow && jc < wj) P[(int64_t)jc * mj + R0 + i] -= v;
    } else if (h < wk.ntiles) {
      scratch[(int64_t)slot * (TM * NB) + jc * TM + (i & (TM - 1))] = v;
    }
  };
  if (MFMA && MF == 16) {
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) put(32 * wv + 16 * ib + li, 16 * jb + lk + 4 * r, acc16[jb][ib][r]);
  } else {
#pragma unroll
    for (int pr = 0; pr < DTR / 16; ++pr)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) put(16 * pr + li, 16 * wv + 4 * q4 + lk, acc4[pr][q4]);
  }
}

// ------------------------------------------------------------------------------------------------
// k_dense_g: k_dense<16, true> with the staging done by LDS-DMA (global_load_lds_dwordx4, gfx950): a wave-instruction
// moves 64 x 16 bytes from per-lane global addresses straight into 1 KiB of contiguous LDS -- exactly one k-row of
// the B image (128 columns) or half a k-row of the A image (128 of the 256 rows), the padding between k-rows stays.
// No staging registers, no ds_write pass, no selects: the copy of chunk c+1 into the other buffer is issued right
// after the barrier that retired that buffer's readers and lands while the MFMAs of chunk c run; one vmcnt(0) +
// barrier per chunk.  k-rows past the end of a descendant are sourced from a zero page (`zeros`, >= 1 KiB), rows /
// columns past the item's edge from row / column 0 (their accumulators are never stored).
// (Measured and dropped: spreading the six DMA issues of a wave behind the MFMAs of the four k-steps instead of ahead
// of them -- with the builtin hipcc drains the DMA before the next fragment read, 19.8 instead of 56.6 TFLOP/s alone;
// from an asm statement, which it does not count, the statement's memory clobber still stops the fragment reads of
// the next k-step from moving above it, 28.8 TFLOP/s.  The asm form issued in one go equals the builtin.)

__global__ __launch_bounds__(512, 1) void k_dense_g(DevSym S, int32_t dense_first, const DenseWork* __restrict__ work,
                                                    double* __restrict__ L, double* __restrict__ scratch,
                                                    const double* __restrict__ zeros) {
  static_assert(KC == 16 && NB == 128 && DTR == 256, "k_dense_g: 8 waves x (2 A k-rows x 2 halves + 2 B k-rows) per chunk");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Abuf = smem;                      // [2][KC][LDA2]
  double* Bbuf = smem + 2 * KC * LDA2;      // [2][KC][LDB]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const DenseWork wk = work[blockIdx.x];
  const int32_t j = wk.front;
  const int32_t c0j = S.sn_start[j], wj = S.sn_start[j + 1] - c0j;
  const int32_t mj = S.n - c0j;
  const int32_t R0 = wk.ti0 * TM;
  const int32_t nrow = min(wk.ntiles * TM, mj - R0);
  // per-lane source offsets (doubles) inside a panel column: row pair of each A half, column pair of B
  const int32_t a_off0 = R0 + (2 * lane < nrow ? 2 * lane : 0);
  const int32_t a_off1 = R0 + (TM + 2 * lane < nrow ? TM + 2 * lane : 0);
  const int32_t b_off = 2 * lane < wj ? 2 * lane : 0;
  const double* zsrc = zeros + 2 * lane;
  int32_t kd = wk.k0, kk0 = 0;
  int kc_ld = 0;
  auto issue_chunk = [&](int b) {
    const int32_t d = dense_first + kd;
    const int32_t c0d = S.sn_start[d], wd = S.sn_start[d + 1] - c0d;
    const int64_t md = S.n - c0d;
    const double* Pd = L + S.sn_loff[d] + (int64_t)kk0 * md + (c0j - c0d);
    kc_ld = min(KC, wd - kk0);
    double* As = Abuf + b * KC * LDA2;
    double* Bs = Bbuf + b * KC * LDB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kr = 2 * wv + i;
      const double* col = Pd + (int64_t)kr * md;
      const bool in = kr < kc_ld;
      __builtin_amdgcn_global_load_lds((gl_vptr)(in ? col + a_off0 : zsrc), (lds_vptr)(As + kr * LDA2), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gl_vptr)(in ? col + a_off1 : zsrc), (lds_vptr)(As + kr * LDA2 + TM), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kr = wv + 8 * i;
      const bool in = kr < kc_ld;
      __builtin_amdgcn_global_load_lds((gl_vptr)(in ? Pd + (int64_t)kr * md + b_off : zsrc), (lds_vptr)(Bs + kr * LDB), 16, 0, 0);
    }
    kk0 += KC;
    if (kk0 >= wd) { kk0 = 0; ++kd; }
  };
  d4 acc16[NJB][2];
#pragma unroll
  for (int a = 0; a < NJB; ++a) { acc16[a][0] = (d4){0.0, 0.0, 0.0, 0.0}; acc16[a][1] = (d4){0.0, 0.0, 0.0, 0.0}; }
  const int li = lane & 15, lk = lane >> 4;
  if (wk.k0 >= wk.k1) return;
  issue_chunk(0);
  int kc_cur = kc_ld;
  __syncthreads();  // (hipcc drains the outstanding LDS-DMA -- vmcnt(0) -- ahead of the barrier)
  int buf = 0;
  while (true) {
    const bool more = kd < wk.k1;
    if (more) issue_chunk(buf ^ 1);
    const double* Ac = Abuf + buf * KC * LDA2;
    const double* Bc = Bbuf + buf * KC * LDB;
    const int kc4 = (kc_cur + 3) & ~3;
#pragma unroll 2
    for (int k4 = 0; k4 < kc4; k4 += 4) {
      const double a0 = Ac[(k4 + lk) * LDA2 + 32 * wv + li], a1 = Ac[(k4 + lk) * LDA2 + 32 * wv + 16 + li];
      double b[NJB];
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb) b[jb] = Bc[(k4 + lk) * LDB + 16 * jb + li];
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb) {
        acc16[jb][0] = mfma_f64(b[jb], a0, acc16[jb][0]);
        acc16[jb][1] = mfma_f64(b[jb], a1, acc16[jb][1]);
      }
    }
    if (!more) break;
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
        const int i = 32 * wv + 16 * ib + li, jc = 16 * jb + lk + 4 * r;
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
// Compact path of the supernodal update: combos whose rows / columns are scattered over the target tile
// (a family subtree updating a few dozen of the 128 x 128 cells' rows and columns) would keep all eight
// waves and all column blocks of k_update busy although only ceil(nt/16) x ceil(nq/16) blocks carry data.
// Here such a combo is multiplied in ITS OWN coordinates (rows ta.. and p0.. of the descendant, staged
// contiguously), and the nt x nq result is subtracted from the panel cell by cell.  A work item owns one half
// (64 rows) of a tile and runs after the dense update and its reduce on the same stream, so it is the only
// writer of its cells; fixed order of its combos + one writer per cell per combo => bitwise reproducible (the L2
// atomics of one combo are drained before the next one starts).
constexpr int KCQ = 16;  // K depth per staging chunk of the compact path

template <bool MFMA>
__global__ __launch_bounds__(256) void k_update_compact(DevSym S, const UpdWork* __restrict__ work,
                                                        const ComboDesc* __restrict__ combos, double* __restrict__ L,
                                                        double* __restrict__ slabs) {
  // slabs == nullptr: the item owns its half tile and subtracts from the panel.  Otherwise it accumulates into
  // its private partial slab slabs[slot] (zeroed here; k_reduce folds it) and may run beside other writers.
  __shared__ __attribute__((aligned(16))) double As[KCQ * LDA];  // [k][t]  descendant rows of the tile (compact)
  __shared__ __attribute__((aligned(16))) double Bs[KCQ * LDB];  // [k][q]  descendant rows = target columns (compact)
  __shared__ int32_t rowlab[TM];
  __shared__ int32_t pos[TM];   // compact row t -> tile position
  __shared__ int32_t col[NB];   // compact column q -> target column
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const UpdWork wk = work[blockIdx.x];
  const int32_t g = wk.tile;
  const int32_t s = S.tile_front[g];
  const int32_t ti = (int32_t)(g - S.tile_base[s]);
  const int32_t c0 = S.sn_start[s];
  const int32_t* rs = S.sn_rows + S.sn_rowptr[s];
  const int32_t m = (int32_t)(S.sn_rowptr[s + 1] - S.sn_rowptr[s]);
  const int32_t R0 = ti * TM;
  const int32_t nrow = min(TM, m - R0);
  double* Q = slabs ? slabs + (int64_t)wk.slot * (TM * NB) : L + S.sn_loff[s] + R0;  // cell (i, j) at Q[j * ldq + i]
  const int64_t ldq = slabs ? TM : m;
  const double sgn = slabs ? 1.0 : -1.0;
  if (slabs)
    for (int idx = tid; idx < TM * NB; idx += 256) Q[idx] = 0.0;
  if (tid < TM) rowlab[tid] = tid < nrow ? rs[R0 + tid] : 0x7fffffff;
  __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");  // slab zeros are in L2 before the first atomic of this item
  __syncthreads();
  for (int64_t c = wk.cb; c < wk.ce; ++c) {
    const ComboDesc d = combos[c];
    if (tid < d.nt) {
      int p;
      if (d.ip0 >= 0) {
        p = d.ip0 + tid;
      } else {
        const int32_t lab = S.sn_rows[d.rowoff + d.ta + tid];
        int lo = 0, hi = nrow;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (rowlab[mid] < lab) lo = mid + 1; else hi = mid;
        }
        p = lo;
      }
      pos[tid] = p;
    }
    if (tid >= 128 && tid - 128 < d.nq) {
      const int q = tid - 128;
      col[q] = (d.jp0 >= 0) ? d.jp0 + q : S.sn_rows[d.rowoff + d.p0 + q] - c0;
    }
    const int bq = (d.nq + 15) >> 4;
    const int nblk = ((d.nt + 15) >> 4) * bq;  // <= 16 by the host's classification
    d4 acc[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) acc[a] = (d4){0.0, 0.0, 0.0, 0.0};
    const double* Pd = L + d.loff;
    const int64_t md = d.md;
    for (int k0 = 0; k0 < d.wd; k0 += KCQ) {
      const int kc = min(KCQ, d.wd - k0);
      const int kc4 = (kc + 3) & ~3;
      __syncthreads();  // the previous chunk has been consumed
      {
        // 256 threads: thread (x = tid & 127, kk = tid >> 7) stages k = kk, kk + 2, ... of row x for A and for B
        const int x = tid & 127, kk = tid >> 7;
        const bool ha = x < d.nt, hb = x < d.nq;
        const double* pa = Pd + (int64_t)(k0 + kk) * md + d.ta + (ha ? x : 0);
        const double* pb = Pd + (int64_t)(k0 + kk) * md + d.p0 + (hb ? x : 0);
        double va[KCQ / 2], vb[KCQ / 2];
#pragma unroll
        for (int i = 0; i < KCQ / 2; ++i) {
          const int k = min(kk + 2 * i, kc - 1) - kk;  // clamped into the chunk: unconditional, independent loads
          va[i] = pa[(int64_t)k * md];
          vb[i] = pb[(int64_t)k * md];
        }
#pragma unroll
        for (int i = 0; i < KCQ / 2; ++i) {
          const int k = kk + 2 * i;
          if (k < kc4) {
            As[k * LDA + x] = (ha && k < kc) ? va[i] : 0.0;
            Bs[k * LDB + x] = (hb && k < kc) ? vb[i] : 0.0;
          }
        }
      }
      __syncthreads();
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int b = wv + 4 * a;
        if (b < nblk) {
          const int ib = b / bq, jb = b - ib * bq;
          if (MFMA) {
            for (int k4 = 0; k4 < kc4; k4 += 4)
              acc[a] = mfma_f64(Bs[(k4 + lk) * LDB + 16 * jb + li], As[(k4 + lk) * LDA + 16 * ib + li], acc[a]);
          } else {
            for (int k = 0; k < kc4; ++k)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[a][r] += Bs[k * LDB + 16 * jb + lk + 4 * r] * As[k * LDA + 16 * ib + li];
          }
        }
      }
    }
    // scatter: D[M = q][N = t] -> slab(col[q], pos[t]); one writer per cell within a combo
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int b = wv + 4 * a;
      if (b < nblk) {
        const int ib = b / bq, jb = b - ib * bq;
        const int t = 16 * ib + li;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int q = 16 * jb + lk + 4 * r;
          if (t < d.nt && q < d.nq) unsafeAtomicAdd(&Q[(int64_t)col[q] * ldq + pos[t]], sgn * acc[a][r]);
        }
      }
    }
    __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this combo's adds are complete before the next combo's
    __syncthreads();                                       // (fixed summation order), and pos / col may be rewritten
  }
}

// ------------------------------------------------------------------------------------------------
// k_trsm4: k_trsm with the product on v_mfma_f64_4x4x4f64 (see k_update3 for the lane roles and why it is the faster
// form on gfx950: this launch is one 128 x 128 x 128 product per workgroup at one wave per SIMD, i.e. pure MFMA
// issue latency -- 2048 instructions of 17 cycles per wave instead of 512 of 140).  Wave wv owns target columns
// [32 wv, 32 wv + 32) x all 128 rows as 8 x 8 pieces of 16 rows x 4 columns.
__global__ __launch_bounds__(256) void k_trsm4(DevSym S, const int32_t* __restrict__ tiles, double* __restrict__ L,
                                               const double* __restrict__ invD) {
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
  double* P = L + S.sn_loff[s];
  const double* I = invD + S.inv_off[s];
  constexpr int NRB = TM / 16, NQ = 8;
  double acc4[NRB][NQ];
#pragma unroll
  for (int a = 0; a < NRB; ++a)
#pragma unroll
    for (int b = 0; b < NQ; ++b) acc4[a][b] = 0.0;
  constexpr int PA = KCS / 2, PB = KCS / (256 / NB);  // values per thread and chunk: A row t, B column q
  const int t = tid & 127, ka = tid >> 7;
  const int q = tid % NB, kb = tid / NB;
  const bool ha = t < nrow, hb = q < w;
  const double* pa = P + R0 + (ha ? t : 0);
  const double* pb = I + (hb ? q : 0);
  double ra[PA], rb[PB];
  auto fetch = [&](int k0) {
    const int kc = min(KCS, w - k0);
#pragma unroll
    for (int i = 0; i < PA; ++i) ra[i] = pa[(int64_t)(k0 + min(ka + 2 * i, kc - 1)) * m];
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
      if (k < kc4) Bs[k * LDB + q] = (hb && k < kc) ? rb[i] : 0.0;  // invL[j][k] -> Bs[k][j]
    }
  };
  const int l15 = lane & 15, l3 = lane & 3, lk4 = lane >> 4;
  const int prn = (nrow + 15) >> 4;      // 16-row groups that carry rows
  const bool wave_on = 32 * wv < w;      // this wave's 32 columns exist
  fetch(0);
  for (int32_t k0 = 0; k0 < w; k0 += KCS) {
    const int kc = min(KCS, w - k0);
    const int kc4 = (kc + 3) & ~3;
    if (k0 > 0) __syncthreads();  // the previous chunk has been consumed
    stage(k0);
    if (k0 + KCS < w) fetch(k0 + KCS);
    __syncthreads();
    if (wave_on) {
      for (int k4 = 0; k4 < kc4; k4 += 4) {
        double cv[NQ];
#pragma unroll
        for (int q4 = 0; q4 < NQ; ++q4) cv[q4] = Bs[(k4 + lk4) * LDB + 32 * wv + 4 * q4 + l3];
#pragma unroll
        for (int pr = 0; pr < NRB; ++pr)
          if (pr < prn) {
            const double rv = As[(k4 + lk4) * LDA + 16 * pr + l15];
#pragma unroll
            for (int q4 = 0; q4 < NQ; ++q4) acc4[pr][q4] = __builtin_amdgcn_mfma_f64_4x4x4f64(cv[q4], rv, acc4[pr][q4], 0, 0, 0);
          }
      }
    }
  }
#pragma unroll
  for (int pr = 0; pr < NRB; ++pr)
#pragma unroll
    for (int q4 = 0; q4 < NQ; ++q4) {
      const int j = 32 * wv + 4 * q4 + lk4;
      const int i = 16 * pr + l15;
      if (i < nrow && R0 + i >= w && j < w) P[(int64_t)j * m + R0 + i] = acc4[pr][q4];
    }
}

// ------------------------------------------------------------------------------------------------
// k_dense_a: the dense-tail update with ONLY the B operand in LDS.  A wave multiplies its own 32 rows and nobody
// else's, so its A fragments need no sharing: they are loaded straight from the panel into registers (a lane's two
// rows for its k of a k-step: 16 consecutive rows = 128 contiguous bytes per quarter wave), one 16-deep sub-chunk
// ahead of their use.  LDS then holds B alone, 64 k-rows deep per buffer (2 x 64 x 144 doubles = 144 KB) and filled by
// LDS-DMA (global_load_lds_dwordx4, gfx950: a wave-instruction moves 64 x 16 bytes from per-lane global addresses
// straight into 1 KiB of contiguous LDS = one k-row of the B image; k-rows past the end of a descendant come from a
// zero page): one barrier per 64 k, no A image to write or read, no staging registers.
// (Measured and dropped: three register sets with the A loads TWO sub-chunks ahead -- 256 VGPRs, same 61.5 / 58.6
// TFLOP/s alone: the loads cost issue and bandwidth, not latency.)
#ifndef SCILMM_DENSE_A_ABL
#define SCILMM_DENSE_A_ABL 0  // tuning harness only: 1 = A fragments loaded once per item (WRONG numbers; what do the loads cost?),
                              // 2 = B copied once per item
#endif
#ifndef SCILMM_DENSE_A_SCHED
#define SCILMM_DENSE_A_SCHED 0  // tuning harness: 0 = scheduling barrier after every k-step, 1 = after every sub-chunk, 2 = every 2 k-steps
                                // (alone, zero / random operands: 61.6 / 59.5, 61.0 / 58.6, 61.6 / 57.9 TFLOP/s -- no difference)
#endif
__global__ __launch_bounds__(512, 1) void k_dense_a(DevSym S, int32_t dense_first, const DenseWork* __restrict__ work,
                                                    double* __restrict__ L, double* __restrict__ scratch,
                                                    const double* __restrict__ zeros) {
  static_assert(NB == 128 && DTR == 256, "k_dense_a: 8 waves x 32 rows, one B k-row per DMA instruction");
  extern __shared__ __attribute__((aligned(16))) double smem[];  // [2][KBA][LDB]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const DenseWork wk = work[blockIdx.x];
  const int32_t j = wk.front;
  const int32_t c0j = S.sn_start[j], wj = S.sn_start[j + 1] - c0j;
  const int32_t mj = S.n - c0j;
  const int32_t R0 = wk.ti0 * TM;
  const int32_t nrow = min(wk.ntiles * TM, mj - R0);
  const int32_t ra0 = R0 + (32 * wv + li < nrow ? 32 * wv + li : 0);
  const int32_t ra1 = R0 + (32 * wv + 16 + li < nrow ? 32 * wv + 16 + li : 0);
  const int32_t b_off = 2 * lane < wj ? 2 * lane : 0;
  const double* zsrc = zeros + 2 * lane;
  struct Chunk { const double* Pd; int64_t md; int kc; };
  int32_t kd = wk.k0, kk0 = 0;
  bool first_a = true, first_b = true;  // (only read by the tuning-harness ablations)
  auto next_chunk = [&]() {
    const int32_t d = dense_first + kd;
    const int32_t c0d = S.sn_start[d], wd = S.sn_start[d + 1] - c0d;
    Chunk c;
    c.md = S.n - c0d;
    c.Pd = L + S.sn_loff[d] + (int64_t)kk0 * c.md + (c0j - c0d);
    c.kc = min(KBA, wd - kk0);
    kk0 += KBA;
    if (kk0 >= wd) { kk0 = 0; ++kd; }
    return c;
  };
  auto issue_B = [&](const Chunk& c, int b) {
    if (SCILMM_DENSE_A_ABL == 2 && !first_b) return;
    double* Bs = smem + b * KBA * LDB;
#pragma unroll
    for (int i = 0; i < KBA / 8; ++i) {
      const int kr = wv + 8 * i;
      __builtin_amdgcn_global_load_lds((gl_vptr)(kr < c.kc ? c.Pd + (int64_t)kr * c.md + b_off : zsrc), (lds_vptr)(Bs + kr * LDB), 16, 0, 0);
    }
  };
  auto load_A = [&](const Chunk& c, int s, double (&ra)[4][2]) {
    if (SCILMM_DENSE_A_ABL == 1 && !first_a) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double* p = c.Pd + (int64_t)min(16 * s + 4 * q + lk, c.kc - 1) * c.md;  // past the end: any valid column (B is 0 there)
      ra[q][0] = p[ra0];
      ra[q][1] = p[ra1];
    }
  };
  d4 acc16[NJB][2];
#pragma unroll
  for (int a = 0; a < NJB; ++a) { acc16[a][0] = (d4){0.0, 0.0, 0.0, 0.0}; acc16[a][1] = (d4){0.0, 0.0, 0.0, 0.0}; }
  if (wk.k0 >= wk.k1) return;
  double rA[2][4][2];
  Chunk cur = next_chunk();
  issue_B(cur, 0);
  load_A(cur, 0, rA[0]);
  if (SCILMM_DENSE_A_ABL == 1) { load_A(cur, 1, rA[1]); first_a = false; }
  first_b = false;
  __syncthreads();
  int buf = 0;
  while (true) {
    const bool more = kd < wk.k1;
    Chunk nxt = cur;
    if (more) {
      nxt = next_chunk();
      issue_B(nxt, buf ^ 1);
    }
    const double* Bc = smem + buf * KBA * LDB;
    auto kstep = [&](int k4, double a0, double a1) {
      double b[NJB];
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb) b[jb] = Bc[(k4 + lk) * LDB + 16 * jb + li];
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb) {
        acc16[jb][0] = mfma_f64(b[jb], a0, acc16[jb][0]);
        acc16[jb][1] = mfma_f64(b[jb], a1, acc16[jb][1]);
      }
#if SCILMM_DENSE_A_SCHED == 0
      __builtin_amdgcn_sched_barrier(0);  // keep the k-steps apart: unrolled 16 deep, the scheduler otherwise hoists every
                                          // fragment read to the top and spills 149 registers
#endif
    };
    {
      // sub-chunk s + 1 (or sub-chunk 0 of the next chunk) is in flight while sub-chunk s is multiplied.  A chunk shorter
      // than 64 (last chunk of a descendant whose width is no multiple of 64 -- rare) runs the same 16 k-steps: its B
      // k-rows past the end are zero, its A loads re-read the last column.  (A separate rolled loop for it cost the
      // common path its registers: 156 spills.)
#if SCILMM_DENSE_A_SCHED == 1
#define SCILMM_SUBSYNC(q) if ((q) == 3) __builtin_amdgcn_sched_barrier(0)
#elif SCILMM_DENSE_A_SCHED == 2
#define SCILMM_SUBSYNC(q) if ((q) & 1) __builtin_amdgcn_sched_barrier(0)
#else
#define SCILMM_SUBSYNC(q)
#endif
      load_A(cur, 1, rA[1]);
#pragma unroll
      for (int q = 0; q < 4; ++q) { kstep(4 * q, rA[0][q][0], rA[0][q][1]); SCILMM_SUBSYNC(q); }
      load_A(cur, 2, rA[0]);
#pragma unroll
      for (int q = 0; q < 4; ++q) { kstep(16 + 4 * q, rA[1][q][0], rA[1][q][1]); SCILMM_SUBSYNC(q); }
      load_A(cur, 3, rA[1]);
#pragma unroll
      for (int q = 0; q < 4; ++q) { kstep(32 + 4 * q, rA[0][q][0], rA[0][q][1]); SCILMM_SUBSYNC(q); }
      if (more) load_A(nxt, 0, rA[0]);
#pragma unroll
      for (int q = 0; q < 4; ++q) { kstep(48 + 4 * q, rA[1][q][0], rA[1][q][1]); SCILMM_SUBSYNC(q); }
#undef SCILMM_SUBSYNC
    }
    if (!more) break;
    cur = nxt;
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
        const int i = 32 * wv + 16 * ib + li, jc = 16 * jb + lk + 4 * r;
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
// k_dense_f (RETIRED before it shipped: 70 TFLOP/s alone against k_dense32's 83, see the note at the end): the fp32-product
// dense-tail update, round 3 -- k_dense_b's operand streams with the products on the fp32 matrix
// pipe.  On gfx950 v_mfma_f32_16x16x4_f32 runs at the fp32 VECTOR rate and measurably shares its issue time with the
// double-precision vector instructions of the same wave pair, so the kernel is organised around issuing as few of those
// as possible per k-step of 16 products (k_dense32 above: 3 conversions + 32 fold instructions per k-step, 83 TFLOP/s
// alone; the straightforward fp32 form of k_dense_b, every wave converting the B fragments it reads: 10 + 8, 84 TFLOP/s):
//  * B: the 32 k-rows of a chunk arrive as fp64 by LDS-DMA (two staging buffers, copy issued three chunks ahead) and are
//    rounded to an fp32 image ONCE per workgroup, each wave converting exactly the k-rows it copied itself (no barrier
//    between copy and conversion): 1 conversion per lane and k-step.  The fragments are then 4-byte LDS reads, prefetched
//    one k-step ahead as in k_dense_b.
//  * A: straight from the panel into registers one 16-deep sub-chunk ahead (k_dense_b's addressing), rounded when they
//    arrive: 2 conversions per k-step.
//  * sums: fp32 accumulators take SCILMM_DENSE_F_FOLD chunks (64 k by default), then are folded into the fp64 accumulators
//    that live across the item (8 instructions per k-step); the product after a fold starts from a zero C operand instead
//    of cleared registers.  The two waves of a SIMD fold in different chunks, so one of them always feeds the matrix pipe.
// Same work items, slabs and epilogue contract as k_dense_b; error model as k_dense32 with 64 instead of 16 products per
// fp32 sum (measured against the fp64 kernel on random operands: 9e-5 against 2e-5 absolute at entries of 2e3).
constexpr int KF = 32;      // k-rows per chunk
constexpr int DROW = NB;    // doubles per k-row of a staging buffer (read back only by the lane that the DMA wrote for)
#ifndef SCILMM_DENSE_F_FOLD
#define SCILMM_DENSE_F_FOLD 2
#endif
constexpr size_t dense_f_lds = sizeof(double) * 2 * KF * DROW + sizeof(float) * 2 * KF * LDBF;

__global__ __launch_bounds__(512, 1) void k_dense_f(DevSym S, int32_t dense_first, const DenseWork* __restrict__ work,
                                                    double* __restrict__ L, double* __restrict__ scratch,
                                                    const double* __restrict__ zeros) {
  static_assert(NB == 128 && DTR == 256 && (SCILMM_DENSE_F_FOLD & (SCILMM_DENSE_F_FOLD - 1)) == 0, "k_dense_f: 8 waves x 32 rows");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  double* Dbuf = smem;                              // [2][KF][DROW]  fp64 k-rows as copied
  float* Fimg = (float*)(smem + 2 * KF * DROW);     // [2][KF][LDBF]  their fp32 image
  typedef double d2 __attribute__((ext_vector_type(2)));
  typedef float f2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const DenseWork wk = work[blockIdx.x];
  const int32_t j = wk.front;
  const int32_t c0j = S.sn_start[j], wj = S.sn_start[j + 1] - c0j;
  const int32_t mj = S.n - c0j;
  const int32_t R0 = wk.ti0 * TM;
  const int32_t nrow = min(wk.ntiles * TM, mj - R0);
  const int32_t ia = 32 * wv + li, ib_ = ia + 16;
  const int32_t ra0 = R0 + (ia < nrow ? ia : 0);
  const int32_t ra1 = R0 + (ib_ < nrow ? ib_ : 0);
  const int32_t b_off = 2 * lane < wj ? 2 * lane : 0;
  const double* zsrc = zeros + 2 * lane;
  if (wk.k0 >= wk.k1) return;
  struct Chunk { const double* Pd; int32_t md; int kc; };
  struct Iter { int32_t kd, kk0; };
  auto next_chunk = [&](Iter& it) {
    const int32_t d = dense_first + it.kd;
    const int32_t c0d = S.sn_start[d], wd = S.sn_start[d + 1] - c0d;
    Chunk c;
    c.md = __builtin_amdgcn_readfirstlane(S.n - c0d);
    c.Pd = L + uniform_i64(S.sn_loff[d] + (int64_t)it.kk0 * c.md + (c0j - c0d));
    c.kc = __builtin_amdgcn_readfirstlane(min(KF, wd - it.kk0));
    it.kk0 += KF;
    if (it.kk0 >= wd) { it.kk0 = 0; ++it.kd; }
    return c;
  };
  Iter it_c{wk.k0, 0}, it_d{wk.k0, 0};  // the same chunk sequence twice: multiplied / copied (three chunks ahead)
  auto dma = [&](int b) {
    if (it_d.kd >= wk.k1) return;
    const Chunk c = next_chunk(it_d);
    double* Db = Dbuf + b * KF * DROW;
#pragma unroll
    for (int i = 0; i < KF / 8; ++i) {
      const int kr = wv + 8 * i;
      __builtin_amdgcn_global_load_lds((gl_vptr)(kr < c.kc ? c.Pd + (int64_t)kr * c.md + b_off : zsrc), (lds_vptr)(Db + kr * DROW), 16, 0, 0);
    }
  };
  auto convert = [&](int b) {  // this wave's k-rows of staging buffer b -> image b
    const double* Db = Dbuf + b * KF * DROW;
    float* Fi = Fimg + b * KF * LDBF;
#pragma unroll
    for (int h = 0; h < KF / 16; ++h) {  // (two k-rows at a time: the registers are all but full)
      d2 v[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) v[i] = *(const d2*)(Db + (wv + 8 * (2 * h + i)) * DROW + 2 * lane);
#pragma unroll
      for (int i = 0; i < 2; ++i) *(f2*)(Fi + (wv + 8 * (2 * h + i)) * LDBF + 2 * lane) = (f2){(float)v[i][0], (float)v[i][1]};
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  double rA[4][2];
  float raf[4][2];
  auto load_A = [&](const Chunk& c, int s) {
    const int klast = (c.kc - 1) & ~3;
    const uint32_t v0 = (uint32_t)(lk * c.md + ra0) * 8u, v1 = (uint32_t)(lk * c.md + ra1) * 8u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double* sp = c.Pd + (int64_t)min(16 * s + 4 * q, klast) * c.md;  // wave-uniform
      rA[q][0] = ld_off(sp, v0);
      rA[q][1] = ld_off(sp, v1);
    }
  };
  d4 acc[NJB][2];
  f4 c32[NJB][2];
#pragma unroll
  for (int a = 0; a < NJB; ++a) {
    acc[a][0] = (d4){0.0, 0.0, 0.0, 0.0}; acc[a][1] = (d4){0.0, 0.0, 0.0, 0.0};
    c32[a][0] = (f4){0.f, 0.f, 0.f, 0.f}; c32[a][1] = (f4){0.f, 0.f, 0.f, 0.f};
  }
  float bf[2][NJB];
  auto ldB = [&](const float* Fc, int k4, float (&b)[NJB]) {
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) b[jb] = Fc[(k4 + lk) * LDBF + 16 * jb + li];
  };
  auto fold = [&]() {
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int ib = 0; ib < 2; ++ib)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[jb][ib][r] += (double)c32[jb][ib][r];
  };
  const int fold_phase = (wv >> 2) & (SCILMM_DENSE_F_FOLD - 1);
  Chunk cur = next_chunk(it_c);
  dma(0);
  dma(1);
  load_A(cur, 0);
  bool more = it_c.kd < wk.k1;
  Chunk nxt = cur;
  if (more) nxt = next_chunk(it_c);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  convert(0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  dma(0);
  __syncthreads();
  ldB(Fimg, 0, bf[0]);
  int cidx = 0;
  while (true) {
    const float* Fc = Fimg + (cidx & 1) * KF * LDBF;
    const float* Fn = Fimg + ((cidx + 1) & 1) * KF * LDBF;
    bool more2 = false;
    Chunk nn = nxt;
    const bool do_fold = (cidx & (SCILMM_DENSE_F_FOLD - 1)) == fold_phase;
#pragma unroll
    for (int t = 0; t < KF / 4; ++t) {
      const int s = t >> 2, q = t & 3;
      if (q == 0) {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) { raf[qq][0] = (float)rA[qq][0]; raf[qq][1] = (float)rA[qq][1]; }
        if (s == 1 && more) {
          // the next chunk's fp32 image (its copy was issued two chunks ago), then the copy of the chunk three ahead into
          // the staging buffer just read
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          convert((cidx + 1) & 1);
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          dma((cidx + 1) & 1);
        }
        if (s == 0) load_A(cur, 1);
        else if (more) load_A(nxt, 0);
      }
      if (t < KF / 4 - 1) {
        ldB(Fc, 4 * (t + 1), bf[(t + 1) & 1]);
      } else if (more) {
        // chunk boundary: every wave has read its last fragments of this image and written its rows of the next one
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ldB(Fn, 0, bf[0]);
        more2 = it_c.kd < wk.k1;
        if (more2) nn = next_chunk(it_c);
      }
      if (t == 0 && do_fold) {
        fold();
#pragma unroll
        for (int jb = 0; jb < NJB; ++jb) { c32[jb][0] = (f4){0.f, 0.f, 0.f, 0.f}; c32[jb][1] = (f4){0.f, 0.f, 0.f, 0.f}; }
      }
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb) {
        c32[jb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[t & 1][jb], raf[q][0], c32[jb][0], 0, 0, 0);
        c32[jb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[t & 1][jb], raf[q][1], c32[jb][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (!more) break;
    cur = nxt;
    nxt = nn;
    more = more2;
    ++cidx;
  }
  fold();
  double* P = L + S.sn_loff[j];
#pragma unroll
  for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ib == 0 ? ia : ib_, jc = 16 * jb + 4 * lk + r;  // fp32 MFMA layout: M = 4 (l >> 4) + r
        const double v = acc[jb][ib][r];
        const int h = i >> 7;
        const int32_t slot = h ? wk.slot1 : wk.slot0;
        if (slot < 0) {
          if (i < nrow && jc < wj) P[(int64_t)jc * mj + R0 + i] -= v;
        } else if (h < wk.ntiles) {
          scratch[(int64_t)slot * (TM * NB) + jc * TM + (i & (TM - 1))] = v;
        }
      }
}

// Why k_dense_f is here: profiles/r3_mfma_f32_mix.txt (csrc/tools/mfma_f32_mix.hip) shows that the instruction mix is not
// what holds the fp32-product kernels at half of the fp32 matrix rate (16 MFMAs + LDS fragments + a fold every 16 k-steps:
// 135 of 155 TFLOP/s; a fold every 4 k-steps, k_dense32's period: 107), so removing conversions and folds bought nothing,
// and the shorter chunks (a barrier per 32 k) cost 15 %.  (The explanation that followed at the time -- the fp64 operand
// stream -- was tested by k_dense_s / k_dense_t below and did not hold; the cause is the register pressure of the fp64 + fp32
// sums of a 32 x 128 wave tile: DESIGN.md section 4.1, k_dense_h in csrc/kernels.hip.h.)
// that stream, not by the matrix pipe.

// ------------------------------------------------------------------------------------------------
// k_dense_s (RETIRED before it shipped: 86 TFLOP/s alone against k_dense32's 83 for + 50 % tail storage, see the note at the end):
// the fp32-product dense-tail update reading an fp32 SHADOW of the tail panels (round 3).  k_dense32 above rounds
// its fp64 operands while staging them and is bound by that operand stream: a 256 x 128 tile reads 32 flop per HBM byte, 2.6
// TB/s at its 83 TFLOP/s (the instruction mix alone allows 107 - 135).  With front precision 32 the
// engine therefore keeps a second, fp32 copy of every finished tail panel (k_shadow, one pass right after the panel's
// k_trsm: + 50 % tail storage) and this kernel reads ONLY that copy: half the bytes per flop.  Same values as k_dense32's
// operands (one rounding of the finished fp64 entry), the products on the fp32 matrix pipe, folded into fp64 accumulators
// every 64 k; the subtraction from the fp64 panel, k_potrf, k_trsm and the solves stay fp64.  Structure = k_dense_b's:
//  * A: a lane owns two ADJACENT target rows (32 wv + 2 li, + 1) and loads both with one 8-byte load per k (a quarter wave
//    reads 128 contiguous bytes), straight into registers one 16-deep sub-chunk ahead;
//  * B: 64 k-rows per LDS buffer by LDS-DMA, 4 bytes per lane (global_load_lds_dword: 64 lanes x 4 B = half a k-row per
//    instruction -- the shadow's rows are only 4-byte aligned), two buffers, copy issued two chunks ahead; fragments are
//    4-byte LDS reads prefetched one k-step ahead (row stride LDBF = 144 floats: conflict-free);
//  * one barrier per 64 k; the fold follows the chunk's last product.
// Same work items, slabs and epilogue contract as k_dense_b / k_dense32.

constexpr size_t dense_s_lds = sizeof(float) * 2 * KBA * LDBF;

__global__ __launch_bounds__(512, 1) void k_dense_s(DevSym S, int32_t dense_first, const DenseWork* __restrict__ work,
                                                    double* __restrict__ L, const float* __restrict__ L32, int64_t base32,
                                                    double* __restrict__ scratch, const float* __restrict__ zeros32) {
  static_assert(NB == 128 && DTR == 256, "k_dense_s: 8 waves x 32 rows");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  float* Bimg = (float*)smem;  // [2][KBA][LDBF]
  typedef float f2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const DenseWork wk = work[blockIdx.x];
  const int32_t j = wk.front;
  const int32_t c0j = S.sn_start[j], wj = S.sn_start[j + 1] - c0j;
  const int32_t mj = S.n - c0j;
  const int32_t R0 = wk.ti0 * TM;
  const int32_t nrow = min(wk.ntiles * TM, mj - R0);
  const int32_t ia = 32 * wv + 2 * li;                    // this lane's rows inside the item: ia, ia + 1
  const int32_t ra = R0 + (ia < nrow ? ia : 0);           // (rows past the item's edge: the pair 0, 1, never stored; a pair that
                                                          //  straddles the edge reads one entry past it -- the next column's, or the
                                                          //  shadow's slack behind the last panel -- into a row that is never stored)
  const int32_t bcol0 = lane < wj ? lane : 0, bcol1 = 64 + lane < wj ? 64 + lane : 0;
  if (wk.k0 >= wk.k1) return;
  struct Chunk { const float* Pd; int32_t md; int kc; };
  int32_t kd = wk.k0, kk0 = 0;
  auto next_chunk = [&]() {
    const int32_t d = dense_first + kd;
    const int32_t c0d = S.sn_start[d], wd = S.sn_start[d + 1] - c0d;
    Chunk c;
    c.md = __builtin_amdgcn_readfirstlane(S.n - c0d);
    c.Pd = L32 + uniform_i64(S.sn_loff[d] - base32 + (int64_t)kk0 * c.md + (c0j - c0d));
    c.kc = __builtin_amdgcn_readfirstlane(min(KBA, wd - kk0));
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
  d4 acc[NJB][2];
  f4 c32[NJB][2];
#pragma unroll
  for (int a = 0; a < NJB; ++a) {
    acc[a][0] = (d4){0.0, 0.0, 0.0, 0.0}; acc[a][1] = (d4){0.0, 0.0, 0.0, 0.0};
    c32[a][0] = (f4){0.f, 0.f, 0.f, 0.f}; c32[a][1] = (f4){0.f, 0.f, 0.f, 0.f};
  }
  f2 rA[2][4];
  float bf[2][NJB];
  auto ldB = [&](const float* Bc, int k4, float (&b)[NJB]) {
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) b[jb] = Bc[(k4 + lk) * LDBF + 16 * jb + li];
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
      for (int jb = 0; jb < NJB; ++jb) {
        c32[jb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[t & 1][jb], a[0], c32[jb][0], 0, 0, 0);
        c32[jb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[t & 1][jb], a[1], c32[jb][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#ifndef SCILMM_DENSE_S_NOFOLD
    // fold the chunk's fp32 sums (64 products each) into the fp64 accumulators
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[jb][ib][r] += (double)c32[jb][ib][r];
        c32[jb][ib] = (f4){0.f, 0.f, 0.f, 0.f};
      }
#endif
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
        const int i = ia + ib, jc = 16 * jb + 4 * lk + r;  // fp32 MFMA layout: M = 4 (l >> 4) + r
#ifdef SCILMM_DENSE_S_NOFOLD
        const double v = (double)c32[jb][ib][r];  // (register-pressure experiment: fp32 sums over the whole item)
#else
        const double v = acc[jb][ib][r];
#endif
        const int h = i >> 7;
        const int32_t slot = h ? wk.slot1 : wk.slot0;
        if (slot < 0) {
          if (i < nrow && jc < wj) P[(int64_t)jc * mj + R0 + i] -= v;
        } else if (h < wk.ntiles) {
          scratch[(int64_t)slot * (TM * NB) + jc * TM + (i & (TM - 1))] = v;
        }
      }
}

// Why k_dense_s is here: it tests the explanation given for k_dense_f above.  With operands that are ALREADY fp32 in HBM (half
// the bytes per flop, no conversions at all) it reaches 86.3 TFLOP/s on the 1M-shaped launch against 83.2 (k_dense32) and 84.5
// (k_dense_b's streams with fp32 products); with the A fragments fetched two sub-chunks ahead and a partial vmcnt wait at the
// chunk barrier: 78 (more spills).  So neither the operand bytes nor the conversions nor the load latency explain the 0.55 of
// the fp32 pipe all four forms share.  What does was found two kernels later: this very kernel WITHOUT its fp64 accumulators
// (-DSCILMM_DENSE_S_NOFOLD: 134 registers, no spills) runs at 118.5 -- the sums of a 32 x 128 wave tile do not fit two waves per SIMD.
// k_dense_h (csrc/kernels.hip.h) keeps the shadow and halves the wave's columns.

// ------------------------------------------------------------------------------------------------
// k_dense_t (RETIRED before it shipped: 87 - 88 TFLOP/s alone, as the other two-waves-per-SIMD forms; the note at the end):
// the fp32-product dense-tail update on a TILE-MAJOR fp32 shadow of the tail panels (round 3).  The fp32-product
// kernels are bound by the DRAM efficiency of their A stream (csrc/tools/mfma_f32_feed): out of a
// column-major panel a workgroup reads one 1 KB piece of 256 rows per k-column, strided by the leading dimension, and the same
// loop runs at twice the rate on long runs.  With front precision 32 the engine therefore keeps an fp32 copy of every finished
// tail panel in 128-ROW BLOCKS (k_shadow_t, one pass right after the panel's k_trsm): block b of panel d holds the panel's rows
// with tail index g = row - c0_tail in [128 b, 128 b + 128) as [k][128 rows], consecutive k adjacent -- a wave's 32 rows of one
// descendant are ONE contiguous run (128 B per k, 16 KB per 128-wide descendant block), whatever the other workgroups do.
//   shadow(d)[((g >> 7) - (c0_d - c0_tail >> 7)) * w_d + k) * 128 + (g & 127)] = (float) L_d[row g, column k]
// Operand values, products, folds and the epilogue are k_dense32's (one rounding of the finished fp64 entry; fp32 MFMA, fp64 sums
// every 64 k; fp64 subtraction); loop structure = k_dense_b's (A straight into registers one sub-chunk ahead, B image by LDS-DMA
// 4 bytes per lane, one barrier per 64 k).
__global__ __launch_bounds__(256) void k_shadow_t(const double* __restrict__ P, float* __restrict__ dst, int32_t g0, int32_t md, int32_t w) {
  // one panel: element (r, k) -> block-major; threads run along r (coalesced reads and writes)
  const int64_t cnt = (int64_t)md * w;
  const int32_t b0 = g0 >> 7;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < cnt; i += (int64_t)gridDim.x * 256) {
    const int32_t k = (int32_t)(i / md), r = (int32_t)(i - (int64_t)k * md);
    const int32_t g = g0 + r;
    dst[((int64_t)((g >> 7) - b0) * w + k) * 128 + (g & 127)] = (float)P[i];
  }
}

__global__ __launch_bounds__(512, 1) void k_dense_t(DevSym S, int32_t dense_first, const DenseWork* __restrict__ work,
                                                    double* __restrict__ L, const float* __restrict__ S32,
                                                    const int64_t* __restrict__ soff, double* __restrict__ scratch,
                                                    const float* __restrict__ zeros32) {
  static_assert(NB == 128 && DTR == 256, "k_dense_t: 8 waves x 32 rows");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  float* Bimg = (float*)smem;  // [2][KBA][LDBF]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const DenseWork wk = work[blockIdx.x];
  const int32_t j = wk.front;
  const int32_t c0t = S.sn_start[dense_first];
  const int32_t c0j = S.sn_start[j], wj = S.sn_start[j + 1] - c0j;
  const int32_t mj = S.n - c0j;
  const int32_t R0 = wk.ti0 * TM;
  const int32_t nrow = min(wk.ntiles * TM, mj - R0);
  const int32_t ia = 32 * wv + li, ib_ = ia + 16;  // this lane's rows inside the item (k_dense_b's assignment)
  // tail indices g of the lane's two A rows and of its two B columns (rows c0j + col of the descendant panels)
  const int32_t gA0 = c0j - c0t + R0 + (ia < nrow ? ia : 0), gA1 = c0j - c0t + R0 + (ib_ < nrow ? ib_ : 0);
  const int32_t gB0 = c0j - c0t + (lane < wj ? lane : 0), gB1 = c0j - c0t + (64 + lane < wj ? 64 + lane : 0);
  if (wk.k0 >= wk.k1) return;
  // The item's descendants' entries of the symbolic arrays go to LDS ONCE, up front: a chunk descriptor built from global
  // loads at the chunk boundary stalls every wave of the workgroup for a memory latency right before the last k-step's MFMAs
  // (csrc/tools/mfma_f32_feed: that alone takes the loop from 125 to 73 TFLOP/s)
  constexpr int DMAXD = 256;  // descendants per item the LDS table holds (the plan's items have <= 64)
  int32_t* t_c0 = (int32_t*)(Bimg + 2 * KBA * LDBF);   // [DMAXD + 1]
  int64_t* t_so = (int64_t*)(t_c0 + DMAXD + 2);         // [DMAXD]
  const int32_t ndesc = min(wk.k1 - wk.k0, DMAXD);
  for (int i = tid; i <= ndesc; i += 512) t_c0[i] = S.sn_start[dense_first + wk.k0 + i];
  for (int i = tid; i < ndesc; i += 512) t_so[i] = soff[wk.k0 + i];
  __syncthreads();
  struct Chunk { const float* base; int32_t kk; int kc; uint32_t vA0, vA1; int32_t vB0, vB1; };
  int32_t kd = wk.k0, kk0 = 0;
  auto next_chunk = [&]() {
    const int32_t e = kd - wk.k0;
    const bool in_table = e < DMAXD;
    const int32_t d = dense_first + kd;
    const int32_t c0d = in_table ? t_c0[e] : S.sn_start[d];
    const int32_t wd = __builtin_amdgcn_readfirstlane((in_table ? t_c0[e + 1] : S.sn_start[d + 1]) - c0d);
    const int32_t b0d = __builtin_amdgcn_readfirstlane((c0d - c0t) >> 7);
    Chunk c;
    c.base = S32 + uniform_i64(in_table ? t_so[e] : soff[d - dense_first]);
    c.kk = kk0;
    c.kc = min(KBA, wd - kk0);
    // float offsets of this lane's rows / columns in k-column 0 of the descendant's shadow (+ the lane's k within a 4-deep block)
    c.vA0 = (uint32_t)((((gA0 >> 7) - b0d) * wd + lk) * 128 + (gA0 & 127)) * 4u;
    c.vA1 = (uint32_t)((((gA1 >> 7) - b0d) * wd + lk) * 128 + (gA1 & 127)) * 4u;
    c.vB0 = (((gB0 >> 7) - b0d) * wd) * 128 + (gB0 & 127);
    c.vB1 = (((gB1 >> 7) - b0d) * wd) * 128 + (gB1 & 127);
    kk0 += KBA;
    if (kk0 >= wd) { kk0 = 0; ++kd; }
    return c;
  };
  auto issue_B = [&](const Chunk& c, int b) {
    float* Bs = Bimg + b * KBA * LDBF;
#pragma unroll
    for (int i = 0; i < KBA / 8; ++i) {
      const int kr = wv + 8 * i;
      const float* col = c.base + (int64_t)(c.kk + kr) * 128;
      const bool on = kr < c.kc;
      __builtin_amdgcn_global_load_lds((gl_vptr)(on ? col + c.vB0 : zeros32 + lane), (lds_vptr)(Bs + kr * LDBF), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((gl_vptr)(on ? col + c.vB1 : zeros32 + lane), (lds_vptr)(Bs + kr * LDBF + 64), 4, 0, 0);
    }
  };
  auto load_A = [&](const Chunk& c, int s, float (&a)[4][2]) {
    const int klast = (c.kc - 1) & ~3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const char* sp = (const char*)(c.base + (int64_t)(c.kk + min(16 * s + 4 * q, klast)) * 128);  // wave-uniform
      a[q][0] = *(const float*)(sp + c.vA0);
      a[q][1] = *(const float*)(sp + c.vA1);
    }
  };
  d4 acc[NJB][2];
  f4 c32[NJB][2];
#pragma unroll
  for (int a = 0; a < NJB; ++a) {
    acc[a][0] = (d4){0.0, 0.0, 0.0, 0.0}; acc[a][1] = (d4){0.0, 0.0, 0.0, 0.0};
    c32[a][0] = (f4){0.f, 0.f, 0.f, 0.f}; c32[a][1] = (f4){0.f, 0.f, 0.f, 0.f};
  }
  float rA[2][4][2];
  float bf[2][NJB];
  auto ldB = [&](const float* Bc, int k4, float (&b)[NJB]) {
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) b[jb] = Bc[(k4 + lk) * LDBF + 16 * jb + li];
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
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ldB(Bn, 0, bf[0]);
        more2 = kd < wk.k1;
        if (more2) {
          nn = next_chunk();
          issue_B(nn, buf);
        }
      }
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb) {
        c32[jb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[t & 1][jb], rA[s & 1][q][0], c32[jb][0], 0, 0, 0);
        c32[jb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[t & 1][jb], rA[s & 1][q][1], c32[jb][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    // fold the chunk's fp32 sums (64 products each) into the fp64 accumulators
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[jb][ib][r] += (double)c32[jb][ib][r];
        c32[jb][ib] = (f4){0.f, 0.f, 0.f, 0.f};
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
        const int i = ib == 0 ? ia : ib_, jc = 16 * jb + 4 * lk + r;  // fp32 MFMA layout: M = 4 (l >> 4) + r
        const double v = acc[jb][ib][r];
        const int h = i >> 7;
        const int32_t slot = h ? wk.slot1 : wk.slot0;
        if (slot < 0) {
          if (i < nrow && jc < wj) P[(int64_t)jc * mj + R0 + i] -= v;
        } else if (h < wk.ntiles) {
          scratch[(int64_t)slot * (TM * NB) + jc * TM + (i & (TM - 1))] = v;
        }
      }
}

// Why k_dense_t is here: it tested the DRAM-efficiency explanation (contiguous runs instead of 1 KB column pieces): 87.0 TFLOP/s,
// 88.4 with the descriptor table in LDS -- again the same figure.  A synthetic loop on a launch of the real shape (column-major fp32
// operands!) sustains 120; the difference was the real kernels' register spills (k_dense_w in csrc/kernels.hip.h).

// ------------------------------------------------------------------------------------------------
// k_dense_w (RETIRED before it shipped: 69 TFLOP/s -- the compiler spills 59 registers even with the whole file for one wave):
// the fp32-product dense-tail update with ONE wave per SIMD, a wave owns 64 rows x 128 columns.
// The four earlier forms (k_dense32 above; three rewrites in csrc/tools/retired_kernels.hip.h) all run at 83 - 88 TFLOP/s, 0.55 of
// the fp32 matrix pipe, whatever their operand format -- while a synthetic loop with the same operand feeds sustains 114 - 124 on
// a launch of the real shape (csrc/tools/mfma_f32_feed, mfma_f32_wave64; profiles/r3_mfma_f32_feed.txt).  What the real kernels
// had and the synthetic loop had not: SPILLS.  Two waves per SIMD leave 256 registers per wave; 128 (fp64 sums) + 64 (fp32 sums)
// + fragments + addressing do not fit, and the spilled accumulators come back from scratch -- through the same memory pipeline
// as the 1 - 2 TB/s operand stream -- exactly where the chunk's fold needs them.  With 256-thread workgroups a wave has the
// whole 512-register file: 256 fp64 + 128 fp32 accumulators (64 x 128 per wave), both fragment sets, no spill; every B fragment
// feeds four MFMAs instead of two.
// Operands come from an fp32 SHADOW of the finished tail panels (k_shadow right after a panel's k_trsm; same column-major
// layout, + 50 % tail storage): half the bytes of the fp64 panels per flop, no conversion in the loop.  Values = k_dense32's
// operands (one rounding of the finished fp64 entry); products on the fp32 matrix pipe; sums folded into fp64 every 64 k; the
// subtraction from the fp64 panel, k_potrf, k_trsm and the solves stay fp64.  Loop = k_dense_b's: A fragments straight into
// registers one 16-deep sub-chunk ahead, B image (64 k-rows) by LDS-DMA 4 bytes per lane, fragments prefetched one k-step ahead,
// one barrier per 64 k; the item's descendants' symbolic entries sit in an LDS table.  Same work items, slabs and epilogue
// contract as k_dense_b / k_dense32.

constexpr int DW_MAXD = 256;  // descendants per item held in the LDS table (the plan's items have <= 64)
constexpr size_t dense_w_lds = sizeof(float) * 2 * KBA * LDBF + sizeof(int32_t) * (DW_MAXD + 2) + sizeof(int64_t) * DW_MAXD;

__global__ __launch_bounds__(256, 1) void k_dense_w(DevSym S, int32_t dense_first, const DenseWork* __restrict__ work,
                                                    double* __restrict__ L, const float* __restrict__ L32, int64_t base32,
                                                    double* __restrict__ scratch, const float* __restrict__ zeros32) {
  static_assert(NB == 128 && DTR == 256, "k_dense_w: 4 waves x 64 rows");
  extern __shared__ __attribute__((aligned(16))) double smem[];
  float* Bimg = (float*)smem;                            // [2][KBA][LDBF]
  int32_t* t_c0 = (int32_t*)(Bimg + 2 * KBA * LDBF);     // [DW_MAXD + 2]  first columns of the item's descendants
  int64_t* t_lo = (int64_t*)(t_c0 + DW_MAXD + 2);        // [DW_MAXD]      their panel offsets
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 15, lk = lane >> 4;
  const DenseWork wk = work[blockIdx.x];
  const int32_t j = wk.front;
  const int32_t c0j = S.sn_start[j], wj = S.sn_start[j + 1] - c0j;
  const int32_t mj = S.n - c0j;
  const int32_t R0 = wk.ti0 * TM;
  const int32_t nrow = min(wk.ntiles * TM, mj - R0);
  if (wk.k0 >= wk.k1) return;
  const int32_t ndesc = min(wk.k1 - wk.k0, DW_MAXD);
  for (int i = tid; i <= ndesc; i += 256) t_c0[i] = S.sn_start[dense_first + wk.k0 + i];
  for (int i = tid; i < ndesc; i += 256) t_lo[i] = S.sn_loff[dense_first + wk.k0 + i];
  // this lane's four rows inside the item: 64 wv + 16 rb + li (rows past the item's edge: row 0, never stored)
  uint32_t rowb[4];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb) {
    const int32_t i = 64 * wv + 16 * rb + li;
    rowb[rb] = (uint32_t)(R0 + (i < nrow ? i : 0)) * 4u;
  }
  const int32_t bcol0 = lane < wj ? lane : 0, bcol1 = 64 + lane < wj ? 64 + lane : 0;
  __syncthreads();
  struct Chunk { const float* Pd; int32_t md; int kc; };
  int32_t kd = wk.k0, kk0 = 0;
  auto next_chunk = [&]() {
    const int32_t e = kd - wk.k0, d = dense_first + kd;
    const bool tab = e < DW_MAXD;
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
    for (int i = 0; i < KBA / 4; ++i) {
      const int kr = wv + 4 * i;
      const float* row = c.Pd + (int64_t)kr * c.md;
      const bool on = kr < c.kc;
      __builtin_amdgcn_global_load_lds((gl_vptr)(on ? row + bcol0 : zeros32 + lane), (lds_vptr)(Bs + kr * LDBF), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((gl_vptr)(on ? row + bcol1 : zeros32 + lane), (lds_vptr)(Bs + kr * LDBF + 64), 4, 0, 0);
    }
  };
  auto load_A = [&](const Chunk& c, int s, float (&a)[4][4]) {
    const int klast = (c.kc - 1) & ~3;
    const uint32_t vk = (uint32_t)(lk * c.md) * 4u;  // this lane's k-column inside a 4-deep block
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const char* sp = (const char*)(c.Pd + (int64_t)min(16 * s + 4 * q, klast) * c.md);  // wave-uniform
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) a[q][rb] = *(const float*)(sp + (vk + rowb[rb]));
    }
  };
  d4 acc[NJB][4];
  f4 c32[NJB][4];
#pragma unroll
  for (int a = 0; a < NJB; ++a)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) { acc[a][rb] = (d4){0.0, 0.0, 0.0, 0.0}; c32[a][rb] = (f4){0.f, 0.f, 0.f, 0.f}; }
  float rA[2][4][4];
  float bf[2][NJB];
  auto ldB = [&](const float* Bc, int k4, float (&b)[NJB]) {
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb) b[jb] = Bc[(k4 + lk) * LDBF + 16 * jb + li];
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
#pragma unroll
      for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
          c32[jb][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[t & 1][jb], rA[s & 1][q][rb], c32[jb][rb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // fold the chunk's fp32 sums (64 products each) into the fp64 accumulators
#pragma unroll
    for (int jb = 0; jb < NJB; ++jb)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[jb][rb][r] += (double)c32[jb][rb][r];
        c32[jb][rb] = (f4){0.f, 0.f, 0.f, 0.f};
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
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 64 * wv + 16 * rb + li, jc = 16 * jb + 4 * lk + r;  // fp32 MFMA layout: M = 4 (l >> 4) + r
        const double v = acc[jb][rb][r];
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
// k_dense_q (RETIRED before it shipped: 63.6 TFLOP/s against k_dense_b's 68.0 on the 1M-shaped launch, bit-identical slabs --
// the fp64 kernel is not short of latency hiding, so smaller wave tiles only add fragment traffic and barriers):
// k_dense_h's tiling for the fp64 kernel: a wave owns
// 32 rows x 64 columns (64 registers of sums instead of k_dense_b's 128), B chunks of 32 k-rows (two 36 KB buffers), so that two
// workgroups share a CU.  Two workgroups serve one work item.  Same operands, same products as k_dense_b; the sum ORDER inside an
// item is the same too (k ascending), so the slabs are bit-identical.
constexpr int KQ = 32;
constexpr size_t dense_q_lds = sizeof(double) * 2 * KQ * LDB + sizeof(int32_t) * (DH_MAXD + 2) + sizeof(int64_t) * DH_MAXD;

__global__ __launch_bounds__(512, 4) void k_dense_q(DevSym S, int32_t dense_first, const DenseWork* __restrict__ work,
                                                    double* __restrict__ L, double* __restrict__ scratch,
                                                    const double* __restrict__ zeros) {
  static_assert(NB == 128 && TM == 128, "k_dense_q: 4 x 2 waves of 32 rows x 64 columns");
  extern __shared__ __attribute__((aligned(16))) double smem[];  // [2][KQ][LDB]
  int32_t* t_c0 = (int32_t*)(smem + 2 * KQ * LDB);
  int64_t* t_lo = (int64_t*)(t_c0 + DH_MAXD + 2);
  constexpr int NJH = NJB / 2;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wv >> 1, ch = wv & 1;
  const int li = lane & 15, lk = lane >> 4;
  const DenseWork wk = work[blockIdx.x >> 1];
  const int th = blockIdx.x & 1;
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
  const int32_t ia = 32 * rg + li, ib_ = ia + 16;
  const int32_t ra0 = R0 + (ia < nrow ? ia : 0), ra1 = R0 + (ib_ < nrow ? ib_ : 0);
  const int32_t b_off = 2 * lane < wj ? 2 * lane : 0;
  const double* zsrc = zeros + 2 * lane;
  __syncthreads();
  struct Chunk { const double* Pd; int32_t md; int kc; };
  int32_t kd = wk.k0, kk0 = 0;
  auto next_chunk = [&]() {
    const int32_t e = kd - wk.k0, d = dense_first + kd;
    const bool tab = e < DH_MAXD;
    const int32_t c0d = __builtin_amdgcn_readfirstlane(tab ? t_c0[e] : S.sn_start[d]);
    const int32_t wd = __builtin_amdgcn_readfirstlane(tab ? t_c0[e + 1] : S.sn_start[d + 1]) - c0d;
    Chunk c;
    c.md = S.n - c0d;
    c.Pd = L + (uniform_i64(tab ? t_lo[e] : S.sn_loff[d]) + (int64_t)kk0 * c.md + (c0j - c0d));
    c.kc = min(KQ, wd - kk0);
    kk0 += KQ;
    if (kk0 >= wd) { kk0 = 0; ++kd; }
    return c;
  };
  auto issue_B = [&](const Chunk& c, int b) {
    double* Bs = smem + b * KQ * LDB;
#pragma unroll
    for (int i = 0; i < KQ / 8; ++i) {
      const int kr = wv + 8 * i;
      __builtin_amdgcn_global_load_lds((gl_vptr)(kr < c.kc ? c.Pd + (int64_t)kr * c.md + b_off : zsrc), (lds_vptr)(Bs + kr * LDB), 16, 0, 0);
    }
  };
  auto load_A = [&](const Chunk& c, int s, double (&a)[4][2]) {
    const int klast = (c.kc - 1) & ~3;
    const uint32_t v0 = (uint32_t)(lk * c.md + ra0) * 8u, v1 = (uint32_t)(lk * c.md + ra1) * 8u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double* sp = c.Pd + (int64_t)min(16 * s + 4 * q, klast) * c.md;  // wave-uniform
      a[q][0] = ld_off(sp, v0);
      a[q][1] = ld_off(sp, v1);
    }
  };
  d4 acc[NJH][2];
#pragma unroll
  for (int a = 0; a < NJH; ++a) { acc[a][0] = (d4){0.0, 0.0, 0.0, 0.0}; acc[a][1] = (d4){0.0, 0.0, 0.0, 0.0}; }
  double rA[2][4][2];
  double bf[2][NJH];
  auto ldB = [&](const double* Bc, int k4, double (&b)[NJH]) {
#pragma unroll
    for (int jb = 0; jb < NJH; ++jb) b[jb] = Bc[(k4 + lk) * LDB + 64 * ch + 16 * jb + li];
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
    const double* Bc = smem + buf * KQ * LDB;
    const double* Bn = smem + (buf ^ 1) * KQ * LDB;
    bool more2 = false;
    Chunk nn = nxt;
#pragma unroll
    for (int t = 0; t < KQ / 4; ++t) {
      const int s = t >> 2, q = t & 3;
      if (q == 0) {
        if (s == 0) load_A(cur, 1, rA[1]);
        else if (more) load_A(nxt, 0, rA[0]);
      }
      if (t < KQ / 4 - 1) {
        ldB(Bc, 4 * (t + 1), bf[(t + 1) & 1]);
      } else if (more) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        ldB(Bn, 0, bf[0]);
        more2 = kd < wk.k1;
        if (more2) {
          nn = next_chunk();
          issue_B(nn, buf);
        }
      }
#pragma unroll
      for (int jb = 0; jb < NJH; ++jb) {
        acc[jb][0] = mfma_f64(bf[t & 1][jb], rA[s & 1][q][0], acc[jb][0]);
        acc[jb][1] = mfma_f64(bf[t & 1][jb], rA[s & 1][q][1], acc[jb][1]);
      }
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
  for (int jb = 0; jb < NJH; ++jb)
#pragma unroll
    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = ib == 0 ? ia : ib_, jc = 64 * ch + 16 * jb + lk + 4 * r;
        const double v = acc[jb][ib][r];
        if (slot < 0) {
          if (i < nrow && jc < wj) P[(int64_t)jc * mj + R0 + i] -= v;
        } else {
          scratch[(int64_t)slot * (TM * NB) + jc * TM + i] = v;
        }
      }
}


}  // namespace scilmm
