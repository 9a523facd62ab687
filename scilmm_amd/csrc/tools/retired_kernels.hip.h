// RETIRED kernel variants -- tuning-harness material only (csrc/tools/dense_bench2.hip); nothing in libscilmm_hip.so includes
// this file.  What is kept is what dense_bench2 still times side by side with the kernels of the hot path (DESIGN.md sections
// 4.0 / 4.1: every one of them same speed or slower, same results):
//   k_dense_a       -> k_dense_b   (A from registers, B by LDS-DMA, without the in-wave software pipeline of round 3)
//   k_dense_f / _s / _t (+ k_shadow_t) / _w   the forms that located what bounds the fp32-product fronts (-> k_dense_h)
//   k_dense_q                      k_dense_h's 32 x 64 wave tiling for the fp64 kernel (63.6 vs k_dense_b's 68.0)
// Pruned in round 4 (their numbers stay in DESIGN.md section 4.2, their code in the history before commit "prune the harness"):
// k_update, k_update3, k_dense<>, k_dense_g, k_update_compact, k_trsm4 and the round-2 harness dense_bench.hip.
// Include AFTER ../kernels.hip.h.
#pragma once
#include "../kernels.hip.h"

namespace scilmm {

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
