// Microbenchmark, round 3 (continues mfma_f32_mix): which FEED of the fp32-product dense-tail kernels costs them the matrix
// pipe?  Base = variant 3 of mfma_f32_mix (16 MFMAs + 8 fp32 LDS fragments per k-step, fold every 16 k-steps: 135 TFLOP/s),
// 512-thread workgroups, one per CU.  Added one at a time:
//   A  the A fragments come from GLOBAL memory (8 bytes per lane and k-step, a quarter wave reads 128 contiguous bytes, each
//      workgroup streams its own region, loads issued one 4-k-step sub-chunk ahead), like k_dense_s
//   D  the B image is refilled by LDS-DMA (global_load_lds_dword, 64 k-rows per buffer, two buffers, every workgroup the same
//      source: L2 hits) with one barrier per 16 k-steps, like k_dense_s
//   AD both
//   hipcc -O3 --offload-arch=gfx950 mfma_f32_feed.hip -o mfma_f32_feed && ./mfma_f32_feed
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* gl_vptr;
constexpr int KR = 64, LDBF = 144;

template <bool GA, bool DMA, int FEAT>
__global__ __launch_bounds__(512, 1) void k_feed(double* out, const float* __restrict__ Ag, const float* __restrict__ Bg, int md, int iters,
                                               const int* __restrict__ desc, const float* __restrict__ zeros) {
  constexpr bool PF = FEAT & 1, SEL = FEAT & 2, DESC = FEAT & 4, DESYNC = FEAT & 8, ITEMS = FEAT & 16;
  extern __shared__ float sm[];
  float* Bf = sm;                    // [2][KR][LDBF]
  float* Af = sm + 2 * KR * LDBF;    // [KR][272]   (A fragments when they do not come from global memory)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < 2 * KR * LDBF; i += 512) Bf[i] = 1e-3f * (i % 5);
  for (int i = tid; i < KR * 272; i += 512) Af[i] = 1e-3f * (i % 7);
  __syncthreads();
  const int li = lane & 15, lk = lane >> 4;
  f4 c[8][2];
  d4 acc[8][2];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) { c[a][b] = (f4){0, 0, 0, 0}; acc[a][b] = (d4){0, 0, 0, 0}; }
  // this workgroup's A region: rows R0 .. R0 + 255 of a column-major [md x K] float matrix; k advances with the chunks
  const int npair = 347;
  const float* Abase = Ag + (size_t)(ITEMS ? blockIdx.x % npair : blockIdx.x) * 256 + 32 * wv + 2 * li;
  f2 rA[2][4];
  int kc_cur = 64;  // SEL: chunk depth (always 64 here, but the kernel does not know)
  auto load_A = [&](long k0, int sub, f2 (&a)[4]) {
    const int klast = (kc_cur - 1) & ~3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kq = SEL ? min(16 * sub + 4 * q, klast) : 16 * sub + 4 * q;
      a[q] = *(const f2*)(Abase + (size_t)(k0 + kq + lk) * md);
    }
  };
  auto issue_B = [&](long k0, int b) {
    float* Bs = Bf + b * KR * LDBF;
    int kc = 64;
    if (DESC) {  // chunk depth and column count out of arrays, as next_chunk() does
      const int dd = __builtin_amdgcn_readfirstlane(desc[(k0 >> 6) & 1023]);
      kc = __builtin_amdgcn_readfirstlane(min(64, desc[1024 + dd] - desc[dd]));
    }
    if (SEL) kc_cur = kc;
#pragma unroll
    for (int i = 0; i < KR / 8; ++i) {
      const int kr = wv + 8 * i;
      const float* row = Bg + (size_t)(k0 + kr) * md;
      const bool on = !SEL || kr < kc;
      __builtin_amdgcn_global_load_lds((gl_vptr)(on ? row + lane : zeros + lane), (lds_vptr)(Bs + kr * LDBF), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((gl_vptr)(on ? row + 64 + lane : zeros + lane), (lds_vptr)(Bs + kr * LDBF + 64), 4, 0, 0);
    }
  };
  const long kwrap = 64L * iters;
  long kpos = ITEMS ? 64L * iters * (blockIdx.x / npair) : DESYNC ? 64L * ((blockIdx.x * 37) % iters) : 0;
  if (GA) load_A(kpos, 0, rA[0]);
  if (DMA) {
    issue_B(kpos, 0);
    issue_B(kpos + 64, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  int buf = 0;
  float bfr[2][8];
  if (PF) {
#pragma unroll
    for (int jb = 0; jb < 8; ++jb) bfr[0][jb] = Bf[lk * LDBF + 16 * jb + li];
  }
  for (int it = 0; it < iters; ++it) {
    const float* Bc = Bf + (DMA ? buf : 0) * KR * LDBF;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int k4 = 4 * t, s = t >> 2, q = t & 3;
      float a0, a1;
      if (GA) {
        if (q == 0) { if (s < 3) load_A(kpos, s + 1, rA[(s + 1) & 1]); else load_A(kpos + 64, 0, rA[0]); }
        a0 = rA[s & 1][q][0];
        a1 = rA[s & 1][q][1];
      } else {
        a0 = Af[(k4 + lk) * 272 + 32 * wv + li];
        a1 = Af[(k4 + lk) * 272 + 32 * wv + 16 + li];
      }
      if (!PF) {
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) bfr[t & 1][jb] = Bc[(k4 + lk) * LDBF + 16 * jb + li];
      } else if (t < 15) {
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) bfr[(t + 1) & 1][jb] = Bc[(k4 + 4 + lk) * LDBF + 16 * jb + li];
      }
      if (DMA && t == 15) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (PF) {
          const float* Bn = Bf + (buf ^ 1) * KR * LDBF;
#pragma unroll
          for (int jb = 0; jb < 8; ++jb) bfr[0][jb] = Bn[lk * LDBF + 16 * jb + li];
        }
        issue_B(kpos + 128, buf);
      } else if (PF && t == 15) {
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) bfr[0][jb] = Bc[lk * LDBF + 16 * jb + li];
      }
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) {
        c[jb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[t & 1][jb], a0, c[jb][0], 0, 0, 0);
        c[jb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfr[t & 1][jb], a1, c[jb][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
      for (int ib = 0; ib < 2; ++ib) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[jb][ib][r] += (double)c[jb][ib][r];
        c[jb][ib] = (f4){0, 0, 0, 0};
      }
    kpos += 64;
    if (DESYNC && kpos >= kwrap) kpos -= kwrap;
    buf ^= 1;
  }
  double s = 0;
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += acc[a][b][r];
  out[blockIdx.x * 512 + tid] = s;
}

template <bool GA, bool DMA, int FEAT>
void run(double* d, const float* A, const float* B, int md, int iters, const int* desc, const float* zeros, const char* what, int grid = 256) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const size_t lds = sizeof(float) * (2 * KR * LDBF + KR * 272);
  (void)hipFuncSetAttribute((const void*)k_feed<GA, DMA, FEAT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_feed<GA, DMA, FEAT>), dim3(grid), dim3(512), lds, 0, d, A, B, md, iters, desc, zeros);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 8 * iters * 16 * 16 * 2048.0;
    if (rep == 2) printf("%-44s %.3f ms -> %.1f TFLOP/s (%s)\n", what, ms, flops / ms / 1e9, hipGetErrorString(hipGetLastError()));
  }
}

int main() {
  const int md = 256 * 256 + 64, iters = 400;          // 256 workgroups x 256 rows; K = 64 * (iters + 3) columns
  const size_t K = 64 * (size_t)(iters + 4);
  double* d;
  float *A, *B;
  (void)hipMalloc(&d, sizeof(double) * 512 * 8192);
  (void)hipMalloc(&A, sizeof(float) * (size_t)md * K);  // 6.9 GB: every workgroup streams its own rows, nothing is re-read
  (void)hipMemset(A, 0, sizeof(float) * (size_t)md * K);
  B = A;                                                 // the B rows: columns 0..127 of the same matrix (shared by all workgroups)
  int* desc;
  float* zeros;
  (void)hipMalloc(&desc, sizeof(int) * 4096);
  (void)hipMalloc(&zeros, 4096);
  (void)hipMemset(zeros, 0, 4096);
  {
    int h[4096];
    for (int i = 0; i < 1024; ++i) { h[i] = i; h[1024 + i] = 128 * i; }
    for (int i = 2048; i < 4096; ++i) h[i] = 128 * (i - 1024);
    (void)hipMemcpy(desc, h, sizeof(h), hipMemcpyHostToDevice);
  }
  run<false, false, 0>(d, A, B, md, iters, desc, zeros, "base (LDS fragments, fold / 16 k-steps)");
  run<true, false, 0>(d, A, B, md, iters, desc, zeros, "+ A fragments from global memory");
  run<false, true, 0>(d, A, B, md, iters, desc, zeros, "+ B image by LDS-DMA, barrier / 16 k-steps");
  run<true, true, 0>(d, A, B, md, iters, desc, zeros, "+ both");
  run<true, true, 1>(d, A, B, md, iters, desc, zeros, "both + fragments prefetched one k-step ahead");
  run<true, true, 2>(d, A, B, md, iters, desc, zeros, "both + depth tests (zero page select, clamp)");
  run<true, true, 3>(d, A, B, md, iters, desc, zeros, "both + prefetch + depth tests");
  run<true, true, 7>(d, A, B, md, iters, desc, zeros, "both + prefetch + depth tests + descriptors");
  run<true, true, 8>(d, A, B, md, iters, desc, zeros, "both, every workgroup at its own k position");
  run<true, true, 11>(d, A, B, md, iters, desc, zeros, "both + prefetch + depth tests, own k positions");
  {
    // a launch shaped like the real one: 15 K segments x 347 row pairs = 5205 workgroups of 96 chunks, segment-major
    const int md2 = 347 * 256 + 64, it2 = 96;
    float* A2;
    (void)hipFree(A);
    (void)hipMalloc(&A2, sizeof(float) * (size_t)md2 * 64 * (size_t)(15 * it2 + 4));
    (void)hipMemset(A2, 0, sizeof(float) * (size_t)md2 * 64 * (size_t)(15 * it2 + 4));
    run<true, true, 16 + 3>(d, A2, A2, md2, it2, desc, zeros, "5205 items (15 segments x 347 row pairs), 96 chunks each", 5205);
    run<true, true, 16 + 3>(d, A2, A2, md2, it2, desc, zeros, "... 256 of them (one round)", 256);
    run<true, true, 16 + 3>(d, A2, A2, md2, it2, desc, zeros, "... 2560 of them (ten full rounds)", 2560);
  }
  return 0;
}
