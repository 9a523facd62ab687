// Microbenchmark, round 3: mfma_f32_feed's question for the fp64 dense-tail kernel.  A synthetic loop with k_dense_b's operand
// feeds -- 16 v_mfma_f64_16x16x4_f64 per k-step and wave, A fragments (two rows per lane) straight from global memory one
// sub-chunk ahead, B image (64 k-rows x 128 columns) by LDS-DMA with one barrier per 64 k, fragments prefetched one k-step
// ahead -- with the 256 workgroups either walking the operand matrix in LOCKSTEP (long contiguous runs per k-column) or each at
// ITS OWN column (2 KB pieces, as a real launch's drifting items produce).
//   hipcc -O3 --offload-arch=gfx950 mfma_f64_feed.hip -o mfma_f64_feed && ./mfma_f64_feed
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* gl_vptr;
constexpr int KR = 64, LDB = 144;

template <bool DESYNC>
__global__ __launch_bounds__(512, 1) void k_feed(double* out, const double* __restrict__ Ag, int md, int iters) {
  extern __shared__ double sm[];  // [2][KR][LDB]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  d4 c[8][2];
#pragma unroll
  for (int a = 0; a < 8; ++a) { c[a][0] = (d4){0, 0, 0, 0}; c[a][1] = (d4){0, 0, 0, 0}; }
  const double* A0 = Ag + (size_t)blockIdx.x * 256 + 32 * wv + li;  // this lane's rows: A0, A0 + 16
  const double* Bg = Ag;                                             // B rows: the matrix's first 128 rows (shared by all workgroups)
  double rA[2][4][2];
  auto load_A = [&](long k0, int sub, double (&a)[4][2]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const double* p = A0 + (size_t)(k0 + 16 * sub + 4 * q + lk) * md;
      a[q][0] = p[0];
      a[q][1] = p[16];
    }
  };
  auto issue_B = [&](long k0, int b) {
    double* Bs = sm + b * KR * LDB;
#pragma unroll
    for (int i = 0; i < KR / 8; ++i) {
      const int kr = wv + 8 * i;
      __builtin_amdgcn_global_load_lds((gl_vptr)(Bg + (size_t)(k0 + kr) * md + 2 * lane), (lds_vptr)(Bs + kr * LDB), 16, 0, 0);
    }
  };
  const long kwrap = 64L * iters;
  long kpos = DESYNC ? 64L * ((blockIdx.x * 37) % iters) : 0;
  load_A(kpos, 0, rA[0]);
  issue_B(kpos, 0);
  issue_B(kpos + 64, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  double bf[2][8];
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) bf[0][jb] = sm[lk * LDB + 16 * jb + li];
  for (int it = 0; it < iters; ++it) {
    const double* Bc = sm + buf * KR * LDB;
    const double* Bn = sm + (buf ^ 1) * KR * LDB;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int s = t >> 2, q = t & 3;
      if (q == 0) { if (s < 3) load_A(kpos, s + 1, rA[(s + 1) & 1]); else load_A(kpos + 64, 0, rA[0]); }
      if (t < 15) {
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) bf[(t + 1) & 1][jb] = Bc[(4 * (t + 1) + lk) * LDB + 16 * jb + li];
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) bf[0][jb] = Bn[lk * LDB + 16 * jb + li];
        issue_B(kpos + 128, buf);
      }
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) {
        c[jb][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[t & 1][jb], rA[s & 1][q][0], c[jb][0], 0, 0, 0);
        c[jb][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(bf[t & 1][jb], rA[s & 1][q][1], c[jb][1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
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
      for (int r = 0; r < 4; ++r) s += c[a][b][r];
  out[blockIdx.x * 512 + tid] = s;
}

template <bool DESYNC>
void run(double* d, const double* A, int md, int iters, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = 256;
  const size_t lds = sizeof(double) * 2 * KR * LDB;
  (void)hipFuncSetAttribute((const void*)k_feed<DESYNC>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_feed<DESYNC>), dim3(grid), dim3(512), lds, 0, d, A, md, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 8 * iters * 16 * 16 * 2048.0;
    if (rep == 2) printf("%-52s %.3f ms -> %.1f TFLOP/s (%s)\n", what, ms, flops / ms / 1e9, hipGetErrorString(hipGetLastError()));
  }
}

int main() {
  const int md = 256 * 256 + 64, iters = 400;
  const size_t K = 64 * (size_t)(iters + 4);
  double *d, *A;
  (void)hipMalloc(&d, sizeof(double) * 512 * 256);
  (void)hipMalloc(&A, sizeof(double) * (size_t)md * K);  // 13.6 GB
  (void)hipMemset(A, 0, sizeof(double) * (size_t)md * K);
  run<false>(d, A, md, iters, "k_dense_b's feeds, workgroups in lockstep");
  run<true>(d, A, md, iters, "k_dense_b's feeds, every workgroup at its own column");
  return 0;
}
