// Microbenchmark, round 3: the fp32-product loop of mfma_f32_feed with ONE wave per SIMD -- 256-thread workgroups, a wave owns
// 64 rows x 128 columns (fp64 accumulators 256 registers + fp32 accumulators 128: no spills in a 512-register budget, every B
// fragment feeds four MFMAs instead of two).  Real-launch shape (5205 workgroups x 96 chunks).
//   hipcc -O3 --offload-arch=gfx950 mfma_f32_wave64.hip -o mfma_f32_wave64 && ./mfma_f32_wave64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_vptr;
typedef const __attribute__((address_space(1))) void* gl_vptr;
constexpr int KR = 64, LDBF = 144;

__global__ __launch_bounds__(256, 1) void k_feed(double* out, const float* __restrict__ Ag, int md, int iters) {
  extern __shared__ float Bf[];  // [2][KR][LDBF]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  f4 c[8][4];
  d4 acc[8][4];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) { c[a][b] = (f4){0, 0, 0, 0}; acc[a][b] = (d4){0, 0, 0, 0}; }
  const int npair = 347;
  const float* Abase = Ag + (size_t)(blockIdx.x % npair) * 256 + 64 * wv + li;  // rows +0, +16, +32, +48
  const float* Bg = Ag;
  float rA[2][4][4];
  auto load_A = [&](long k0, int sub, float (&a)[4][4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float* p = Abase + (size_t)(k0 + 16 * sub + 4 * q + lk) * md;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) a[q][rb] = p[16 * rb];
    }
  };
  auto issue_B = [&](long k0, int b) {
    float* Bs = Bf + b * KR * LDBF;
#pragma unroll
    for (int i = 0; i < KR / 4; ++i) {
      const int kr = wv + 4 * i;
      const float* row = Bg + (size_t)(k0 + kr) * md;
      __builtin_amdgcn_global_load_lds((gl_vptr)(row + lane), (lds_vptr)(Bs + kr * LDBF), 4, 0, 0);
      __builtin_amdgcn_global_load_lds((gl_vptr)(row + 64 + lane), (lds_vptr)(Bs + kr * LDBF + 64), 4, 0, 0);
    }
  };
  long kpos = 64L * iters * (blockIdx.x / npair);
  load_A(kpos, 0, rA[0]);
  issue_B(kpos, 0);
  issue_B(kpos + 64, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  float bf[2][8];
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) bf[0][jb] = Bf[lk * LDBF + 16 * jb + li];
  for (int it = 0; it < iters; ++it) {
    const float* Bc = Bf + buf * KR * LDBF;
    const float* Bn = Bf + (buf ^ 1) * KR * LDBF;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int s = t >> 2, q = t & 3;
      if (q == 0) { if (s < 3) load_A(kpos, s + 1, rA[(s + 1) & 1]); else load_A(kpos + 64, 0, rA[0]); }
      if (t < 15) {
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) bf[(t + 1) & 1][jb] = Bc[(4 * (t + 1) + lk) * LDBF + 16 * jb + li];
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) bf[0][jb] = Bn[lk * LDBF + 16 * jb + li];
        issue_B(kpos + 128, buf);
      }
#pragma unroll
      for (int jb = 0; jb < 8; ++jb)
#pragma unroll
        for (int rb = 0; rb < 4; ++rb)
          c[jb][rb] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[t & 1][jb], rA[s & 1][q][rb], c[jb][rb], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int jb = 0; jb < 8; ++jb)
#pragma unroll
      for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[jb][rb][r] += (double)c[jb][rb][r];
        c[jb][rb] = (f4){0, 0, 0, 0};
      }
    kpos += 64;
    buf ^= 1;
  }
  double s = 0;
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += acc[a][b][r];
  out[blockIdx.x * 256 + tid] = s;
}

int main() {
  const int md = 347 * 256 + 64, iters = 96;
  double* d;
  float* A;
  (void)hipMalloc(&d, sizeof(double) * 256 * 8192);
  (void)hipMalloc(&A, sizeof(float) * (size_t)md * 64 * (size_t)(15 * iters + 4));
  (void)hipMemset(A, 0, sizeof(float) * (size_t)md * 64 * (size_t)(15 * iters + 4));
  (void)hipFuncSetAttribute((const void*)k_feed, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int grid : {256, 2560, 5205}) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_feed, dim3(grid), dim3(256), sizeof(float) * 2 * KR * LDBF, 0, d, A, md, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double flops = (double)grid * 4 * iters * 16 * 32 * 2048.0;
      if (rep == 2) printf("one wave per SIMD, wave tile 64 x 128, %d workgroups: %.3f ms -> %.1f TFLOP/s (%s)\n", grid, ms, flops / ms / 1e9, hipGetErrorString(hipGetLastError()));
    }
  }
  return 0;
}
