// Microbenchmark: the f64 MFMA rate WITH the operand traffic of a real GEMM tile loop -- every MFMA operand comes
// out of LDS with the same read pattern as k_dense (16x16x4 form: 10 ds_read_b64 per 16 MFMAs; 4x4x4 form: 20 per
// 64), no global traffic.  Also reports the shader clock the waves saw (s_memtime ticks per 100 MHz wall-clock
// tick), to tell an issue limit from a power / clock limit.
//   hipcc -O3 --offload-arch=gfx950 mfma_lds.hip -o mfma_lds && ./mfma_lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int KC = 16, LDA2 = 272, LDB = 144;

template <int MF>
__global__ __launch_bounds__(512, 1) void k_mix(double* out, unsigned long long* ticks, int iters) {
  __shared__ double As[KC * LDA2];
  __shared__ double Bs[KC * LDB];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < KC * LDA2; i += 512) As[i] = 1e-3 * (i % 7);
  for (int i = tid; i < KC * LDB; i += 512) Bs[i] = 1e-3 * (i % 5);
  __syncthreads();
  const int li = lane & 15, lk = lane >> 4, l3 = lane & 3;
  d4 acc16[8][2];
  double acc4[16][4];
#pragma unroll
  for (int a = 0; a < 8; ++a) { acc16[a][0] = (d4){0, 0, 0, 0}; acc16[a][1] = (d4){0, 0, 0, 0}; }
#pragma unroll
  for (int a = 0; a < 16; ++a)
#pragma unroll
    for (int q = 0; q < 4; ++q) acc4[a][q] = 0;
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll 2
    for (int k4 = 0; k4 < KC; k4 += 4) {
      if (MF == 16) {
        const double a0 = As[(k4 + lk) * LDA2 + 32 * wv + li], a1 = As[(k4 + lk) * LDA2 + 32 * wv + 16 + li];
        double b[8];
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) b[jb] = Bs[(k4 + lk) * LDB + 16 * jb + li];
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) {
          acc16[jb][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[jb], a0, acc16[jb][0], 0, 0, 0);
          acc16[jb][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(b[jb], a1, acc16[jb][1], 0, 0, 0);
        }
      } else {
        double cv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) cv[q] = Bs[(k4 + lk) * LDB + 16 * wv + 4 * q + l3];
#pragma unroll
        for (int pr = 0; pr < 16; ++pr) {
          const double rv = As[(k4 + lk) * LDA2 + 16 * pr + li];
#pragma unroll
          for (int q = 0; q < 4; ++q) acc4[pr][q] = __builtin_amdgcn_mfma_f64_4x4x4f64(cv[q], rv, acc4[pr][q], 0, 0, 0);
        }
      }
    }
  }
  const unsigned long long w1 = wall_clock64(), c1 = clock64();
  double s = 0;
  if (MF == 16) {
#pragma unroll
    for (int a = 0; a < 8; ++a) s += acc16[a][0][0] + acc16[a][0][1] + acc16[a][0][2] + acc16[a][0][3] + acc16[a][1][0] + acc16[a][1][1] + acc16[a][1][2] + acc16[a][1][3];
  } else {
#pragma unroll
    for (int a = 0; a < 16; ++a) s += acc4[a][0] + acc4[a][1] + acc4[a][2] + acc4[a][3];  // every accumulator is live
  }
  out[blockIdx.x * 512 + tid] = s;
  if (lane == 0) { ticks[(blockIdx.x * 8 + wv) * 2] = w1 - w0; ticks[(blockIdx.x * 8 + wv) * 2 + 1] = c1 - c0; }
}

int main() {
  double* d;
  unsigned long long* t;
  const int maxg = 1024;
  (void)hipMalloc(&d, sizeof(double) * 512 * maxg);
  (void)hipMalloc(&t, sizeof(unsigned long long) * 16 * maxg);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 4000;
  for (int mf : {16, 4})
    for (int grid : {32, 256, 512}) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mf == 16) hipLaunchKernelGGL(k_mix<16>, dim3(grid), dim3(512), 0, 0, d, t, iters);
        else hipLaunchKernelGGL(k_mix<4>, dim3(grid), dim3(512), 0, 0, d, t, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      std::vector<unsigned long long> h(16 * grid);
      (void)hipMemcpy(h.data(), t, sizeof(unsigned long long) * 16 * grid, hipMemcpyDeviceToHost);
      double wsum = 0, csum = 0;
      for (int i = 0; i < 8 * grid; ++i) { wsum += h[2 * i]; csum += h[2 * i + 1]; }
      const double flops = (double)grid * 2.0 * 256 * 128 * KC * iters;  // one 256 x 128 x 16 chunk per iteration
      printf("MFMA form %2d, %4d workgroups of 8 waves (1 per CU): %.2f TFLOP/s (kernel %.2f ms); clock64/wall_clock64 = %.3f (x100 MHz)\n", mf,
             grid, flops / ms / 1e9, ms, csum / wsum);
    }
  return 0;
}
