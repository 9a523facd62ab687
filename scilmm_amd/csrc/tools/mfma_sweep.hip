// Microbenchmark: is the sustained rate of v_mfma_f64_16x16x4_f64 (49 TF of a 78.6 TF datasheet figure, see
// mfma_peak.hip) limited per SIMD (issue rate) or chip-wide (power / clock management)?  Runs the same
// register-only MFMA loop on 8 .. 2048 workgroups and reports, per grid size, the time per MFMA per wave measured
// INSIDE the kernel with the 100 MHz wall clock, plus the 4x4x4 (4-block) form of the instruction.
//   hipcc -O3 --offload-arch=gfx950 mfma_sweep.hip -o mfma_sweep && ./mfma_sweep
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void k_loop(double* out, unsigned long long* ticks, int iters, double a0, double b0) {
  d4 acc[8];
  double acc1[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { acc[i] = (d4){0, 0, 0, 0}; acc1[i] = 0; }
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  const unsigned long long t0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      else acc1[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc1[i], 0, 0, 0);
    }
  }
  const unsigned long long t1 = wall_clock64();
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + acc1[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) ticks[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

int main() {
  double* d;
  unsigned long long* t;
  const int maxg = 4096;
  (void)hipMalloc(&d, sizeof(double) * 256 * maxg);
  (void)hipMalloc(&t, sizeof(unsigned long long) * 4 * maxg);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int kind = 0; kind < 2; ++kind) {
    const double flop_per = kind == 0 ? 2048.0 : 512.0;  // 16x16x4 MACs x2 ; 4 blocks of 4x4x4 x2
    for (int grid : {8, 32, 64, 128, 256, 512, 1024, 2048}) {
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (kind == 0) hipLaunchKernelGGL(k_loop<0>, dim3(grid), dim3(256), 0, 0, d, t, iters, 1.0, 1e-3);
        else hipLaunchKernelGGL(k_loop<1>, dim3(grid), dim3(256), 0, 0, d, t, iters, 1.0, 1e-3);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      std::vector<unsigned long long> h(4 * grid);
      (void)hipMemcpy(h.data(), t, sizeof(unsigned long long) * 4 * grid, hipMemcpyDeviceToHost);
      std::sort(h.begin(), h.end());
      const double med_ns = h[h.size() / 2] * 10.0 / ((double)iters * 8);  // 100 MHz ticks -> ns per MFMA per wave
      const double flops = (double)grid * 4 * iters * 8 * flop_per;
      printf("%s grid %4d WGs (256 thr): kernel %.3f ms -> %.2f TFLOP/s; median %.1f ns per MFMA per wave (= %.0f cycles @2.4GHz), min %.1f max %.1f\n",
             kind == 0 ? "mfma_f64_16x16x4" : "mfma_f64_4x4x4_4b", grid, ms, flops / ms / 1e9, med_ns, med_ns * 2.4,
             h.front() * 10.0 / ((double)iters * 8), h.back() * 10.0 / ((double)iters * 8));
    }
  }
  return 0;
}
