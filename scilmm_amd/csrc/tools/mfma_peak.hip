// Microbenchmark: sustained rate of v_mfma_f64_16x16x4_f64 (and the fp64 VALU FMA rate beside it) on
// this GPU.  MI355X_MICROARCH.md lists no fp64 MFMA figure; DESIGN.md quotes what this prints.
//   hipcc -O3 --offload-arch=gfx950 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k_mfma(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void k_fma(double* out, int iters, double a0, double b0) {
  double acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = i;
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_fma(a, acc[i], b);
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Do the matrix pipe and the vector ALU overlap?  Same two kernels on two streams at once.
static void both(double* d, int iters) {
  hipStream_t s0, s1;
  hipStreamCreateWithFlags(&s0, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
  hipEvent_t e0, e1, e2;
  hipEventCreate(&e0); hipEventCreate(&e1); hipEventCreate(&e2);
  for (int rep = 0; rep < 2; ++rep) {
    hipDeviceSynchronize();
    hipEventRecord(e0, s0);
    hipStreamWaitEvent(s1, e0, 0);
    hipLaunchKernelGGL(k_mfma<8>, dim3(256 * 4), dim3(256), 0, s0, d, iters, 1.0, 1e-3);
    hipLaunchKernelGGL(k_fma, dim3(256 * 4), dim3(256), 0, s1, d + 256 * 2048, iters, 1.0000001, 1e-3);
    hipEventRecord(e1, s0);
    hipEventRecord(e2, s1);
    hipEventSynchronize(e1); hipEventSynchronize(e2);
    float m0, m1;
    hipEventElapsedTime(&m0, e0, e1);
    hipEventElapsedTime(&m1, e0, e2);
    const double fm = 256.0 * 4 * 4 * iters * 8 * 2048.0, fv = 256.0 * 4 * 256 * iters * 16 * 2.0;
    if (rep) printf("concurrent: mfma kernel %.3f ms, fma kernel %.3f ms -> %.1f TF combined over the longer one\n", m0, m1,
                    (fm + fv) / (m0 > m1 ? m0 : m1) / 1e9);
  }
}

int main() {
  double* d;
  (void)hipMalloc(&d, sizeof(double) * 256 * 4096);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  both(d, iters);
  for (int wgs_per_cu = 1; wgs_per_cu <= 8; wgs_per_cu *= 2) {
    int grid = 256 * wgs_per_cu;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_mfma<8>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0, 1e-3);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)grid * 4 * iters * 8 * 2048.0;
      if (rep) printf("mfma_f64_16x16x4 x8 acc, %d WG/CU: %.2f TFLOP/s (%.3f ms) -> %.1f cycles/MFMA/SIMD @2.4GHz\n", wgs_per_cu,
                      flops / ms / 1e9, ms, ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wgs_per_cu));
    }
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_mfma<1>, dim3(grid), dim3(256), 0, 0, d, iters, 1.0, 1e-3);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)grid * 4 * iters * 1 * 2048.0;
      if (rep) printf("mfma_f64_16x16x4 dependent chain, %d WG/CU: %.2f TFLOP/s -> %.1f cycles/MFMA\n", wgs_per_cu, flops / ms / 1e9,
                      ms * 1e-3 * 2.4e9 / ((double)iters * wgs_per_cu));
    }
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k_fma, dim3(grid), dim3(256), 0, 0, d, iters, 1.0000001, 1e-3);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      double flops = (double)grid * 256 * iters * 16 * 2.0;
      if (rep) printf("v_fma_f64 x16 acc, %d WG/CU: %.2f TFLOP/s\n", wgs_per_cu, flops / ms / 1e9);
    }
  }
  return 0;
}
