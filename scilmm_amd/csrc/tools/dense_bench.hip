// Kernel-tuning harness: k_dense (csrc/kernels.hip.h) ALONE on a synthetic dense tail -- one level's early update
// (target panel j of a T-panel dense lower-triangular matrix, all tile pairs, K split into segments), no other
// stream, no host logic.  Prints the sustained TFLOP/s of the launch for both f64 MFMA forms.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DSCILMM_DENSE_CLK dense_bench.hip -o dense_bench && ./dense_bench [T] [j] [panels per item]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "retired_kernels.hip.h"
using namespace scilmm;

// operands: 0 = zeros (read HIGH: zero MFMA operands raise the clock), 1 = full-range uniform [-1, 1)
__global__ void k_fill(double* p, size_t n, int mode) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned long long x = i * 0x9E3779B97F4A7C15ull + 0x1234567ull;
    x ^= x >> 31; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 29;
    p[i] = mode == 0 ? 0.0 : ((double)(x >> 11) * (1.0 / 9007199254740992.0)) * 2.0 - 1.0;
  }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main(int argc, char** argv) {
  const int T = argc > 1 ? atoi(argv[1]) : 400;        // panels of the tail
  const int j = argc > 2 ? atoi(argv[2]) : 300;        // target panel
  const int per = argc > 3 ? atoi(argv[3]) : 10;       // descendants per item
  const int n = NB * T;
  std::vector<int32_t> sn_start(T + 1);
  std::vector<int64_t> sn_loff(T + 1, 0);
  for (int k = 0; k <= T; ++k) sn_start[k] = NB * k;
  for (int k = 0; k < T; ++k) sn_loff[k + 1] = sn_loff[k] + (int64_t)(n - NB * k) * NB;
  double* L;
  CK(hipMalloc(&L, sizeof(double) * (size_t)sn_loff[T]));
  CK(hipMemset(L, 0, sizeof(double) * (size_t)sn_loff[T]));
  DevSym S{};
  S.n = n;
  S.nsuper = T;
  int32_t* d_start; int64_t* d_loff;
  CK(hipMalloc(&d_start, sizeof(int32_t) * (T + 1)));
  CK(hipMalloc(&d_loff, sizeof(int64_t) * (T + 1)));
  CK(hipMemcpy(d_start, sn_start.data(), sizeof(int32_t) * (T + 1), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_loff, sn_loff.data(), sizeof(int64_t) * (T + 1), hipMemcpyHostToDevice));
  S.sn_start = d_start;
  S.sn_loff = d_loff;
  std::vector<DenseWork> work;
  const int ntl = T - j;  // tiles of panel j
  int slot = 0;
  for (int k0 = 0; k0 < j; k0 += per)  // K-segment major, tile-pair minor (the engine's launch order)
    for (int q = 0; q < ntl; q += 2) {
      const int nt2 = ntl - q >= 2 ? 2 : 1;
      DenseWork w{j, q, nt2, k0, k0 + per < j ? k0 + per : j, slot, nt2 == 2 ? slot + 1 : -1, 0};
      slot += 2;
      work.push_back(w);
    }
  DenseWork* d_work;
  CK(hipMalloc(&d_work, sizeof(DenseWork) * work.size()));
  CK(hipMemcpy(d_work, work.data(), sizeof(DenseWork) * work.size(), hipMemcpyHostToDevice));
  double* scratch;
  CK(hipMalloc(&scratch, sizeof(double) * (size_t)slot * TM * NB));
  const size_t sm = sizeof(double) * (size_t)(2 * KC * LDA2 + 2 * KC * LDB);
  CK(hipFuncSetAttribute((const void*)k_dense<16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_dense<4, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_dense_g, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_dense_a, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  double* zeros;
  CK(hipMalloc(&zeros, 2048));
  CK(hipMemset(zeros, 0, 2048));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double flops = 0;
  for (auto& w : work) flops += 2.0 * (w.ntiles * TM) * NB * (double)NB * (w.k1 - w.k0);
  printf("T=%d panels, target %d: %zu items (%d descendants each), %.3f TFLOP per launch, %.1f MB of slabs\n", T, j, work.size(), per,
         flops / 1e12, slot * TM * NB * 8 / 1e6);
  for (int fill : {0, 1})
  for (int mf : {16, 160, 170}) {
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, L, (size_t)sn_loff[T], fill);
    for (int rep = 0; rep < 3; ++rep) {
      unsigned long long z[2] = {0, 0};
      CK(hipMemcpyToSymbol(HIP_SYMBOL(g_dense_clk), z, sizeof(z)));
      hipEventRecord(e0);
      if (mf == 16) hipLaunchKernelGGL((k_dense<16, true>), dim3((unsigned)work.size()), dim3(512), sm, 0, S, 0, d_work, L, scratch);
      else if (mf == 160) hipLaunchKernelGGL(k_dense_g, dim3((unsigned)work.size()), dim3(512), sm, 0, S, 0, d_work, L, scratch, (const double*)zeros);
      else if (mf == 170) hipLaunchKernelGGL(k_dense_a, dim3((unsigned)work.size()), dim3(512), sizeof(double) * 2 * KBA * LDB, 0, S, 0, d_work, L, scratch, (const double*)zeros);
      else if (mf == 4) hipLaunchKernelGGL((k_dense<4, true>), dim3((unsigned)work.size()), dim3(512), sm, 0, S, 0, d_work, L, scratch);
      else hipLaunchKernelGGL(k_dense32, dim3((unsigned)work.size()), dim3(512), sizeof(float) * (size_t)(2 * KC * LDA2F + 2 * KC * LDBF), 0, S, 0, d_work, L, scratch);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      CK(hipGetLastError());
      CK(hipMemcpyFromSymbol(z, HIP_SYMBOL(g_dense_clk), sizeof(z)));
      if (rep && mf == 170) printf("k_dense_a (A from registers, B 64 deep by LDS-DMA), %s operands: %.3f ms -> %.2f TFLOP/s\n", fill ? "random" : "zero", ms, flops / ms / 1e9);
      else if (rep && mf == 160) printf("k_dense_g (LDS-DMA staging), %s operands: %.3f ms -> %.2f TFLOP/s\n", fill ? "random" : "zero", ms, flops / ms / 1e9);
      else if (rep && mf != 32) printf("k_dense<%d>, %s operands: %.3f ms -> %.2f TFLOP/s at %.2f GHz shader clock\n", mf, fill ? "random" : "zero", ms,
                      flops / ms / 1e9, 0.1 * (double)z[1] / (double)z[0]);
      if (rep && mf == 32) printf("k_dense32 (fp32 products, fp64 sums), %s operands: %.3f ms -> %.2f TFLOP/s\n", fill ? "random" : "zero", ms, flops / ms / 1e9);
    }
  }
  {
    // same slabs from both staging forms (random operands; the arithmetic order is identical, so the bits are too)
    const size_t ns = (size_t)slot * TM * NB;
    std::vector<double> r0(ns), r1(ns);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, L, (size_t)sn_loff[T], 1);
    CK(hipMemset(scratch, 0, sizeof(double) * ns));
    hipLaunchKernelGGL((k_dense<16, true>), dim3((unsigned)work.size()), dim3(512), sm, 0, S, 0, d_work, L, scratch);
    CK(hipMemcpy(r0.data(), scratch, sizeof(double) * ns, hipMemcpyDeviceToHost));
    CK(hipMemset(scratch, 0, sizeof(double) * ns));
    hipLaunchKernelGGL(k_dense_a, dim3((unsigned)work.size()), dim3(512), sizeof(double) * 2 * KBA * LDB, 0, S, 0, d_work, L, scratch, (const double*)zeros);
    CK(hipMemcpy(r1.data(), scratch, sizeof(double) * ns, hipMemcpyDeviceToHost));
    double mx = 0, ref = 0;
    for (size_t i = 0; i < ns; ++i) { mx = std::max(mx, std::fabs(r0[i] - r1[i])); ref = std::max(ref, std::fabs(r0[i])); }
    printf("k_dense_a vs k_dense<16>: max |difference| of the slabs %.3g (largest entry %.3g)\n", mx, ref);
  }
  return 0;
}
