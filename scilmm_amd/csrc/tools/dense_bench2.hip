// Kernel-tuning harness, round 3: k_dense_a against k_dense_b (csrc/kernels.hip.h) ALONE on a synthetic dense tail -- one
// level's early update (target panel j of a T-panel dense lower-triangular matrix, all tile pairs, K split into segments
// of `per` descendants), no other stream, no host logic.  Prints the sustained TFLOP/s of the launch for zero and random
// operands (zeros read HIGH: zero MFMA operands raise the clock) and checks that both kernels write the same bits.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 [-DSCILMM_DENSE_B_SG=0] [-DSCILMM_DENSE_B_X4=1] dense_bench2.hip -o dense_bench2
//   ./dense_bench2 [T] [j] [panels per item]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "retired_kernels.hip.h"
using namespace scilmm;

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
  const int wlast = argc > 4 ? atoi(argv[4]) : NB;     // width of every 7th descendant (short chunks: < 128, not a multiple of 64)
  std::vector<int32_t> sn_start(T + 1, 0);
  for (int k = 0; k < T; ++k) sn_start[k + 1] = sn_start[k] + ((k % 7 == 3) ? wlast : NB);
  const int n = sn_start[T];
  std::vector<int64_t> sn_loff(T + 1, 0);
  for (int k = 0; k < T; ++k) sn_loff[k + 1] = sn_loff[k] + (((int64_t)(n - sn_start[k]) * (sn_start[k + 1] - sn_start[k]) + 1) & ~(int64_t)1);
  double* L;
  const size_t nL = (size_t)sn_loff[T] + 4 * (size_t)n + 4 * NB;
  CK(hipMalloc(&L, sizeof(double) * nL));
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
  const int mj = n - sn_start[j];
  const int ntl = (mj + TM - 1) / TM;  // tiles of panel j
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
  CK(hipFuncSetAttribute((const void*)k_dense_a, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_dense_b, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_dense_f, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_dense32, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  // tile-major fp32 shadow (k_dense_t): per panel ceil-blocks of 128 tail rows x w columns
  std::vector<int64_t> h_soff(T + 1, 0);
  for (int d = 0; d < T; ++d) {
    const int64_t nb = ((int64_t)(n - 1) >> 7) - (sn_start[d] >> 7) + 1;
    h_soff[d + 1] = h_soff[d] + nb * (sn_start[d + 1] - sn_start[d]) * 128;
  }
  float* S32;
  int64_t* d_soff;
  CK(hipMalloc(&S32, sizeof(float) * ((size_t)h_soff[T] + 65536)));
  CK(hipMemset(S32, 0, sizeof(float) * ((size_t)h_soff[T] + 65536)));
  CK(hipMalloc(&d_soff, sizeof(int64_t) * (T + 1)));
  CK(hipMemcpy(d_soff, h_soff.data(), sizeof(int64_t) * (T + 1), hipMemcpyHostToDevice));
  CK(hipFuncSetAttribute((const void*)k_dense_t, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_dense_w, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_dense_h, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  CK(hipFuncSetAttribute((const void*)k_dense_q, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  float* L32;
  CK(hipMalloc(&L32, sizeof(float) * ((size_t)nL + 16384)));
  CK(hipMemset(L32, 0, sizeof(float) * ((size_t)nL + 16384)));
  CK(hipFuncSetAttribute((const void*)k_dense_s, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
  double* zeros;
  CK(hipMalloc(&zeros, 2048));
  CK(hipMemset(zeros, 0, 2048));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double flops = 0;
  for (auto& w : work)
    for (int d = w.k0; d < w.k1; ++d) flops += 2.0 * std::min(w.ntiles * TM, mj - w.ti0 * TM) * (double)(sn_start[j + 1] - sn_start[j]) * (sn_start[d + 1] - sn_start[d]);
  printf("T=%d panels (every 7th %d wide), target %d: %zu items (%d descendants each), %.3f TFLOP per launch; k_dense_b: SG=%d X4=%d\n", T, wlast, j,
         work.size(), per, flops / 1e12, SCILMM_DENSE_B_SG, 0);
  const size_t smb = sizeof(double) * 2 * KBA * LDB;
  for (int fill : {0, 1})
  for (int which : {1, 8, 3, 7}) {
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, L, nL, fill);
    for (int rep = 0; rep < 4; ++rep) {
      hipEventRecord(e0);
      if (which == 0) hipLaunchKernelGGL(k_dense_a, dim3((unsigned)work.size()), dim3(512), smb, 0, S, 0, d_work, L, scratch, (const double*)zeros);
      else if (which == 1) hipLaunchKernelGGL(k_dense_b, dim3((unsigned)work.size()), dim3(512), smb, 0, S, 0, d_work, L, scratch, (const double*)zeros);
      else if (which == 2) hipLaunchKernelGGL(k_dense_f, dim3((unsigned)work.size()), dim3(512), dense_f_lds, 0, S, 0, d_work, L, scratch, (const double*)zeros);
      else if (which == 8) hipLaunchKernelGGL(k_dense_q, dim3(2 * (unsigned)work.size()), dim3(512), dense_q_lds, 0, S, 0, d_work, L, scratch, (const double*)zeros);
      else if (which == 7) {
        if (rep == 0) hipLaunchKernelGGL(k_shadow, dim3(8192), dim3(256), 0, 0, (const double*)L, L32, (int64_t)nL);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_dense_h, dim3(2 * (unsigned)work.size()), dim3(512), dense_h_lds, 0, S, 0, d_work, L, (const float*)L32, (int64_t)0, scratch, (const float*)zeros);
      }
      else if (which == 6) {
        if (rep == 0) hipLaunchKernelGGL(k_shadow, dim3(8192), dim3(256), 0, 0, (const double*)L, L32, (int64_t)nL);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_dense_w, dim3((unsigned)work.size()), dim3(256), dense_w_lds, 0, S, 0, d_work, L, (const float*)L32, (int64_t)0, scratch, (const float*)zeros);
      }
      else if (which == 5) {
        if (rep == 0)
          for (int d = 0; d < T; ++d)
            hipLaunchKernelGGL(k_shadow_t, dim3(2048), dim3(256), 0, 0, (const double*)(L + sn_loff[d]), S32 + h_soff[d], sn_start[d], n - sn_start[d], sn_start[d + 1] - sn_start[d]);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_dense_t, dim3((unsigned)work.size()), dim3(512), dense_s_lds + 4096, 0, S, 0, d_work, L, (const float*)S32, (const int64_t*)d_soff, scratch, (const float*)zeros);
      }
      else if (which == 4) {
        if (rep == 0) hipLaunchKernelGGL(k_shadow, dim3(8192), dim3(256), 0, 0, (const double*)L, L32, (int64_t)nL);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_dense_s, dim3((unsigned)work.size()), dim3(512), dense_s_lds, 0, S, 0, d_work, L, (const float*)L32, (int64_t)0, scratch, (const float*)zeros);
      }
      else hipLaunchKernelGGL(k_dense32, dim3((unsigned)work.size()), dim3(512), sizeof(float) * (size_t)(2 * KC * LDA2F + 2 * KC * LDBF), 0, S, 0, d_work, L, scratch);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      CK(hipGetLastError());
      if (rep) printf("%s, %s operands: %.3f ms -> %.2f TFLOP/s\n", which == 0 ? "k_dense_a" : which == 1 ? "k_dense_b" : which == 2 ? "k_dense_f (fp32 products)" : which == 4 ? "k_dense_s (fp32 shadow operands)" : which == 5 ? "k_dense_t (tile-major fp32 shadow)" : which == 6 ? "k_dense_w (one wave per SIMD, fp32 shadow)" : which == 7 ? "k_dense_h (32 x 64 wave tiles, fp32 shadow)" : which == 8 ? "k_dense_q (fp64, 32 x 64 wave tiles, two workgroups per CU)" : "k_dense32 (round 2)", fill ? "random" : "zero", ms, flops / ms / 1e9);
    }
  }
  {
    const size_t ns = (size_t)slot * TM * NB;
    std::vector<double> r0(ns), r1(ns);
    // rows past a partial tile's edge hold whatever the kernel's clamped loads produced (never read by k_reduce): masked
    std::vector<uint8_t> live(ns, 0);
    for (auto& w : work)
      for (int h = 0; h < w.ntiles; ++h) {
        const int sl = h ? w.slot1 : w.slot0;
        const int nr = std::min(TM, mj - (w.ti0 + h) * TM);
        for (int jc = 0; jc < NB; ++jc)
          for (int i = 0; i < nr; ++i) live[(size_t)sl * TM * NB + (size_t)jc * TM + i] = 1;
      }
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, L, nL, 1);
    CK(hipMemset(scratch, 0, sizeof(double) * ns));
    hipLaunchKernelGGL(k_dense_a, dim3((unsigned)work.size()), dim3(512), smb, 0, S, 0, d_work, L, scratch, (const double*)zeros);
    CK(hipMemcpy(r0.data(), scratch, sizeof(double) * ns, hipMemcpyDeviceToHost));
    CK(hipMemset(scratch, 0, sizeof(double) * ns));
    hipLaunchKernelGGL(k_dense_b, dim3((unsigned)work.size()), dim3(512), smb, 0, S, 0, d_work, L, scratch, (const double*)zeros);
    CK(hipMemcpy(r1.data(), scratch, sizeof(double) * ns, hipMemcpyDeviceToHost));
    double mx = 0, ref = 0;
    for (size_t i = 0; i < ns; ++i) if (live[i]) { mx = std::max(mx, std::fabs(r0[i] - r1[i])); ref = std::max(ref, std::fabs(r0[i])); }
    printf("k_dense_b vs k_dense_a: max |difference| of the slabs %.3g (largest entry %.3g)\n", mx, ref);
    for (int v = 0; v < 7; ++v) {
      CK(hipMemset(scratch, 0, sizeof(double) * ns));
      if (v == 6) {
        hipLaunchKernelGGL(k_dense_q, dim3(2 * (unsigned)work.size()), dim3(512), dense_q_lds, 0, S, 0, d_work, L, scratch, (const double*)zeros);
      } else if (v == 5) {
        hipLaunchKernelGGL(k_shadow, dim3(8192), dim3(256), 0, 0, (const double*)L, L32, (int64_t)nL);
        hipLaunchKernelGGL(k_dense_h, dim3(2 * (unsigned)work.size()), dim3(512), dense_h_lds, 0, S, 0, d_work, L, (const float*)L32, (int64_t)0, scratch, (const float*)zeros);
      } else if (v == 4) {
        hipLaunchKernelGGL(k_shadow, dim3(8192), dim3(256), 0, 0, (const double*)L, L32, (int64_t)nL);
        hipLaunchKernelGGL(k_dense_w, dim3((unsigned)work.size()), dim3(256), dense_w_lds, 0, S, 0, d_work, L, (const float*)L32, (int64_t)0, scratch, (const float*)zeros);
      } else if (v == 3) {
        for (int d = 0; d < T; ++d)
          hipLaunchKernelGGL(k_shadow_t, dim3(2048), dim3(256), 0, 0, (const double*)(L + sn_loff[d]), S32 + h_soff[d], sn_start[d], n - sn_start[d], sn_start[d + 1] - sn_start[d]);
        hipLaunchKernelGGL(k_dense_t, dim3((unsigned)work.size()), dim3(512), dense_s_lds + 4096, 0, S, 0, d_work, L, (const float*)S32, (const int64_t*)d_soff, scratch, (const float*)zeros);
      } else if (v == 2) {
        hipLaunchKernelGGL(k_shadow, dim3(8192), dim3(256), 0, 0, (const double*)L, L32, (int64_t)nL);
        hipLaunchKernelGGL(k_dense_s, dim3((unsigned)work.size()), dim3(512), dense_s_lds, 0, S, 0, d_work, L, (const float*)L32, (int64_t)0, scratch, (const float*)zeros);
      } else if (v == 0) hipLaunchKernelGGL(k_dense_f, dim3((unsigned)work.size()), dim3(512), dense_f_lds, 0, S, 0, d_work, L, scratch, (const double*)zeros);
      else hipLaunchKernelGGL(k_dense32, dim3((unsigned)work.size()), dim3(512), sizeof(float) * (size_t)(2 * KC * LDA2F + 2 * KC * LDBF), 0, S, 0, d_work, L, scratch);
      CK(hipMemcpy(r1.data(), scratch, sizeof(double) * ns, hipMemcpyDeviceToHost));
      mx = 0;
      for (size_t i = 0; i < ns; ++i) if (live[i]) mx = std::max(mx, std::fabs(r0[i] - r1[i]));
      printf("%s vs fp64: max |difference| of the slabs %.3g (largest entry %.3g)\n", v == 6 ? "k_dense_q (fp64, 32 x 64 wave tiles)" : v == 5 ? "k_dense_h (32 x 64 wave tiles, fp32 shadow)" : v == 4 ? "k_dense_w (one wave per SIMD, fp32 shadow)" : v == 3 ? "k_dense_t (tile-major fp32 shadow)" : v == 2 ? "k_dense_s (fp32 shadow operands)" : v ? "k_dense32 (round 2)" : "k_dense_f (fp32 products)", mx, ref);
    }
  }
  return 0;
}
