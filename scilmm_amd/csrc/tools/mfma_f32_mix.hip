// Microbenchmark, round 3: what the fp32 matrix pipe (v_mfma_f32_16x16x4_f32, 64 FLOP/clk/SIMD) loses to the other
// instructions of a "fp32 products, fp64 sums" tile loop.  512-thread workgroups, one per CU (two waves per SIMD), each wave
// 16 independent fp32 accumulators of 4 registers (k_dense32 / k_dense_f's tile), 16 MFMAs per k-step.  Variants:
//   0  MFMAs alone (operands in registers)
//   1  + the 8 B and 2 A fragments of every k-step read from LDS as fp32 (ds_read_b32)
//   2  + the 8 B fragments read as fp64 and converted (v_cvt_f32_f64), A from registers
//   3  variant 1 + a fold of the 64 fp32 sums into fp64 accumulators every 16 k-steps (all waves at the same k-step)
//   4  variant 1 + the fold every 4 k-steps (k_dense32's period)
//   5  variant 3 with the two waves of a SIMD folding 8 k-steps apart
//   hipcc -O3 --offload-arch=gfx950 mfma_f32_mix.hip -o mfma_f32_mix && ./mfma_f32_mix
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int KR = 64, LDBF = 144, LDBD = 144;

template <int V>
__global__ __launch_bounds__(512, 1) void k_mix(double* out, int iters) {
  __shared__ float Bf[KR * LDBF];
  __shared__ float Af[KR * 272];
  __shared__ double Bd[32 * LDBD];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  for (int i = tid; i < KR * LDBF; i += 512) { Bf[i] = 1e-3f * (i % 5); if (i < 32 * LDBD) Bd[i] = 1e-3 * (i % 5); }
  for (int i = tid; i < KR * 272; i += 512) Af[i] = 1e-3f * (i % 7);
  __syncthreads();
  const int li = lane & 15, lk = lane >> 4;
  f4 c[8][2];
  d4 acc[8][2];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) { c[a][b] = (f4){0, 0, 0, 0}; acc[a][b] = (d4){0, 0, 0, 0}; }
  float a0 = 1e-3f * lane, a1 = 2e-3f * lane;
  float bb[8];
#pragma unroll
  for (int jb = 0; jb < 8; ++jb) bb[jb] = 1e-3f * (lane + jb);
  const int phase = V == 5 ? ((wv >> 2) & 1) * 8 : 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int k4 = 4 * t;
      if (V == 1 || V >= 3) {
        a0 = Af[(k4 + lk) * 272 + 32 * wv + li];
        a1 = Af[(k4 + lk) * 272 + 32 * wv + 16 + li];
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) bb[jb] = Bf[(k4 + lk) * LDBF + 16 * jb + li];
      } else if (V == 2) {
#pragma unroll
        for (int jb = 0; jb < 8; ++jb) bb[jb] = (float)Bd[((k4 + lk) & 31) * LDBD + 16 * jb + li];
      }
#pragma unroll
      for (int jb = 0; jb < 8; ++jb) {
        c[jb][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb[jb], a0, c[jb][0], 0, 0, 0);
        c[jb][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(bb[jb], a1, c[jb][1], 0, 0, 0);
      }
      const bool fold = (V == 3 && t == 15) || (V == 4 && (t & 3) == 3) || (V == 5 && t == (phase == 0 ? 15 : 7));
      if (V >= 3 && fold) {
#pragma unroll
        for (int jb = 0; jb < 8; ++jb)
#pragma unroll
          for (int ib = 0; ib < 2; ++ib) {
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[jb][ib][r] += (double)c[jb][ib][r];
            c[jb][ib] = (f4){0, 0, 0, 0};
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  double s = 0;
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) s += acc[a][b][r] + c[a][b][r];
  out[blockIdx.x * 512 + tid] = s;
}

template <int V>
void run(double* d, const char* what) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 2000, grid = 256;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mix<V>, dim3(grid), dim3(512), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 8 * iters * 16 * 16 * 2048.0;
    if (rep == 2) printf("variant %d (%s): %.3f ms -> %.1f TFLOP/s\n", V, what, ms, flops / ms / 1e9);
  }
}

int main() {
  double* d;
  (void)hipMalloc(&d, sizeof(double) * 512 * 256);
  run<0>(d, "MFMAs alone");
  run<1>(d, "+ fp32 fragments from LDS");
  run<2>(d, "+ fp64 B fragments from LDS, converted");
  run<3>(d, "fp32 fragments + fold every 16 k-steps");
  run<4>(d, "fp32 fragments + fold every 4 k-steps");
  run<5>(d, "fp32 fragments + fold every 16 k-steps, wave pairs 8 k-steps apart");
  return 0;
}
