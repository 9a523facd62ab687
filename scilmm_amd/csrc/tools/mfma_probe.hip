// Probe: operand / result lane mapping of v_mfma_f64_4x4x4f64 (4 blocks of 4x4x4) on this GPU, found by one-hot
// inputs: for every (A lane la, B lane lb) the lanes of D that receive the product.  Prints, per lane, which
// (block, row i, k) its A operand is, which (block, k, col j) its B operand is and which (block, i, j) its result is.
//   hipcc -O3 --offload-arch=gfx950 mfma_probe.hip -o mfma_probe && ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void k_probe(int* out) {  // one wave; out[la*64+lb] = bitmask-free: store lane id of the nonzero result or -1/-2
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      const unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) out[la * 64 + lb] = m == 0 ? -1 : (__popcll(m) == 1 ? __builtin_ctzll(m) : -2);
    }
}

int main() {
  int* d;
  (void)hipMalloc(&d, sizeof(int) * 4096);
  hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d);
  std::vector<int> h(4096);
  (void)hipMemcpy(h.data(), d, sizeof(int) * 4096, hipMemcpyDeviceToHost);
  // for each A lane: the set of B lanes it pairs with and where the result lands
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d pairs with B lanes ->D lane:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (h[la * 64 + lb] != -1) printf(" %d->%d", lb, h[la * 64 + lb]);
    printf("\n");
  }
  return 0;
}
