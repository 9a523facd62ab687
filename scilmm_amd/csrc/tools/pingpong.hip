// Cross-workgroup signalling latency on MI355X: two workgroups (forced onto different XCDs by using block ids
// 0 and 1 of a 2-block grid: consecutive ids go to consecutive XCDs) bounce an agent-scope flag N times.
// Build: hipcc --offload-arch=gfx950 -O3 pingpong.hip -o pingpong ; run: ./pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_pp(int* flag, int n, unsigned long long* out, int with_data, double* data) {
  const int me = blockIdx.x;
  unsigned long long t0 = wall_clock64();
  for (int it = 0; it < n; ++it) {
    const int want = 2 * it + me;  // block 0 waits for even values ... block 0 starts
    if (threadIdx.x == 0) {
      if (me == 0 && it == 0) {
      } else {
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {}
      }
    }
    __syncthreads();
    if (with_data) {
      // producer side of the next message: 8 write-through stores per thread, drained before the flag
      for (int u = 0; u < 8; ++u)
        __hip_atomic_store(&data[(me * 8 + u) * 256 + threadIdx.x], (double)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    if (threadIdx.x == 0) __hip_atomic_store(flag, want + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (threadIdx.x == 0) out[me] = wall_clock64() - t0;
}
int main() {
  int* flag; unsigned long long* out; double* data;
  hipMalloc(&flag, 4); hipMalloc(&out, 16); hipMalloc(&data, 8 * 16 * 256);
  for (int wd = 0; wd < 2; ++wd) {
    hipMemset(flag, 0, 4);
    const int n = 2000;
    hipLaunchKernelGGL(k_pp, dim3(2), dim3(256), 0, 0, flag, n, out, wd, data);
    hipDeviceSynchronize();
    unsigned long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
    printf("with_data=%d: %d round trips in %.1f us -> one-way hop %.2f us\n", wd, n, h[0] / 100.0, h[0] / 100.0 / (2.0 * n));
  }
  return 0;
}
