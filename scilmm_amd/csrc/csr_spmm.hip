// Y = A X for a CSR matrix and a row-major n x r block, on the device, WITHOUT a symbolic analysis: the products of the
// Haseman-Elston standard error (reference scilmm/SparseCholesky.py:259-278: four SciPy products of n x 100 blocks per matrix
// pair -- the estimator the reference's authors prefer above 250k individuals, README.md:63) and of any caller that holds a
// relationship matrix but no factor.  (With an analysis at hand scilmm_spmm_dev streams the slot-ordered lower triangle
// instead: half the bytes.)
// HBM / L2-bound gather: one wavefront per row; the wave loads 64 (column, value) pairs at a time, broadcasts them lane by
// lane and every lane accumulates its two right-hand-side columns -- a row of X is 8 r contiguous bytes, so each
// wave-instruction reads one 512-byte piece of it.  Sums run in storage order of the row: deterministic.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <string>

#include "../../include/scilmm_hip.h"

namespace {

__global__ __launch_bounds__(256) void k_csr_spmm(int32_t n, const int64_t* __restrict__ indptr, const int32_t* __restrict__ indices,
                                                  const double* __restrict__ data, const double* __restrict__ X, int32_t r,
                                                  double* __restrict__ Y) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * (blockDim.x >> 6);
  const int c0 = (int)blockIdx.y * 128 + lane, c1 = c0 + 64;  // this lane's two columns of the 128-column window
  for (int64_t i = wave; i < n; i += nwaves) {
    const int64_t rs = indptr[i], re = indptr[i + 1];
    double acc0 = 0.0, acc1 = 0.0;
    for (int64_t base = rs; base < re; base += 64) {
      const int64_t t = base + lane;
      const int32_t j = t < re ? indices[t] : 0;
      const double a = t < re ? data[t] : 0.0;
      const int cnt = (int)std::min<int64_t>(64, re - base);
      for (int q = 0; q < cnt; ++q) {
        const int32_t jq = __shfl(j, q, 64);
        const double aq = __shfl(a, q, 64);
        const double* xr = X + (int64_t)jq * r;
        if (c0 < r) acc0 += aq * xr[c0];
        if (c1 < r) acc1 += aq * xr[c1];
      }
    }
    if (c0 < r) Y[i * (int64_t)r + c0] = acc0;
    if (c1 < r) Y[i * (int64_t)r + c1] = acc1;
  }
}

thread_local std::string g_spmm_err;

}  // namespace

extern "C" {

const char* scilmm_csr_spmm_error(void) { return g_spmm_err.c_str(); }

int scilmm_csr_spmm_dev(int32_t n, const int64_t* d_indptr, const int32_t* d_indices, const double* d_data, const double* d_X, int32_t r,
                        double* d_Y, void* stream) {
  if (n < 0 || r <= 0 || !d_indptr || !d_X || !d_Y || d_X == d_Y) return SCILMM_ERR_ARG;
  if (n == 0) return SCILMM_OK;
  if (!d_indices || !d_data) return SCILMM_ERR_ARG;
  const unsigned bx = (unsigned)std::min<int64_t>(((int64_t)n + 3) / 4, 256 * 64);
  const unsigned by = (unsigned)((r + 127) / 128);
  hipLaunchKernelGGL(k_csr_spmm, dim3(bx, by), dim3(256), 0, (hipStream_t)stream, n, d_indptr, d_indices, d_data, d_X, r, d_Y);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    g_spmm_err = std::string("k_csr_spmm launch: ") + hipGetErrorString(e);
    return SCILMM_ERR_DEVICE;
  }
  return SCILMM_OK;
}

}  // extern "C"
