// How many host threads actually run in parallel on this box?  A compute-bound loop and a random-gather loop, timed
// for several team sizes (tuning aid for the analysis phase; not part of the library).
#include <omp.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>
int main() {
  const size_t N = size_t(1) << 28;  // 1 GiB of int32
  std::vector<int32_t> a(N);
#pragma omp parallel for
  for (size_t i = 0; i < N; ++i) a[i] = (int32_t)((i * 2654435761u) & (N - 1));
  for (int th : {1, 4, 8, 16, 32, 64, 128, 256}) {
    auto t0 = std::chrono::steady_clock::now();
    double s = 0;
#pragma omp parallel for num_threads(th) reduction(+ : s)
    for (int q = 0; q < 4096; ++q) {
      double x = q;
      for (int it = 0; it < 2000000; ++it) x = x * 1.0000001 + 1e-9;
      s += x;
    }
    auto t1 = std::chrono::steady_clock::now();
    int64_t g = 0;
#pragma omp parallel for num_threads(th) reduction(+ : g)
    for (size_t i = 0; i < N / 8; ++i) g += a[(size_t)a[i * 8]];
    auto t2 = std::chrono::steady_clock::now();
    printf("threads %3d: compute %.3f s, gather %.3f s (%g %lld)\n", th, std::chrono::duration<double>(t1 - t0).count(),
           std::chrono::duration<double>(t2 - t1).count(), s, (long long)g);
    fflush(stdout);
  }
  return 0;
}
