// Size of the host thread teams (OpenMP regions of the analysis, std::thread pools of the plan).
//
// The CPUs this process may actually use are the minimum of the affinity mask and the cgroup CPU quota: a container
// that sees 256 logical CPUs but is limited to 16 CPUs' worth of time runs a 256-thread team SLOWER than a
// 16-thread one (the quota throttles the whole group inside barriers; measured with tools/omp_probe.cpp on the
// MI355X host: no gain beyond 16 threads, gathers 5x slower at 256).  OMP_NUM_THREADS / SCILMM_HOST_THREADS override.
#pragma once
#include <sched.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <thread>

namespace scilmm {

inline int host_threads() {
  static const int cached = [] {
    if (const char* e = getenv("SCILMM_HOST_THREADS"))
      if (atoi(e) > 0) return atoi(e);
    if (const char* e = getenv("OMP_NUM_THREADS"))
      if (atoi(e) > 0) return atoi(e);
    int nt = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) nt = std::min(nt, std::max(1, CPU_COUNT(&set)));
    // cgroup v2: "<quota> <period>" or "max <period>"; cgroup v1: two files
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
      long long q = 0, p = 0;
      if (fscanf(f, "%lld %lld", &q, &p) == 2 && q > 0 && p > 0) nt = std::min<long long>(nt, std::max<long long>(1, (q + p - 1) / p));
      fclose(f);
    } else {
      long long q = -1, p = -1;
      if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (fscanf(g, "%lld", &q) != 1) q = -1; fclose(g); }
      if (FILE* g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(g, "%lld", &p) != 1) p = -1; fclose(g); }
      if (q > 0 && p > 0) nt = std::min<long long>(nt, std::max<long long>(1, (q + p - 1) / p));
    }
    return nt;
  }();
  return cached;
}

}  // namespace scilmm

#ifdef _OPENMP
#include <omp.h>
namespace scilmm {
// call at the top of every C-ABI entry point that opens OpenMP regions (the team size is a per-thread setting)
inline void use_host_threads() { omp_set_num_threads(host_threads()); }
}  // namespace scilmm
#else
namespace scilmm {
inline void use_host_threads() {}
}  // namespace scilmm
#endif
