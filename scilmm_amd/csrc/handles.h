// Opaque handle layout shared by the host-only and the device translation units.
#pragma once
#include <string>

#include "symbolic.h"

struct scilmm_symbolic {
  scilmm::Symbolic* S = nullptr;
  std::string err;
  void* device = nullptr;              // owned by engine.hip
  void (*device_free)(void*) = nullptr;
  // multi-GPU (scilmm_dist_init): this process is rank `rank` of `world`; collectives go through comm_fn, issued on
  // comm_stream (a hipStream_t owned by the caller, e.g. the raw handle of a torch stream)
  int32_t rank = 0, world = 1;
  void* comm_stream = nullptr;
  int (*comm_fn)(void* ctx, int32_t op, int32_t buffer, int64_t offset, int64_t count, int32_t root) = nullptr;
  void* comm_ctx = nullptr;
  bool maps_released = false;          // scilmm_symbolic_release_host_maps: no further value uploads
};
