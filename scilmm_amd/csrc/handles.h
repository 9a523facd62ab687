// Opaque handle layout shared by the host-only and the device translation units.
#pragma once
#include <string>

#include "symbolic.h"

struct scilmm_symbolic {
  scilmm::Symbolic* S = nullptr;
  std::string err;
  void* device = nullptr;              // owned by engine.hip
  void (*device_free)(void*) = nullptr;
};
