// Shared host-side helpers for libtonal_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/tonal_hip.h"

namespace tl {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return TL_ELAUNCH;
  }
  return TL_OK;
}

#define TL_REQUIRE(cond, ...)                \
  do {                                       \
    if (!(cond)) {                           \
      tl::set_error(__VA_ARGS__);            \
      return TL_EINVAL;                      \
    }                                        \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float lrelu(float z, float slope) { return z > 0.f ? z : z * slope; }

}  // namespace tl
