// Shared helpers for the gfx950 kernels of libdfl_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/dfl_hip.h"

namespace dfl {

void set_error(const char* fmt, ...);

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return DFL_ERR_LAUNCH;
  }
  return DFL_OK;
}

#define DFL_REQUIRE(cond, ...)              \
  do {                                      \
    if (!(cond)) {                          \
      ::dfl::set_error(__VA_ARGS__);        \
      return DFL_ERR_INVALID_ARG;           \
    }                                       \
  } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

typedef float f32x16 __attribute__((ext_vector_type(16)));

// wave64 butterfly step across the two 32-lane halves
__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }

// Row of accumulator register r (0..15) of a 32x32 MFMA tile for this lane; the column is lane & 31.
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

}  // namespace dfl
