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

// hipFuncAttributeMaxDynamicSharedMemorySize is a PER-DEVICE setting of a kernel: asked for once per (kernel instantiation, device) and
// its return code checked.  `done` = the instantiation's bit mask of devices served (a function-local static at the launch site).
inline bool lds_opt_in(const void* fn, int bytes, unsigned long long* done, const char* what) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const unsigned long long bit = 1ull << (dev & 63);
  if (__atomic_load_n(done, __ATOMIC_ACQUIRE) & bit) return true;
  const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) {
    set_error("%s: %d bytes of dynamic LDS refused on device %d: %s", what, bytes, dev, hipGetErrorString(e));
    return false;
  }
  __atomic_fetch_or(done, bit, __ATOMIC_RELEASE);
  return true;
}
#define DFL_LDS_OPT_IN(kernel, bytes, what)                                                                       \
  {                                                                                                               \
    static unsigned long long lds_done_ = 0;                                                                      \
    if (!::dfl::lds_opt_in(reinterpret_cast<const void*>(kernel), (int)(bytes), &lds_done_, what)) return DFL_ERR_LAUNCH; \
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

// ---- split-bf16 products (DFL math modes 1 "bf16x3" and 2 "bf16x6"; see conv_gemm.hip) ------------------------------
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

// x = p[0] + p[1] (+ p[2]) + O(2^-8NP |x|): successive bf16 roundings of the residual; 4 values -> 4 bf16 per part
template <int NP>
__device__ __forceinline__ void split_bf16(const float4 v, uint2* parts) {
  f32x2_t a = {v.x, v.y}, b = {v.z, v.w};
#pragma unroll
  for (int q = 0; q < NP; ++q) {
    const bf16x2_t h0 = __builtin_convertvector(a, bf16x2_t), h1 = __builtin_convertvector(b, bf16x2_t);
    parts[q] = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
    if (q + 1 < NP) {
      a -= __builtin_convertvector(h0, f32x2_t);
      b -= __builtin_convertvector(h1, f32x2_t);
    }
  }
}

// Product arithmetic of the fast GEMM paths: 0 = fp32 matrix instructions, 1 = bf16x3, 2 = bf16x6 (conv only; the
// weight-gradient kernel then stays fp32).  Process-wide (dfl_set_math_mode / DFL_MATH).
int math_mode();

// wave64 butterfly step across the two 32-lane halves
__device__ __forceinline__ float xor32(float v) { return __shfl_xor(v, 32, 64); }

// Row of accumulator register r (0..15) of a 32x32 MFMA tile for this lane; the column is lane & 31.
__device__ __forceinline__ int mfma32_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ---- "live" BatchNorm statistics (dfl_conv_args.stat_totals / in_tot / add_tot, include/dfl_hip.h) ----------------------
// One addend of a workgroup into the totals [DFL_BN_R][2][C]: the hardware fp64 atomic, no return value (fire and forget)
__device__ __forceinline__ void bn_live_add(double* tot, int row, int which, int C, int c, float v) {
  (void)__builtin_amdgcn_global_atomic_fadd_f64(tot + ((int64_t)((row & (DFL_BN_R - 1)) * 2 + which)) * C + c, (double)v);
}
// scale / shift (and mean, 1/std) of channel c from the totals: the arithmetic of bn_finalize_kernel, word for word
__device__ __forceinline__ void bn_live_affine(const double* tot, const float* gamma, const float* beta, double count, float eps,
                                               int C, int c, float* scale, float* shift, double* mean_out = nullptr,
                                               double* var_out = nullptr) {
  double s1 = 0.0, s2 = 0.0;
#pragma unroll
  for (int r = 0; r < DFL_BN_R; ++r) {
    s1 += tot[(int64_t)(r * 2 + 0) * C + c];
    s2 += tot[(int64_t)(r * 2 + 1) * C + c];
  }
  const double mean = s1 / count;
  double var = s2 / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const double invstd = 1.0 / sqrt(var + (double)eps);
  *scale = (float)((double)gamma[c] * invstd);
  *shift = (float)((double)beta[c] - mean * (double)gamma[c] * invstd);
  if (mean_out != nullptr) *mean_out = mean;
  if (var_out != nullptr) *var_out = var;
}

// A, B, C of  d(pre-activation) = [r > 0] * (A dy + B r + C)  for channel c from the totals (sum dy, sum dy*r): the arithmetic of
// bn_bwd_finalize_kernel (training mode), word for word
__device__ __forceinline__ void bn_live_coef(const double* tot, const float* gamma, const float* save_mean, const float* save_invstd,
                                             double count, int C, int c, float* A, float* B, float* Cc) {
  double sdy = 0.0, sdyr = 0.0;
#pragma unroll
  for (int r = 0; r < DFL_BN_R; ++r) {
    sdy += tot[(int64_t)(r * 2 + 0) * C + c];
    sdyr += tot[(int64_t)(r * 2 + 1) * C + c];
  }
  const double mean = (double)save_mean[c], invstd = (double)save_invstd[c], g = (double)gamma[c];
  const double sdyx = invstd * (sdyr - mean * sdy);
  const double s = g * invstd;
  const double c1 = sdy / count, c2 = sdyx / count;
  *A = (float)s;
  *B = (float)(-s * c2 * invstd);
  *Cc = (float)(-s * c1 + s * c2 * invstd * mean);
}

}  // namespace dfl
