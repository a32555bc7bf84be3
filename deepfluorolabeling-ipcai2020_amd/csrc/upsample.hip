// Bilinear x2 up-sampling of NHWC tensors and its adjoint -- up_mode='upsample' of the reference
// (train_test_code/unet.py:242-244: nn.Sequential(nn.Upsample(mode='bilinear', scale_factor=2), nn.Conv2d(in, out, 1))).
// The 1x1 convolution is pointwise and the interpolation weights of a pixel sum to one, so conv1x1(upsample(x)) =
// upsample(conv1x1(x)) with the bias added once: the plan runs the 1x1 convolution on the SMALL grid (a quarter of the
// products) and these kernels spread its result over the 2H x 2W grid, straight into the up half of the concat buffer.
//
// torch semantics (align_corners=False): source coordinate s = (o + 0.5) / 2 - 0.5, clamped at 0; i0 = floor(s),
// i1 = min(i0 + 1, n - 1); weights (1 - frac, frac).  I.e. output 2i takes (i-1: 0.25, i: 0.75), output 2i+1 takes
// (i: 0.75, i+1: 0.25), border rows / columns replicate.
//
// Both kernels stream: one thread per (pixel, 8 bf16 or 4 fp32 channels), 16-byte accesses, fp32 arithmetic.
#include "common.h"

namespace dfl {

typedef unsigned int uu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void up_coords(int o, int n, int* i0, int* i1, float* w0, float* w1) {
  float s = 0.5f * ((float)o + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  const int a = (int)s;
  *i0 = a;
  *i1 = a < n - 1 ? a + 1 : a;
  *w1 = s - (float)a;
  *w0 = 1.f - *w1;
}

// VEC: channels per thread (8 for bf16 tensors, 4 or 1 for fp32)
template <int VEC, bool BF>
__device__ __forceinline__ void up_load(const void* base, int64_t elem, float* v) {
  if constexpr (BF) {
    const uu32x4 w = *reinterpret_cast<const uu32x4*>(reinterpret_cast<const unsigned short*>(base) + elem);
    v[0] = __uint_as_float(w.x << 16); v[1] = __uint_as_float(w.x & 0xffff0000u);
    v[2] = __uint_as_float(w.y << 16); v[3] = __uint_as_float(w.y & 0xffff0000u);
    v[4] = __uint_as_float(w.z << 16); v[5] = __uint_as_float(w.z & 0xffff0000u);
    v[6] = __uint_as_float(w.w << 16); v[7] = __uint_as_float(w.w & 0xffff0000u);
  } else if constexpr (VEC == 4) {
    const float4 w = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(base) + elem);
    v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w;
  } else {
    v[0] = reinterpret_cast<const float*>(base)[elem];
  }
}

template <int VEC, bool BF>
__device__ __forceinline__ void up_store(void* base, int64_t elem, const float* v) {
  if constexpr (BF) {
    uu32x4 w;
    w.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[0], v[1]}, bf16x2_t));
    w.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[2], v[3]}, bf16x2_t));
    w.z = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[4], v[5]}, bf16x2_t));
    w.w = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[6], v[7]}, bf16x2_t));
    *reinterpret_cast<uu32x4*>(reinterpret_cast<unsigned short*>(base) + elem) = w;
  } else if constexpr (VEC == 4) {
    *reinterpret_cast<float4*>(reinterpret_cast<float*>(base) + elem) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
    reinterpret_cast<float*>(base)[elem] = v[0];
  }
}

// y[n, oy, ox, c] = sum of the (up to) four source pixels of (oy, ox)
template <int VEC, bool BF>
__global__ void __launch_bounds__(256) upsample2x_fwd_kernel(const dfl_upsample_args a, int64_t total_units, int cq) {
  const int Ho = 2 * a.H, Wo = 2 * a.W;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_units; i += stride) {
    const int u = (int)(i % cq);
    int64_t pix = i / cq;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    const int c = u * VEC;
    int y0, y1, x0, x1;
    float wy0, wy1, wx0, wx1;
    up_coords(oy, a.H, &y0, &y1, &wy0, &wy1);
    up_coords(ox, a.W, &x0, &x1, &wx0, &wx1);
    const int64_t r0 = ((int64_t)n * a.H + y0) * a.W, r1 = ((int64_t)n * a.H + y1) * a.W;
    float v00[VEC], v01[VEC], v10[VEC], v11[VEC], o[VEC];
    up_load<VEC, BF>(a.x, (r0 + x0) * a.ldx + c, v00);
    up_load<VEC, BF>(a.x, (r0 + x1) * a.ldx + c, v01);
    up_load<VEC, BF>(a.x, (r1 + x0) * a.ldx + c, v10);
    up_load<VEC, BF>(a.x, (r1 + x1) * a.ldx + c, v11);
#pragma unroll
    for (int e = 0; e < VEC; ++e) o[e] = wy0 * (wx0 * v00[e] + wx1 * v01[e]) + wy1 * (wx0 * v10[e] + wx1 * v11[e]);
    up_store<VEC, BF>(a.y, (((int64_t)n * Ho + oy) * Wo + ox) * a.ldy + c, o);
  }
}

// adjoint: x[n, iy, ix, c] (+)= sum over the outputs (oy, ox) that read (iy, ix) of their weight * y[n, oy, ox, c]
template <int VEC, bool BF>
__global__ void __launch_bounds__(256) upsample2x_bwd_kernel(const dfl_upsample_args a, int64_t total_units, int cq) {
  const int Ho = 2 * a.H, Wo = 2 * a.W;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_units; i += stride) {
    const int u = (int)(i % cq);
    int64_t pix = i / cq;
    const int ix = (int)(pix % a.W);
    pix /= a.W;
    const int iy = (int)(pix % a.H);
    const int n = (int)(pix / a.H);
    const int c = u * VEC;
    float acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = 0.f;
    for (int oy = 2 * iy - 1; oy <= 2 * iy + 2; ++oy) {
      if (oy < 0 || oy >= Ho) continue;
      int y0, y1;
      float wy0, wy1;
      up_coords(oy, a.H, &y0, &y1, &wy0, &wy1);
      const float wy = (y0 == iy ? wy0 : 0.f) + (y1 == iy ? wy1 : 0.f);
      if (wy == 0.f) continue;
      for (int ox = 2 * ix - 1; ox <= 2 * ix + 2; ++ox) {
        if (ox < 0 || ox >= Wo) continue;
        int x0, x1;
        float wx0, wx1;
        up_coords(ox, a.W, &x0, &x1, &wx0, &wx1);
        const float w = wy * ((x0 == ix ? wx0 : 0.f) + (x1 == ix ? wx1 : 0.f));
        if (w == 0.f) continue;
        float v[VEC];
        up_load<VEC, BF>(a.y, (((int64_t)n * Ho + oy) * Wo + ox) * a.ldy + c, v);
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = fmaf(w, v[e], acc[e]);
      }
    }
    const int64_t xo = (((int64_t)n * a.H + iy) * a.W + ix) * a.ldx + c;
    if (a.accumulate) {
      float o[VEC];
      up_load<VEC, BF>(a.x, xo, o);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc[e] += o[e];
    }
    up_store<VEC, BF>(const_cast<void*>(a.x), xo, acc);
  }
}

static unsigned up_grid(int64_t units) {
  int64_t b = ceil_div(units, 256);
  if (b > 8192) b = 8192;
  return (unsigned)(b < 1 ? 1 : b);
}

static int up_check(const dfl_upsample_args* a, const char* who, int* vec) {
  DFL_REQUIRE(a && a->x && a->y && a->N > 0 && a->H > 0 && a->W > 0 && a->C > 0, "%s: bad arguments", who);
  DFL_REQUIRE(a->ldx >= a->C && a->ldy >= a->C, "%s: pixel strides below the channel count", who);
  if (a->bf16) {
    DFL_REQUIRE(a->C % 8 == 0 && a->ldx % 8 == 0 && a->ldy % 8 == 0 && aligned16(a->x) && aligned16(a->y),
                "%s (bf16): C and pixel strides must be multiples of 8, tensors 16-byte aligned", who);
    *vec = 8;
  } else {
    *vec = (a->C % 4 == 0 && a->ldx % 4 == 0 && a->ldy % 4 == 0 && aligned16(a->x) && aligned16(a->y)) ? 4 : 1;
  }
  return DFL_OK;
}

}  // namespace dfl

extern "C" int dfl_upsample2x_fwd(const dfl_upsample_args* a, dfl_stream_t stream) {
  int vec;
  int rc = dfl::up_check(a, "dfl_upsample2x_fwd", &vec);
  if (rc != DFL_OK) return rc;
  const int cq = a->C / vec;
  const int64_t total = (int64_t)a->N * (2 * a->H) * (2 * a->W) * cq;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (vec == 8) hipLaunchKernelGGL((dfl::upsample2x_fwd_kernel<8, true>), dim3(dfl::up_grid(total)), dim3(256), 0, s, *a, total, cq);
  else if (vec == 4) hipLaunchKernelGGL((dfl::upsample2x_fwd_kernel<4, false>), dim3(dfl::up_grid(total)), dim3(256), 0, s, *a, total, cq);
  else hipLaunchKernelGGL((dfl::upsample2x_fwd_kernel<1, false>), dim3(dfl::up_grid(total)), dim3(256), 0, s, *a, total, cq);
  return dfl::check_launch("dfl_upsample2x_fwd");
}

extern "C" int dfl_upsample2x_bwd(const dfl_upsample_args* a, dfl_stream_t stream) {
  int vec;
  int rc = dfl::up_check(a, "dfl_upsample2x_bwd", &vec);
  if (rc != DFL_OK) return rc;
  const int cq = a->C / vec;
  const int64_t total = (int64_t)a->N * a->H * a->W * cq;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (vec == 8) hipLaunchKernelGGL((dfl::upsample2x_bwd_kernel<8, true>), dim3(dfl::up_grid(total)), dim3(256), 0, s, *a, total, cq);
  else if (vec == 4) hipLaunchKernelGGL((dfl::upsample2x_bwd_kernel<4, false>), dim3(dfl::up_grid(total)), dim3(256), 0, s, *a, total, cq);
  else hipLaunchKernelGGL((dfl::upsample2x_bwd_kernel<1, false>), dim3(dfl::up_grid(total)), dim3(256), 0, s, *a, total, cq);
  return dfl::check_launch("dfl_upsample2x_bwd");
}
