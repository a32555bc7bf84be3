// Output heads of the U-Net (reference: train_test_code/unet.py:176-191):
//   logits = seg_conv(x)                  1x1, no bias, F -> NC
//   seg    = Softmax2d(logits)            (or logits when do_soft_max=False)
//   heat   = lands_1x1(cat(x, logits))    1x1 (F+NC -> NM) [-> 1x1 (NM -> L)], no bias, no non-linearity
// One thread per pixel, 256 consecutive pixels per workgroup.  A thread-private walk over its 128-byte feature row would
// touch 64 cache lines per wave-instruction (measured: 5x the compulsory HBM reads), so the [256][F] tile is loaded
// cooperatively (consecutive lanes = consecutive 16 bytes) into LDS (pitch F + 4: conflict-free b128 row reads) and the
// same tile carries dx back out in the backward kernel; the head weights (5 KB) are read with
// wave-uniform addresses through `const __restrict__` kernel parameters, i.e. as scalar loads into SGPR operands of the
// FMAs (no LDS traffic: the LDS-broadcast version spent 224 ds_read_b128 per pixel), logits / mid / heat stay in registers, seg and heat are stored NCHW (lane = pixel
// => unit-stride stores per channel plane).  HBM-bound: 4*F bytes in, 4*(NC+L) bytes out per pixel.
// The backward kernel recomputes logits and mid from x (cheaper than saving them), applies the softmax Jacobian
//  and leaves dx plus a per-pixel scratch row from which the three small weight gradients are
// taken by dfl_conv2d_wgrad.  With bf16 features (F = 32) the weight gradients are taken inside the kernel instead (the
// scratch is 448 bytes per pixel to write and read back): a tile's rows [dlogits | dmid | dheat] and [x | logits | mid] go
// to LDS as bf16, four waves multiply them on the matrix cores through transposing LDS reads (contraction over pixels,
// as csrc/wgradp_bf16.hip), workgroups leave 64 x 64 fp32 partials and head_wgrad_finish_kernel adds them up.
#include "common.h"
#include <stdlib.h>


namespace dfl {

constexpr int HT = 256;   // pixels per workgroup tile

__device__ __forceinline__ unsigned hpack2(float x, float y) {   // two floats -> two bf16 (round to nearest even)
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){x, y}, bf16x2_t));
}
// 8 consecutive pixels of one column of a [pixel][column] bf16 image: two transposing reads (csrc/wgradp_bf16.hip)
__device__ __forceinline__ bf16x8_t htr_read8(const unsigned char* base, uint32_t r0, uint32_t r1) {
  typedef short hs16x4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) hs16x4_t* lds_p;
  const hs16x4_t x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + r0));
  const hs16x4_t y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + r1));
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7));
}

// tile[p][0..F) = x[m0 + p][0..F) (zeros past M); optionally the same values go to cat[(m0 + p) * cat_ld + ..] (scratch)
__device__ __forceinline__ void head_load_tile(float* tile, int pitch, const float* __restrict__ x, int ldx, int64_t m0,
                                               int64_t M, int F, float* __restrict__ cat, int cat_ld, int x_bf16) {
  if (x_bf16) {   // bf16 features (math mode 4): 8 channels per 16-byte load, widened to fp32 in the tile
    const int f8 = F / 8;
    const unsigned short* xb = reinterpret_cast<const unsigned short*>(x);
    for (int e = threadIdx.x; e < HT * f8; e += HT) {
      const int p = e / f8, q = e - p * f8;
      float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
      if (m0 + p < M) {
        const uint4 w = *reinterpret_cast<const uint4*>(xb + (m0 + p) * ldx + 8 * q);
        lo = make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                         __uint_as_float(w.y & 0xffff0000u));
        hi = make_float4(__uint_as_float(w.z << 16), __uint_as_float(w.z & 0xffff0000u), __uint_as_float(w.w << 16),
                         __uint_as_float(w.w & 0xffff0000u));
        if (cat != nullptr) {
          *reinterpret_cast<float4*>(cat + (m0 + p) * cat_ld + 8 * q) = lo;
          *reinterpret_cast<float4*>(cat + (m0 + p) * cat_ld + 8 * q + 4) = hi;
        }
      }
      *reinterpret_cast<float4*>(tile + p * pitch + 8 * q) = lo;
      *reinterpret_cast<float4*>(tile + p * pitch + 8 * q + 4) = hi;
    }
    return;
  }
  const int fq = F / 4;
  for (int e = threadIdx.x; e < HT * fq; e += HT) {
    const int p = e / fq, q = e - p * fq;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m0 + p < M) {
      v = *reinterpret_cast<const float4*>(x + (m0 + p) * ldx + 4 * q);
      if (cat != nullptr) *reinterpret_cast<float4*>(cat + (m0 + p) * cat_ld + 4 * q) = v;
    }
    *reinterpret_cast<float4*>(tile + p * pitch + 4 * q) = v;
  }
}

// The channel counts are template parameters for the reference's two head shapes (7 classes alone, 7 + 14 landmarks
// through 21): with run-time bounds every 4-FMA group became its own basic block (scalar load, wait, branch -- ~950
// branches per pixel, the kernels ran at 1 TB/s).  GEN = 1 keeps the run-time bounds for any other shape.
__global__ void __launch_bounds__(256) head_wgrad_finish_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ dw_seg,
                                                                float* __restrict__ dw_l1, float* __restrict__ dw_l2, int F, int NC, int NM, int L) {
  // 16 elements of the 64 x 64 product per workgroup x 16 lanes over the workgroup partials (eight loads in flight per
  // lane), lanes added up through LDS in a fixed order
  __shared__ double red[16][17];
  const int el = threadIdx.x & 15, ln = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int b = ln;
  for (; b + 7 * 16 < nblocks; b += 8 * 16) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(b + u * 16) * 4096 + e];
    s0 += (double)v[0] + (double)v[4];
    s1 += (double)v[1] + (double)v[5];
    s2 += (double)v[2] + (double)v[6];
    s3 += (double)v[3] + (double)v[7];
  }
  for (; b < nblocks; b += 16) s0 += (double)partial[(int64_t)b * 4096 + e];
  red[ln][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ln != 0) return;
  double tot = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) tot += red[k][el];
  const int row = e >> 6, col = e & 63;
  float* dst = nullptr;
  if (row < 8) {
    if (row < NC && col < F) dst = dw_seg + row * F + col;
  } else if (row < 32) {
    if (dw_l1 != nullptr && row - 8 < NM && col < F + NC) dst = dw_l1 + (row - 8) * (F + NC) + col;
  } else if (row < 48) {
    if (dw_l2 != nullptr && row - 32 < L && col >= 40 && col - 40 < NM) dst = dw_l2 + (row - 32) * NM + (col - 40);
  }
  if (dst != nullptr) *dst = (float)tot;
}

static unsigned head_grid(int64_t M) {
  int64_t b = ceil_div(M, 256);
  if (b > 4096) b = 4096;
  return (unsigned)(b < 1 ? 1 : b);
}

static unsigned head_wgrad_grid(int64_t M) {
  int64_t b = ceil_div(M, HT);
  if (b > 512) b = 512;                     // two workgroups per CU walk the tiles; one partial each
  return (unsigned)(b < 1 ? 1 : b);
}

#include "head_mfma.inc"

// bf16 features with F = 32 and a head inside the small capacities (two 1x1 landmark layers, or none): matrix-core kernels
static bool head_mfma_ok(int x_bf16, int F, int ldx, int L, const void* w_l2, const void* w_seg) {
  static const bool on = [] {
    const char* e = getenv("DFL_HEAD_MFMA");
    return e == nullptr || atoi(e) != 0;
  }();
  return on && x_bf16 && F == 32 && ldx % 8 == 0 && (L == 0 || w_l2 != nullptr) && aligned16(w_seg);
}

// The kernels and their launchers, once per capacity (csrc/head_caps.inc)
namespace head_small {
constexpr int MAXNC = DFL_HEAD_MAX_NC, MAXL = DFL_HEAD_MAX_L, MAXNM = DFL_HEAD_MAX_NM;
#define HEAD_HAS_FUSED true
#include "head_caps.inc"
#undef HEAD_HAS_FUSED
}  // namespace head_small

namespace head_large {
constexpr int MAXNC = DFL_HEAD_LARGE_NC, MAXL = DFL_HEAD_LARGE_L, MAXNM = DFL_HEAD_LARGE_NM;
#define HEAD_GENERIC_ONLY
#define HEAD_HAS_FUSED false
#include "head_caps.inc"
#undef HEAD_HAS_FUSED
#undef HEAD_GENERIC_ONLY
}  // namespace head_large

static inline bool head_is_large(int NC, int NM, int L) { return NC > DFL_HEAD_MAX_NC || NM > DFL_HEAD_MAX_NM || L > DFL_HEAD_MAX_L; }

}  // namespace dfl

using namespace dfl;

extern "C" int dfl_head_scratch_ld(int32_t F) { return head_small::scratch_ld(F); }
extern "C" int dfl_head_scratch_off(int32_t F, int32_t which) { return head_small::scratch_off(F, which); }

extern "C" int dfl_head_scratch_ld_for(int32_t F, int32_t NC, int32_t NM, int32_t L) {
  return head_is_large(NC, NM, L) ? head_large::scratch_ld(F) : head_small::scratch_ld(F);
}
extern "C" int dfl_head_scratch_off_for(int32_t F, int32_t NC, int32_t NM, int32_t L, int32_t which) {
  return head_is_large(NC, NM, L) ? head_large::scratch_off(F, which) : head_small::scratch_off(F, which);
}

extern "C" int dfl_head_fwd(const dfl_head_fwd_args* a, dfl_stream_t stream) {
  DFL_REQUIRE(a && a->x && a->w_seg && a->seg, "dfl_head_fwd: missing pointer");
  DFL_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->ldx >= a->F, "dfl_head_fwd: bad sizes");
  DFL_REQUIRE(a->L == 0 || a->heat != nullptr, "dfl_head_fwd: heat output required when L > 0");
  return head_is_large(a->NC, a->NM, a->L) ? head_large::launch_fwd(a, stream) : head_small::launch_fwd(a, stream);
}

extern "C" int dfl_head_wgrad_blocks(int64_t M) {
  DFL_REQUIRE(M > 0, "dfl_head_wgrad_blocks: bad size");
  return (int)head_wgrad_grid(M);
}

extern "C" int dfl_head_bwd(const dfl_head_bwd_args* a, dfl_stream_t stream) {
  DFL_REQUIRE(a && a->x && a->seg && a->dseg && a->w_seg && a->dx && (a->scratch || a->dw_seg), "dfl_head_bwd: missing pointer");
  DFL_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->ldx >= a->F && a->lddx >= a->F, "dfl_head_bwd: bad sizes");
  DFL_REQUIRE(a->lddx % (a->x_bf16 ? 8 : 4) == 0 && aligned16(a->dx), "dfl_head_bwd: dx alignment");
  return head_is_large(a->NC, a->NM, a->L) ? head_large::launch_bwd(a, stream) : head_small::launch_bwd(a, stream);
}
