// Direct (non-GEMM) kernels for layers with a handful of input channels: the network's first block (1-channel image:
// conv3x3 1->32, residual conv1x1 1->32 and their weight gradients; reference unet.py:207,211).  K = taps*Cin <= 12
// leaves the 32x32x2 MFMA tiles of conv_gemm.hip / wgrad_gemm.hip 3-25 % full and their general gather path slow;
// these layers are pure HBM streams (one read of the gradient / one write of the output), so one thread owns 4 output
// channels of a pixel, keeps its K x 4 weights (or K x 4 gradient accumulators) in registers and walks pixels.
// Same contracts as dfl_conv2d / dfl_conv2d_wgrad (include/dfl_hip.h), chosen by those entry points.
#include "common.h"
#include "direct_small.h"

namespace dfl {

constexpr int DK = DIRECT_MAX_K;

struct Gather {   // decoded pixel -> input coordinates of tap (0,0)
  int iy0, ix0, base;
};

// Output-pixel cursor advanced by a fixed step without divisions (the kernels are instruction-bound: three integer
// divisions per pixel cost more than the 9 x 4 multiply-adds of the 3x3 window)
struct PixCursor {
  int ox, oy, n;
  __device__ __forceinline__ void init(int m, int Hg, int Wg) {
    ox = m % Wg;
    const int t = m / Wg;
    oy = t % Hg;
    n = t / Hg;
  }
  __device__ __forceinline__ void advance(int step, int Hg, int Wg) {
    ox += step;
    while (ox >= Wg) {
      ox -= Wg;
      if (++oy == Hg) {
        oy = 0;
        ++n;
      }
    }
  }
  __device__ __forceinline__ Gather origin(int stride, int pad, int Hin, int Win) const {
    Gather g;
    g.iy0 = oy * stride - pad;
    g.ix0 = ox * stride - pad;
    g.base = n * Hin * Win;
    return g;
  }
};

// 4 consecutive channels of a row: fp32 (16 bytes) or bf16 (8 bytes) tensors
template <bool BF>
__device__ __forceinline__ float4 ld4c(const float* base, int64_t elem) {
  if constexpr (BF) {
    const uint2 w = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + elem);
    return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                       __uint_as_float(w.y & 0xffff0000u));
  } else {
    return *reinterpret_cast<const float4*>(base + elem);
  }
}
// store (rounding to bf16 when BF) and return the values as stored
template <bool BF>
__device__ __forceinline__ float4 st4c(float* base, int64_t elem, float4 v) {
  if constexpr (BF) {
    const bf16x2_t h0 = __builtin_convertvector((f32x2_t){v.x, v.y}, bf16x2_t), h1 = __builtin_convertvector((f32x2_t){v.z, v.w}, bf16x2_t);
    const uint2 w = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + elem) = w;
    return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                       __uint_as_float(w.y & 0xffff0000u));
  } else {
    *reinterpret_cast<float4*>(base + elem) = v;
    return v;
  }
}

// ---------------------------------------------------------------------------------------------- forward conv
// BF: y, add and stat_other are bf16 tensors (math mode 4; the input x of these layers is the fp32 network input)
template <int KH, int KW, int CIN, bool BF>
__global__ void __launch_bounds__(256) direct_conv_kernel(const dfl_conv_args a, int Mtot, int rows_per_block) {
  constexpr int K = KH * KW * CIN;
  static_assert(K <= DK, "window too large for the direct kernel");
  __shared__ float red[2][256][4];
  const int cq = a.Ntot / 4, PL = 256 / cq;
  const int q = threadIdx.x % cq, pl = threadIdx.x / cq;
  const int n0 = 4 * q;
  float w[K][4];
#pragma unroll
  for (int k = 0; k < K; ++k) {
    // quad-packed operand: w[(k/4)][n][k%4]
#pragma unroll
    for (int j = 0; j < 4; ++j) w[k][j] = a.w[((int64_t)(k >> 2) * a.Ntot + n0 + j) * 4 + (k & 3)];
  }
  float bias[4] = {0.f, 0.f, 0.f, 0.f}, asc[4] = {1.f, 1.f, 1.f, 1.f}, ash[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (a.bias != nullptr) bias[j] = a.bias[n0 + j];
    if (a.add_scale != nullptr) {
      asc[j] = a.add_scale[n0 + j];
      ash[j] = a.add_shift[n0 + j];
    }
  }
  if (a.add != nullptr && a.add_scale == nullptr && a.add_tot != nullptr) {
    // live statistics (include/dfl_hip.h): derived here, one channel per thread through LDS (four fp64 derivations per thread
    // made this 19 us kernel 32 us long)
    __shared__ float atab[2][1024];
    for (int c = threadIdx.x; c < a.Ntot; c += 256)
      bn_live_affine(a.add_tot, a.add_gamma, a.add_beta, a.add_count, a.bn_eps, a.Ntot, c, &atab[0][c], &atab[1][c]);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      asc[j] = atab[0][n0 + j];
      ash[j] = atab[1][n0 + j];
    }
  }
  float isc[CIN], ish[CIN];
#pragma unroll
  for (int c = 0; c < CIN; ++c) {
    isc[c] = (a.in_scale != nullptr) ? a.in_scale[c] : 1.f;
    ish[c] = (a.in_scale != nullptr) ? a.in_shift[c] : 0.f;
  }
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  const int m_begin = blockIdx.x * rows_per_block;
  const int m_end = min(Mtot, m_begin + rows_per_block);
  PixCursor cur;
  cur.init(min(m_begin + pl, Mtot - 1), a.Hout, a.Wout);
  for (int m = m_begin + pl; m < m_end; m += PL, cur.advance(PL, a.Hout, a.Wout)) {
    const Gather g = cur.origin(a.stride, a.pad, a.Hin, a.Win);
    float acc[4] = {bias[0], bias[1], bias[2], bias[3]};
    float xv[K];
#pragma unroll
    for (int dy = 0; dy < KH; ++dy)
#pragma unroll
      for (int dx = 0; dx < KW; ++dx)
#pragma unroll
        for (int c = 0; c < CIN; ++c) {   // all loads first (clamped addresses), then the arithmetic
          const int iy = g.iy0 + dy, ix = g.ix0 + dx;
          const bool ok = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
          const float x = a.x[ok ? ((int64_t)(g.base + iy * a.Win + ix) * a.ldx + c) : 0];
          xv[(dy * KW + dx) * CIN + c] = ok ? fmaf(x, isc[c], ish[c]) : 0.f;   // zero padding AFTER the BatchNorm affine
        }
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaf(xv[k], w[k][j], acc[j]);
    if (a.add != nullptr) {
      const float4 r = ld4c<BF>(a.add, (int64_t)m * a.ldadd + n0);
      acc[0] += fmaf(r.x, asc[0], ash[0]);
      acc[1] += fmaf(r.y, asc[1], ash[1]);
      acc[2] += fmaf(r.z, asc[2], ash[2]);
      acc[3] += fmaf(r.w, asc[3], ash[3]);
    }
    if (a.relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = fmaxf(acc[j], 0.f);
    }
    const int64_t yo = (int64_t)m * a.ldy + n0;
    if (a.accumulate) {
      const float4 o = ld4c<BF>(a.y, yo);
      acc[0] += o.x; acc[1] += o.y; acc[2] += o.z; acc[3] += o.w;
    }
    const float4 st = st4c<BF>(a.y, yo, make_float4(acc[0], acc[1], acc[2], acc[3]));
    acc[0] = st.x; acc[1] = st.y; acc[2] = st.z; acc[3] = st.w;      // statistics of the values as stored
    if (a.stat_partials != nullptr) {
      float u[4] = {acc[0], acc[1], acc[2], acc[3]};
      if (a.stat_other != nullptr) {
        const float4 o = ld4c<BF>(a.stat_other, (int64_t)m * a.ldso + n0);
        u[0] = o.x; u[1] = o.y; u[2] = o.z; u[3] = o.w;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        s1[j] += acc[j];
        s2[j] = fmaf(acc[j], u[j], s2[j]);
      }
    }
  }
  if (a.stat_partials == nullptr) return;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    red[0][threadIdx.x][j] = s1[j];
    red[1][threadIdx.x][j] = s2[j];
  }
  __syncthreads();
  if (pl == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float t1 = 0.f, t2 = 0.f;
      for (int p = 0; p < PL; ++p) {   // fixed order: bit-reproducible
        t1 += red[0][p * cq + q][j];
        t2 += red[1][p * cq + q][j];
      }
      a.stat_partials[((int64_t)blockIdx.x * 2 + 0) * a.Ntot + n0 + j] = t1;
      a.stat_partials[((int64_t)blockIdx.x * 2 + 1) * a.Ntot + n0 + j] = t2;
    }
  }
}

// ---------------------------------------------------------------------------------------------- 3x3 window, ONE input channel: row form
// The network's first convolution and its weight gradient (unet.py:211: Conv2d(1, 32, 3)) are pure streams: 2.4 MB of image
// against 37.7 MB of bf16 output / output gradient at batch 16.  The general direct kernels above walk pixels with a cursor,
// four channels and ONE pixel per thread and iteration: every iteration waits for its own loads (18 dependent round trips per
// thread in the weight gradient: 69 us where the data take 7 at the HBM rate) and spends more instructions on addresses than on
// multiply-adds.  Here a workgroup owns RB rows of one image, a thread 8 channels of the pixels x = pl, pl + PL, ... of each
// row: no divisions, row validity is workgroup-uniform, the loops are short and fully unrolled by the compiler (all loads of a
// row in flight), stores / loads of the big tensor are 16 bytes, and the weight gradient adds its accumulators up with wave
// shuffles + one LDS round instead of 36 barrier rounds.
template <bool BF>
__device__ __forceinline__ void ld8c(const float* base, int64_t elem, float* v) {
  if constexpr (BF) {
    const uint4 w = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(base) + elem);
    v[0] = __uint_as_float(w.x << 16); v[1] = __uint_as_float(w.x & 0xffff0000u);
    v[2] = __uint_as_float(w.y << 16); v[3] = __uint_as_float(w.y & 0xffff0000u);
    v[4] = __uint_as_float(w.z << 16); v[5] = __uint_as_float(w.z & 0xffff0000u);
    v[6] = __uint_as_float(w.w << 16); v[7] = __uint_as_float(w.w & 0xffff0000u);
  } else {
    const float4 a = *reinterpret_cast<const float4*>(base + elem), b = *reinterpret_cast<const float4*>(base + elem + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
  }
}
// store 8 channels (rounding to bf16 when BF); v receives the values as stored
template <bool BF>
__device__ __forceinline__ void st8c(float* base, int64_t elem, float* v) {
  if constexpr (BF) {
    uint4 w;
    w.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[0], v[1]}, bf16x2_t));
    w.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[2], v[3]}, bf16x2_t));
    w.z = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[4], v[5]}, bf16x2_t));
    w.w = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){v[6], v[7]}, bf16x2_t));
    *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(base) + elem) = w;
    v[0] = __uint_as_float(w.x << 16); v[1] = __uint_as_float(w.x & 0xffff0000u);
    v[2] = __uint_as_float(w.y << 16); v[3] = __uint_as_float(w.y & 0xffff0000u);
    v[4] = __uint_as_float(w.z << 16); v[5] = __uint_as_float(w.z & 0xffff0000u);
    v[6] = __uint_as_float(w.w << 16); v[7] = __uint_as_float(w.w & 0xffff0000u);
  } else {
    *reinterpret_cast<float4*>(base + elem) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(base + elem + 4) = make_float4(v[4], v[5], v[6], v[7]);
  }
}

#ifndef DFL_ROWS_CONV_RB
#define DFL_ROWS_CONV_RB 2
#endif
#ifndef DFL_ROWS_WGRAD_RB
#define DFL_ROWS_WGRAD_RB 4
#endif
#ifndef DFL_ROWS_CONV_BPB
#define DFL_ROWS_CONV_BPB 3
#endif
#ifndef DFL_ROWS_WGRAD_BPB
#define DFL_ROWS_WGRAD_BPB 1
#endif
// bands per workgroup of the 3x3 row forms (round 4).  Forward: 1 / 2 / 3 / 6 / 12 bands: 35 / 25 / 25 / 29 / 44 us (the 72 weights
// of a thread are fetched once per workgroup).  Weight gradient: 1 / 2 / 3 / 4 / 6 bands: 40 / 47 / 55 / 66 / 83 us, and 2 / 3 rows per
// band instead of 4: 64 / 63 us -- its threads wait for their own d / r loads pixel by pixel, workgroups are its parallelism
constexpr int ROWS_CONV_BPB = DFL_ROWS_CONV_BPB, ROWS_WGRAD_BPB = DFL_ROWS_WGRAD_BPB;
constexpr int ROWS_CONV_RB = DFL_ROWS_CONV_RB, ROWS_WGRAD_RB = DFL_ROWS_WGRAD_RB;   // (measured against 4 / 6 rows with four pixels in flight per thread and the
// weights staged through LDS: 37 / 40 us here, 79 / 47 us there -- 184 registers halve the occupancy these short loops live on)

template <bool BF>
__global__ void __launch_bounds__(256) direct_conv3_rows_kernel(const dfl_conv_args a) {
  __shared__ float red[2][256][8];
  const int cq = a.Ntot >> 3, PL = 256 / cq;
  const int q = threadIdx.x % cq, pl = threadIdx.x / cq;
  const int n0 = 8 * q;
  const int bands = (a.Hout + ROWS_CONV_RB - 1) / ROWS_CONV_RB;
  float w[9][8], bias[8], s1[8], s2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k][j] = a.w[((int64_t)(k >> 2) * a.Ntot + n0 + j) * 4 + (k & 3)];   // quad-packed operand
    bias[j] = a.bias != nullptr ? a.bias[n0 + j] : 0.f;
    s1[j] = 0.f;
    s2[j] = 0.f;
  }
  // a workgroup walks ROWS_CONV_BPB bands: the 72 weights of a thread (73 dependent-free but serial scalar loads in front of
  // six pixels of work) are fetched once for all of them
  extern __shared__ float img_rows[];                 // [ROWS_CONV_RB + 2][Wout + 2]
  const int RW = a.Wout + 2;
  for (int bb = 0; bb < ROWS_CONV_BPB; ++bb) {
  const int band = (int)blockIdx.x * ROWS_CONV_BPB + bb;
  if (band >= a.N * bands) break;
  const int n = band / bands, y0 = (band - n * bands) * ROWS_CONV_RB;
  const float* img = a.x + (int64_t)n * a.Hin * a.Win * a.ldx;
  if (bb > 0) __syncthreads();                        // the previous band's rows have been read
  // the band's input rows (its output rows + the window's halo, zero outside the image) go to LDS once, coalesced: the per-pixel
  // gathers below then never wait for global memory (round 4: nine dependent 4-byte loads per pixel had made this 0.3 GFLOP kernel
  // 39 us long)
  for (int idx = threadIdx.x; idx < (ROWS_CONV_RB + 2) * RW; idx += 256) {
    const int r = idx / RW, c = idx - r * RW;
    const int iy = y0 - a.pad + r, ix = c - a.pad;
    const bool ok = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
    img_rows[idx] = ok ? img[(int64_t)(iy * a.Win + ix) * a.ldx] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int ry = 0; ry < ROWS_CONV_RB; ++ry) {
    const int y = y0 + ry;
    if (y >= a.Hout) break;
    for (int x = pl; x < a.Wout; x += PL) {
      float xv[9];
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) xv[dy * 3 + dx] = img_rows[(ry + dy) * RW + x + dx];
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = bias[j];
#pragma unroll
      for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv[k], w[k][j], acc[j]);
      if (a.relu) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaxf(acc[j], 0.f);
      }
      st8c<BF>(a.y, ((int64_t)(n * a.Hout + y) * a.Wout + x) * a.ldy + n0, acc);
#pragma unroll
      for (int j = 0; j < 8; ++j) {   // statistics of the values as stored
        s1[j] += acc[j];
        s2[j] = fmaf(acc[j], acc[j], s2[j]);
      }
    }
  }
  // statistics: one row per BAND (as when a workgroup took one band: the same sums in the same order, whatever ROWS_CONV_BPB)
  if (a.stat_partials != nullptr || a.stat_totals != nullptr) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[0][threadIdx.x][j] = s1[j];
      red[1][threadIdx.x][j] = s2[j];
      s1[j] = 0.f;
      s2[j] = 0.f;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < 2 * a.Ntot; idx += 256) {   // fixed order: bit-reproducible
      const int which = idx / a.Ntot, c = idx - which * a.Ntot;
      const int qq = c >> 3, j = c & 7;
      float t = 0.f;
      for (int p = 0; p < PL; ++p) t += red[which][p * cq + qq][j];
      if (a.stat_totals != nullptr) bn_live_add(a.stat_totals, band, which, a.Ntot, c, t);     // live statistics (include/dfl_hip.h)
      else a.stat_partials[((int64_t)band * 2 + which) * a.Ntot + c] = t;
    }
  }
  }
}

static bool direct_conv_rows_ok(const dfl_conv_args* a) {
  return a->KH == 3 && a->KW == 3 && a->Cin == 1 && a->stride == 1 && a->add == nullptr && !a->accumulate && a->stat_other == nullptr &&
         a->in_scale == nullptr && a->Ntot % 8 == 0 && a->Ntot <= 64 && 256 % (a->Ntot / 8) == 0 && a->ldy % 8 == 0 && aligned16(a->y);
}

// dw[cm][0][t] (or the tap-major partial slot of this workgroup) = sum over its pixels of x(tap t) * d[cm]
// DB (dfl_wgrad_args.d_mode, bf16 d): d is dy, the operand [r > 0] * (A dy + B r + C) (r = d2, rounded to bf16 as the materialised
// tensor was) is formed on the way in and its column sums -- the layer's bias gradient -- leave with the slice (bias_partial).
template <bool BF, bool DB = false>
__global__ void __launch_bounds__(256) direct_wgrad3_rows_kernel(const dfl_wgrad_args a) {
  __shared__ float red[4][8 * 72];              // [wave][q][k][j]
  const int cq = a.Cm >> 3, PL = 256 / cq;
  const int q = threadIdx.x % cq, pl = threadIdx.x / cq;
  const int bands = (a.Hout + ROWS_WGRAD_RB - 1) / ROWS_WGRAD_RB;
  float acc[9][8];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[k][j] = 0.f;
  extern __shared__ float img_rows[];                 // [ROWS_WGRAD_RB + 2][Wout + 2]: the band's input rows with halo (as the forward kernel)
  const int RW = a.Wout + 2;
  float cA[DB ? 8 : 1], cB[DB ? 8 : 1], cC[DB ? 8 : 1], bsum[DB ? 8 : 1];
  if constexpr (DB) {
    if (a.coef_tot != nullptr) {            // live statistics: one channel per thread into LDS (red is free until the end)
      float* ctab = &red[0][0];             // [3][Cm], Cm <= 64
      if ((int)threadIdx.x < a.Cm)
        bn_live_coef(a.coef_tot, a.bn_gamma, a.bn_mean, a.bn_invstd, a.bn_count, a.Cm, (int)threadIdx.x, ctab + threadIdx.x, ctab + 64 + threadIdx.x,
                     ctab + 128 + threadIdx.x);
      __syncthreads();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        cA[j] = ctab[8 * q + j];
        cB[j] = ctab[64 + 8 * q + j];
        cC[j] = ctab[128 + 8 * q + j];
        bsum[j] = 0.f;
      }
      __syncthreads();
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        cA[j] = a.coef != nullptr ? a.coef[8 * q + j] : 1.f;
        cB[j] = a.coef != nullptr ? a.coef[a.Cm + 8 * q + j] : 0.f;
        cC[j] = a.coef != nullptr ? a.coef[2 * a.Cm + 8 * q + j] : 0.f;
        bsum[j] = 0.f;
      }
    }
  }
  // a workgroup walks ROWS_WGRAD_BPB bands (round 4): one coefficient table, one 72-value reduction tree and one partial slot
  // for all of them
  for (int bb = 0; bb < ROWS_WGRAD_BPB; ++bb) {
  const int band = (int)blockIdx.x * ROWS_WGRAD_BPB + bb;
  if (band >= a.N * bands) break;
  const int n = band / bands, y0 = (band - n * bands) * ROWS_WGRAD_RB;
  const float* img = a.g + (int64_t)n * a.Hin * a.Win * a.ldg;
  if (bb > 0) __syncthreads();                        // the previous band's rows have been read
  for (int idx = threadIdx.x; idx < (ROWS_WGRAD_RB + 2) * RW; idx += 256) {
    const int r = idx / RW, c = idx - r * RW;
    const int iy = y0 - a.pad + r, ix = c - a.pad;
    const bool ok = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
    img_rows[idx] = ok ? img[(int64_t)(iy * a.Win + ix) * a.ldg] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int ry = 0; ry < ROWS_WGRAD_RB; ++ry) {
    const int y = y0 + ry;
    if (y >= a.Hout) break;
    for (int x = pl; x < a.Wout; x += PL) {
      float d[8], xv[9];
      ld8c<BF>(a.d, ((int64_t)(n * a.Hout + y) * a.Wout + x) * a.ldd + 8 * q, d);
      if constexpr (DB) {
        float r[8];
        ld8c<BF>(a.d2, ((int64_t)(n * a.Hout + y) * a.Wout + x) * a.ldd2 + 8 * q, r);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = r[j] > 0.f ? fmaf(cA[j], d[j], fmaf(cB[j], r[j], cC[j])) : 0.f;
          d[j] = (float)(__bf16)v;
          bsum[j] += d[j];
        }
      }
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) xv[dy * 3 + dx] = img_rows[(ry + dy) * RW + x + dx];
#pragma unroll
      for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = fmaf(xv[k], d[j], acc[k][j]);
    }
  }
  }
  // lanes of one wave that share q (lane % cq): butterfly over the pixel lanes, then the four waves through LDS -- a fixed tree
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = acc[k][j];
      for (int off = cq; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
      acc[k][j] = v;
    }
  if (lane < cq) {
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) red[wave][(lane * 9 + k) * 8 + j] = acc[k][j];
  }
  __syncthreads();
  const bool sliced = a.splits > 1;
  float* out = sliced ? a.partial + (int64_t)blockIdx.x * a.Cm * 9 : a.dw;
  for (int idx = threadIdx.x; idx < cq * 72; idx += 256) {
    const float t = ((red[0][idx] + red[1][idx]) + red[2][idx]) + red[3][idx];
    const int qq = idx / 72, r = idx - qq * 72, k = r >> 3, j = r & 7;
    const int cm = 8 * qq + j;
    out[sliced ? (int64_t)k * a.Cm + cm : (int64_t)cm * 9 + k] = t;
  }
  if constexpr (DB) {
    if (a.bias_partial == nullptr) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v = bsum[j];
      for (int off = cq; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
      bsum[j] = v;
    }
    __syncthreads();                            // (the weight-gradient sums above are read)
    if (lane < cq) {
#pragma unroll
      for (int j = 0; j < 8; ++j) red[wave][lane * 8 + j] = bsum[j];
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < a.Cm; idx += 256)
      a.bias_partial[(int64_t)blockIdx.x * a.Cm + idx] = ((red[0][idx] + red[1][idx]) + red[2][idx]) + red[3][idx];
  }
}

static bool direct_wgrad_rows_ok(const dfl_wgrad_args* a) {
  return a->KH == 3 && a->KW == 3 && a->Cg == 1 && a->stride == 1 && a->in_scale == nullptr && a->Cm % 8 == 0 && a->Cm <= 64 &&
         256 % (a->Cm / 8) == 0 && 64 % (a->Cm / 8) == 0 && a->ldd % 8 == 0 && aligned16(a->d) && !a->g_bf16;
}

// instantiated windows (KH = KW): 3x3 with 1 channel, 2x2 with <= 3, 1x1 with <= 4
static bool direct_window(int KH, int KW, int C) {
  if (KH != KW) return false;
  return (KH == 3 && C == 1) || (KH == 2 && C >= 1 && C <= 3) || (KH == 1 && C >= 1 && C <= 4);
}

bool direct_conv_rows_usable(const dfl_conv_args* a) { return direct_conv_rows_ok(a); }

bool direct_conv_ok(const dfl_conv_args* a) {
  if (!direct_window(a->KH, a->KW, a->Cin) || a->scatter2x2 || a->splits > 1) return false;
  if (a->Ntot % 4 != 0 || a->Ntot > 1024 || 256 % (a->Ntot / 4) != 0) return false;
  if (a->x_bf16) return false;                       // (the input of these layers is the fp32 network input)
  if (a->ldy % 4 != 0 || !aligned16(a->y)) return false;
  if (a->add != nullptr && (a->ldadd % 4 != 0 || !aligned16(a->add))) return false;
  if (a->stat_other != nullptr && (a->ldso % 4 != 0 || !aligned16(a->stat_other))) return false;
  return true;
}

int direct_conv_blocks(const dfl_conv_args* a) {
  if (direct_conv_rows_ok(a)) return a->N * (int)ceil_div(a->Hout, ROWS_CONV_RB);      // (bands: rows of stat_partials; ROWS_CONV_BPB of them per workgroup)
  const int64_t M = (int64_t)a->N * a->Hout * a->Wout;
  const int PL = 256 / (a->Ntot / 4);
  int64_t b = ceil_div(M, (int64_t)PL * 8);
  if (b > 2048) b = 2048;
  return (int)(b < 1 ? 1 : b);
}

int direct_conv_launch(const dfl_conv_args* a, hipStream_t s) {
  const int64_t M = (int64_t)a->N * a->Hout * a->Wout;
  const int blocks = direct_conv_blocks(a);
  if (direct_conv_rows_ok(a)) {
    const size_t lds = (size_t)(ROWS_CONV_RB + 2) * (a->Wout + 2) * sizeof(float);
    const unsigned wgs = (unsigned)ceil_div(blocks, ROWS_CONV_BPB);
    if (a->y_bf16) hipLaunchKernelGGL(direct_conv3_rows_kernel<true>, dim3(wgs), dim3(256), lds, s, *a);
    else hipLaunchKernelGGL(direct_conv3_rows_kernel<false>, dim3(wgs), dim3(256), lds, s, *a);
    return check_launch("dfl_conv2d");
  }
  const int rpb = (int)ceil_div(M, blocks);
#define DFL_DC(KH_, C_)                                                                                                  \
  {                                                                                                                      \
    if (a->y_bf16) hipLaunchKernelGGL((direct_conv_kernel<KH_, KH_, C_, true>), dim3((unsigned)blocks), dim3(256), 0, s, *a, (int)M, rpb);  \
    else hipLaunchKernelGGL((direct_conv_kernel<KH_, KH_, C_, false>), dim3((unsigned)blocks), dim3(256), 0, s, *a, (int)M, rpb);           \
  }
  switch (a->KH * 10 + a->Cin) {
    case 31: DFL_DC(3, 1); break;
    case 21: DFL_DC(2, 1); break;
    case 22: DFL_DC(2, 2); break;
    case 23: DFL_DC(2, 3); break;
    case 11: DFL_DC(1, 1); break;
    case 12: DFL_DC(1, 2); break;
    case 13: DFL_DC(1, 3); break;
    default: DFL_DC(1, 4); break;
  }
#undef DFL_DC
  return check_launch("dfl_conv2d");
}

// ---------------------------------------------------------------------------------------------- weight gradient
// acc[k][j] = sum over this thread's pixels of G(m, k) * d[m][4q + j];  k = tap*Cg + cg
// BF: d is a bf16 tensor (math mode 4; g is the fp32 network input)
template <int KH, int KW, int CG, bool BF>
__global__ void __launch_bounds__(256) direct_wgrad_kernel(const dfl_wgrad_args a, int Mtot, int rows_per_block) {
  constexpr int K = KH * KW * CG, T = KH * KW;
  static_assert(K <= DK, "window too large for the direct kernel");
  __shared__ float red[256][4];
  const int cq = a.Cm / 4, PL = 256 / cq;
  const int q = threadIdx.x % cq, pl = threadIdx.x / cq;
  const int Cg = CG;
  float isc[CG], ish[CG];
#pragma unroll
  for (int c = 0; c < CG; ++c) {
    isc[c] = (a.in_scale != nullptr) ? a.in_scale[c] : 1.f;
    ish[c] = (a.in_scale != nullptr) ? a.in_shift[c] : 0.f;
  }
  float acc[K][4];
#pragma unroll
  for (int k = 0; k < K; ++k)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[k][j] = 0.f;
  const int m_begin = blockIdx.x * rows_per_block;
  const int m_end = min(Mtot, m_begin + rows_per_block);
  PixCursor cur;
  cur.init(min(m_begin + pl, Mtot - 1), a.Hout, a.Wout);
  for (int m = m_begin + pl; m < m_end; m += PL, cur.advance(PL, a.Hout, a.Wout)) {
    const Gather g = cur.origin(a.stride, a.pad, a.Hin, a.Win);
    const float4 d = ld4c<BF>(a.d, (int64_t)m * a.ldd + 4 * q);
    float xv[K];
#pragma unroll
    for (int dy = 0; dy < KH; ++dy)
#pragma unroll
      for (int dx = 0; dx < KW; ++dx)
#pragma unroll
        for (int c = 0; c < CG; ++c) {
          const int iy = g.iy0 + dy, ix = g.ix0 + dx;
          const bool ok = (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
          const float x = a.g[ok ? ((int64_t)(g.base + iy * a.Win + ix) * a.ldg + c) : 0];
          xv[(dy * KW + dx) * CG + c] = ok ? fmaf(x, isc[c], ish[c]) : 0.f;
        }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      acc[k][0] = fmaf(xv[k], d.x, acc[k][0]);
      acc[k][1] = fmaf(xv[k], d.y, acc[k][1]);
      acc[k][2] = fmaf(xv[k], d.z, acc[k][2]);
      acc[k][3] = fmaf(xv[k], d.w, acc[k][3]);
    }
  }
  const bool sliced = a.splits > 1;
  float* out = sliced ? a.partial + (int64_t)blockIdx.x * a.Cm * a.Cg * T : a.dw;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) red[threadIdx.x][j] = acc[k][j];
    __syncthreads();
    if (pl == 0) {
      const int t = k / Cg, cg = k - t * Cg;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float s = 0.f;
        for (int p = 0; p < PL; ++p) s += red[p * cq + q][j];
        const int cm = 4 * q + j;
        out[sliced ? ((int64_t)t * a.Cm + cm) * Cg + cg : ((int64_t)cm * Cg + cg) * T + t] = s;
      }
    }
  }
}

bool direct_wgrad_ok(const dfl_wgrad_args* a) {
  if (!direct_window(a->KH, a->KW, a->Cg)) return false;
  if (a->Cm % 4 != 0 || a->Cm > 1024 || 256 % (a->Cm / 4) != 0) return false;
  if (a->ldd % 4 != 0 || !aligned16(a->d) || a->g_bf16) return false;
  return true;
}

int direct_wgrad_splits(const dfl_wgrad_args* a) {
  if (direct_wgrad_rows_ok(a)) return (int)ceil_div((int64_t)a->N * ceil_div(a->Hout, ROWS_WGRAD_RB), ROWS_WGRAD_BPB);
  const int64_t M = (int64_t)a->N * a->Hout * a->Wout;
  const int PL = 256 / (a->Cm / 4);
  int64_t b = ceil_div(M, (int64_t)PL * 16);
  if (b > 1024) b = 1024;
  return (int)(b < 1 ? 1 : b);
}

int direct_wgrad_launch(const dfl_wgrad_args* a, hipStream_t s) {
  if (direct_wgrad_rows_ok(a) && a->splits == direct_wgrad_splits(a)) {     // (the row form's slots are its workgroups)
    const size_t rows_lds = (size_t)(ROWS_WGRAD_RB + 2) * (a->Wout + 2) * sizeof(float);
    DFL_REQUIRE(a->coef_tot == nullptr || (a->d_mode == 1 && a->coef == nullptr && a->bn_gamma && a->bn_mean && a->bn_invstd && a->bn_count > 0),
                "dfl_conv2d_wgrad (1-channel 3x3): coef_tot replaces coef (d_mode 1) and needs bn_gamma, bn_mean, bn_invstd, bn_count");
    if (a->d_mode != 0) {
      DFL_REQUIRE(a->d_mode == 1 && a->d_bf16 && a->d2 != nullptr && a->ldd2 % 8 == 0 && aligned16(a->d2),
                  "dfl_conv2d_wgrad (1-channel 3x3): d_mode 1 needs bf16 d / d2 (16-byte aligned, ldd2 %% 8 == 0)");
      hipLaunchKernelGGL((direct_wgrad3_rows_kernel<true, true>), dim3((unsigned)a->splits), dim3(256), rows_lds, s, *a);
    } else if (a->d_bf16) {
      hipLaunchKernelGGL(direct_wgrad3_rows_kernel<true>, dim3((unsigned)a->splits), dim3(256), rows_lds, s, *a);
    } else {
      hipLaunchKernelGGL(direct_wgrad3_rows_kernel<false>, dim3((unsigned)a->splits), dim3(256), rows_lds, s, *a);
    }
    return check_launch("dfl_conv2d_wgrad");
  }
  DFL_REQUIRE(a->d_mode == 0, "dfl_conv2d_wgrad: d_mode is implemented by the bf16 patch kernels and the 1-channel 3x3 row form only");
  const int64_t M = (int64_t)a->N * a->Hout * a->Wout;
  const int rpb = (int)ceil_div(M, a->splits);
#define DFL_DW(KH_, C_)                                                                                                  \
  {                                                                                                                      \
    if (a->d_bf16) hipLaunchKernelGGL((direct_wgrad_kernel<KH_, KH_, C_, true>), dim3((unsigned)a->splits), dim3(256), 0, s, *a, (int)M, rpb);  \
    else hipLaunchKernelGGL((direct_wgrad_kernel<KH_, KH_, C_, false>), dim3((unsigned)a->splits), dim3(256), 0, s, *a, (int)M, rpb);           \
  }
  switch (a->KH * 10 + a->Cg) {
    case 31: DFL_DW(3, 1); break;
    case 21: DFL_DW(2, 1); break;
    case 22: DFL_DW(2, 2); break;
    case 23: DFL_DW(2, 3); break;
    case 11: DFL_DW(1, 1); break;
    case 12: DFL_DW(1, 2); break;
    case 13: DFL_DW(1, 3); break;
    default: DFL_DW(1, 4); break;
  }
#undef DFL_DW
  return check_launch("dfl_conv2d_wgrad");
}

}  // namespace dfl
