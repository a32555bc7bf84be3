// Latency form of the convolution for fp32 TENSORS (round 5): what convs_bf16.hip is for the bf16 storage mode, for the arithmetics
// that hold north_star's 1e-4 forward bar -- math mode 0 (fp32 matrix instructions, v_mfma_f32_32x32x2_f32) and mode 1 (bf16x3:
// every fp32 operand value cut into hi + lo bf16 parts, hi*hi + hi*lo + lo*hi on the bf16 matrix pipe).  Same idea, same contract
// (dfl_conv_args.latency_form: the small problems of a batch-1 inference forward; the per-image loops of train_test_code/util.py:116-165,
// :318-356 call UNet.forward, unet.py:161-193, on one image): a wave owns one 32-pixel x 32-channel tile and a range of k-steps of 16
// channels; both fragments come straight from global memory -- the lane's own 8 consecutive fp32 channels of its pixel at the tap's
// offset (32 bytes, zero padding = out-of-range buffer offset) and two quads of the packed weights [K/4][N][4] (plain fp32 or split
// hi4 | lo4 bf16) -- 8 k-steps are in flight before the first matrix instruction (two register sets of 4: a step holds 16 registers
// here), D = W * X, epilogue on the accumulators in 16-byte stores, K split over the 4 / 8 waves of a workgroup (LDS, fixed order) and
// over workgroups (fp32 slices + the finish kernel of conv_gemm.hip).  Pairs (dfl_conv2d_pair) and the output affine work as for bf16
// tensors; nothing is rounded on the way, so the producer-side BatchNorm IS the consumer's affine on load.
// The fp32 instruction contracts k in pairs {j, 8 + j} of a step (lane half = k half): sums are formed in another order than
// conv_gemm_kernel's, within the 1e-6-class differences every tile shape of that kernel has against every other.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "convp.h"

namespace dfl {

constexpr uint32_t FOOB = 0x80000000u;
typedef unsigned int fu32x4 __attribute__((ext_vector_type(4)));
constexpr int FU = 4;               // k-steps per register set (16 registers per step)
constexpr int FCONST_PER_WAVE = 6 * 32;

struct ConvS32 {
  dfl_conv_args a;
  int Mtot, Hg, Wg, Cout, T;
  int mt, nt, ksplit_shift, cpk_shift, ksteps, kper, splits, grid, waves, math;
  uint32_t x_bytes, w_bytes;
  // second product of a pair (1x1, stride 1): y2 = x3 * w3 + bias3 + add_scale * y1 + add_shift
  const float* x3;
  const float* w3;
  const float* bias3;
  const float* add_scale;
  const float* add_shift;
  float* y2;
  int ldx3, ldy2, ksteps2, kper2, x3_one, w3_split;
  uint32_t x3_bytes, w3_bytes;
};

__device__ __forceinline__ uint32_t fpack_bf2(float a, float b) {
  const bf16x2_t h = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
  return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ float fbf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float fbf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
// 8 fp32 values -> hi and lo bf16 parts (hi = bf16(v), lo = bf16(v - hi)): the split conv_gemm_kernel applies at its LDS write
__device__ __forceinline__ void fsplit8(const fu32x4 v0, const fu32x4 v1, fu32x4* hi, fu32x4* lo) {
  const float f[8] = {__uint_as_float(v0.x), __uint_as_float(v0.y), __uint_as_float(v0.z), __uint_as_float(v0.w),
                      __uint_as_float(v1.x), __uint_as_float(v1.y), __uint_as_float(v1.z), __uint_as_float(v1.w)};
  uint32_t h[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    h[e] = fpack_bf2(f[2 * e], f[2 * e + 1]);
    l[e] = fpack_bf2(f[2 * e] - fbf_lo(h[e]), f[2 * e + 1] - fbf_hi(h[e]));
  }
  *hi = (fu32x4){h[0], h[1], h[2], h[3]};
  *lo = (fu32x4){l[0], l[1], l[2], l[3]};
}

template <int MATH>
__device__ __forceinline__ void fmma(f32x16& acc, const fu32x4 w0, const fu32x4 w1, const fu32x4 x0, const fu32x4 x1, bool w_split) {
  if constexpr (MATH == 0) {
    const float wv[8] = {__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z), __uint_as_float(w0.w),
                         __uint_as_float(w1.x), __uint_as_float(w1.y), __uint_as_float(w1.z), __uint_as_float(w1.w)};
    const float xv[8] = {__uint_as_float(x0.x), __uint_as_float(x0.y), __uint_as_float(x0.z), __uint_as_float(x0.w),
                         __uint_as_float(x1.x), __uint_as_float(x1.y), __uint_as_float(x1.z), __uint_as_float(x1.w)};
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j], xv[j], acc, 0, 0, 0);
  } else {
    fu32x4 whi, wlo, xhi, xlo;
    if (w_split) {                                        // split quads: 4 hi bf16 | 4 lo bf16 per 16-byte slot
      whi = (fu32x4){w0.x, w0.y, w1.x, w1.y};
      wlo = (fu32x4){w0.z, w0.w, w1.z, w1.w};
    } else {
      fsplit8(w0, w1, &whi, &wlo);
    }
    fsplit8(x0, x1, &xhi, &xlo);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wlo), __builtin_bit_cast(bf16x8_t, xhi), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, whi), __builtin_bit_cast(bf16x8_t, xlo), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, whi), __builtin_bit_cast(bf16x8_t, xhi), acc, 0, 0, 0);
  }
}

__global__ void convs32_pair_finish_kernel(const ConvS32 p);

template <int MATH, bool AFF, bool PAIR, int SWAVES>
__global__ void __launch_bounds__(64 * SWAVES, SWAVES == 8 ? 1 : 2) convs32_kernel(const ConvS32 p) {
  constexpr int NACC = PAIR ? 2 : 1;
  constexpr int SRED = SWAVES * NACC * 16 * 64;
  constexpr int TABQ = 1024 / (64 * SWAVES);
  extern __shared__ __attribute__((aligned(16))) float sm[];      // partial tiles, per-wave constants, [2][Cin] scale / shift
  const dfl_conv_args& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int ksh = p.ksplit_shift, ksplit = 1 << ksh;
  const int grp = wave >> ksh, kw = wave & (ksplit - 1);
  const int tile = (int)blockIdx.x * (SWAVES >> ksh) + grp;
  const bool tok = tile < p.mt * p.nt;
  const int mi = tok ? tile / p.nt : 0, ni = tok ? tile - mi * p.nt : 0;
  const int pix = mi * 32 + li;
  const bool pok = tok && pix < p.Mtot;
  const int HW = p.Hg * p.Wg;
  const int img = pix / HW, rem = pix - img * HW;
  const int gy = rem / p.Wg, gx = rem - gy * p.Wg;
  const int iy0 = gy * a.stride - a.pad, ix0 = gx * a.stride - a.pad;
  const int pbase = (img * a.Hin + iy0) * a.Win + ix0;
  const uint32_t ldx4 = (uint32_t)a.ldx * 4u, lh32 = (uint32_t)lh * 32u;
  const int n = ni * 32 + li;
  const bool nok = tok && n < a.Ntot;
  const uint32_t wq = (uint32_t)a.Ntot * 16u, wcol = (uint32_t)n * 16u;          // bytes per quad row, this lane's column

  __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)p.w_bytes, 0x00020000);

  const int cpk_sh = p.cpk_shift, cpk = 1 << cpk_sh;
  const int KW = a.KW, kw_magic = (256 + KW - 1) / KW;
  const int slice = (int)blockIdx.y * ksplit + kw;
  const int s_begin = slice * p.kper;
  const int n1 = max(0, min(p.ksteps, s_begin + p.kper) - s_begin);
  const bool w_split = a.w_split != 0;

  fu32x4 xb[2][FU][2], wb[2][FU][2];
  uint32_t okm[2] = {0u, 0u};
  auto load_group = [&](int buf, int v0) {
    uint32_t m = 0;
#pragma unroll
    for (int u = 0; u < FU; ++u) {
      const int v = v0 + u;
      const int s = s_begin + v;
      const bool live = v < n1;
      const int tap = s >> cpk_sh, cc = s & (cpk - 1);
      const int ty = (tap * kw_magic) >> 8, tx = tap - ty * KW;
      const bool ok = pok && live && (unsigned)(iy0 + ty) < (unsigned)a.Hin && (unsigned)(ix0 + tx) < (unsigned)a.Win;
      const uint32_t xo = (uint32_t)(pbase + ty * a.Win + tx) * ldx4 + (uint32_t)(cc * 64) + lh32;
      xb[buf][u][0] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? xo : FOOB, 0, 0);
      xb[buf][u][1] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? xo + 16u : FOOB, 0, 0);
      const uint32_t wo = (uint32_t)(s * 4 + 2 * lh) * wq + wcol;
      wb[buf][u][0] = __builtin_amdgcn_raw_buffer_load_b128(rsW, (live && nok) ? wo : FOOB, 0, 0);
      wb[buf][u][1] = __builtin_amdgcn_raw_buffer_load_b128(rsW, (live && nok) ? wo + wq : FOOB, 0, 0);
      m |= ok ? (1u << u) : 0u;
    }
    okm[buf] = m;
  };

  f32x16 acc, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f, acc2[r] = 0.f;
  float* cst = sm + SRED + wave * FCONST_PER_WAVE;
  float* tab = sm + SRED + SWAVES * FCONST_PER_WAVE;
  auto compute_group = [&](int buf, int v0) {
#pragma unroll
    for (int u = 0; u < FU; ++u) {
      fu32x4 x0 = xb[buf][u][0], x1 = xb[buf][u][1];
      if constexpr (AFF) {                               // BatchNorm affine of the input; zero padding applies AFTER it
        const int cc = (s_begin + v0 + u) & (cpk - 1);
        const float* sc = tab + cc * 16 + lh * 8;
        const float* sh = sc + a.Cin;
        const float4 s0v = *reinterpret_cast<const float4*>(sc), s1v = *reinterpret_cast<const float4*>(sc + 4);
        const float4 h0v = *reinterpret_cast<const float4*>(sh), h1v = *reinterpret_cast<const float4*>(sh + 4);
        const bool ok = (okm[buf] >> u) & 1u;
        x0.x = ok ? __float_as_uint(fmaf(__uint_as_float(x0.x), s0v.x, h0v.x)) : 0u;
        x0.y = ok ? __float_as_uint(fmaf(__uint_as_float(x0.y), s0v.y, h0v.y)) : 0u;
        x0.z = ok ? __float_as_uint(fmaf(__uint_as_float(x0.z), s0v.z, h0v.z)) : 0u;
        x0.w = ok ? __float_as_uint(fmaf(__uint_as_float(x0.w), s0v.w, h0v.w)) : 0u;
        x1.x = ok ? __float_as_uint(fmaf(__uint_as_float(x1.x), s1v.x, h1v.x)) : 0u;
        x1.y = ok ? __float_as_uint(fmaf(__uint_as_float(x1.y), s1v.y, h1v.y)) : 0u;
        x1.z = ok ? __float_as_uint(fmaf(__uint_as_float(x1.z), s1v.z, h1v.z)) : 0u;
        x1.w = ok ? __float_as_uint(fmaf(__uint_as_float(x1.w), s1v.w, h1v.w)) : 0u;
      }
      fmma<MATH>(acc, wb[buf][u][0], wb[buf][u][1], x0, x1, w_split);
    }
  };

  // ---- epilogue constants of this wave's 32 columns and (AFF) the scale / shift table: requested first, copied to LDS while the
  //      fragments behind them land
  const bool sliced = p.splits > 1;
  const bool scat = a.scatter2x2 != 0;
  const float* asc_p = PAIR ? p.add_scale : a.add_scale;
  const float* ash_p = PAIR ? p.add_shift : a.add_shift;
  float k0 = 0.f, k1 = 1.f, k2 = 0.f, k3 = 0.f, k4 = 1.f, k5 = 0.f;
  {
    const int c = ni * 32 + li;
    const bool on = tok && c < a.Ntot && lh == 0;
    const int cco = scat ? c % p.Cout : c;
    if (on && a.bias != nullptr) k0 = a.bias[cco];
    if (on && asc_p != nullptr) k1 = asc_p[c], k2 = ash_p[c];
    if (PAIR && on && p.bias3 != nullptr) k3 = p.bias3[c];
    if (!PAIR && on && a.out_scale != nullptr) k4 = a.out_scale[cco], k5 = a.out_shift[cco];
  }
  float tsc[TABQ], tsh[TABQ];
  if constexpr (AFF) {
#pragma unroll
    for (int e = 0; e < TABQ; ++e) {
      tsc[e] = 1.f;
      tsh[e] = 0.f;
      const int c = tid + e * 64 * SWAVES;
      if (c < a.Cin) {
        tsc[e] = a.in_scale[c];
        tsh[e] = a.in_shift[c];
      }
    }
  }
  load_group(0, 0);
  if (FU < n1) load_group(1, FU);
  auto owns = [&](int g) { return ksplit == 8 ? ((kw >> 1) == g && (kw & 1) == 0) : ((g & (ksplit - 1)) == kw); };
  float4 addv[4];
  float x3v = 0.f;
  float4 w3v[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c = ni * 32 + 8 * g + 4 * lh;
    const bool on = !sliced && owns(g) && pok && c < a.Ntot;
    addv[g] = (!PAIR && on && a.add != nullptr) ? *reinterpret_cast<const float4*>(a.add + ((int64_t)pix * a.ldadd + c)) : make_float4(0.f, 0.f, 0.f, 0.f);
    w3v[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PAIR && p.x3_one && on) {                        // quad-packed weights of a 1-channel 1x1 window: w[n][0]
      const float* wq3 = p.w3 + (int64_t)c * 4;
      w3v[g] = make_float4(wq3[0], wq3[4], wq3[8], wq3[12]);
    }
  }
  if (PAIR && p.x3_one && pok) x3v = p.x3[(int64_t)pix * p.ldx3];
  if (lh == 0) {
    cst[li] = k0;
    cst[32 + li] = k1;
    cst[64 + li] = k2;
    cst[96 + li] = k3;
    cst[128 + li] = k4;
    cst[160 + li] = k5;
  }
  if constexpr (AFF) {
#pragma unroll
    for (int e = 0; e < TABQ; ++e) {
      const int c = tid + e * 64 * SWAVES;
      if (c < a.Cin) {
        tab[c] = tsc[e];
        tab[a.Cin + c] = tsh[e];
      }
    }
    __syncthreads();
  }

  // ---- k-steps of the first product
  {
    int v0 = 0;
    while (true) {
      compute_group(0, v0);
      v0 += FU;
      if (v0 >= n1) break;
      if (v0 + FU < n1) load_group(0, v0 + FU);
      compute_group(1, v0);
      v0 += FU;
      if (v0 >= n1) break;
      if (v0 + FU < n1) load_group(1, v0 + FU);
    }
  }
  // ---- second product of a pair: its k-steps, shared out over all slices' waves like the first one's, in a second pass through the
  //      same registers (one more load round; the launch it replaces costs more)
  if constexpr (PAIR) {
    if (!p.x3_one) {
      __amdgpu_buffer_rsrc_t rsX3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x3), 0, (int)p.x3_bytes, 0x00020000);
      __amdgpu_buffer_rsrc_t rsW3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w3), 0, (int)p.w3_bytes, 0x00020000);
      const int t_begin = slice * p.kper2;
      const int n2 = max(0, min(p.ksteps2, t_begin + p.kper2) - t_begin);
      const uint32_t x3off = (uint32_t)pix * (uint32_t)p.ldx3 * 4u + lh32;
      const bool w3_split = p.w3_split != 0;
      for (int v0 = 0; v0 < n2; v0 += FU) {
#pragma unroll
        for (int u = 0; u < FU; ++u) {
          const int t = t_begin + v0 + u;
          const bool live = v0 + u < n2;
          const uint32_t xo = x3off + (uint32_t)(t * 64);
          xb[0][u][0] = __builtin_amdgcn_raw_buffer_load_b128(rsX3, (pok && live) ? xo : FOOB, 0, 0);
          xb[0][u][1] = __builtin_amdgcn_raw_buffer_load_b128(rsX3, (pok && live) ? xo + 16u : FOOB, 0, 0);
          const uint32_t wo = (uint32_t)(t * 4 + 2 * lh) * wq + wcol;
          wb[0][u][0] = __builtin_amdgcn_raw_buffer_load_b128(rsW3, (live && nok) ? wo : FOOB, 0, 0);
          wb[0][u][1] = __builtin_amdgcn_raw_buffer_load_b128(rsW3, (live && nok) ? wo + wq : FOOB, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < FU; ++u) fmma<MATH>(acc2, wb[0][u][0], wb[0][u][1], xb[0][u][0], xb[0][u][1], w3_split);
      }
    }
  }

  // ---- the waves of a tile add up through LDS (fixed order), each finishing its share of the accumulator groups
  if (ksplit > 1) {
    float* mine = sm + (wave * NACC * 16) * 64 + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[r * 64] = acc[r];
    if constexpr (PAIR) {
#pragma unroll
      for (int r = 0; r < 16; ++r) mine[(16 + r) * 64] = acc2[r];
    }
    __syncthreads();
  }
  auto finish_group = [&](int g, float4 v, float4 v2) {
    const int c = ni * 32 + 8 * g + 4 * lh;
    if (!pok || c >= a.Ntot) return;
    if (sliced) {
      *reinterpret_cast<float4*>(a.partial + ((int64_t)blockIdx.y * p.Mtot + pix) * a.Ntot + c) = v;
      if constexpr (PAIR) *reinterpret_cast<float4*>(a.partial + ((int64_t)(p.splits + blockIdx.y) * p.Mtot + pix) * a.Ntot + c) = v2;
      return;
    }
    const float4 cb = *reinterpret_cast<const float4*>(cst + 8 * g + 4 * lh);
    const float4 cs = *reinterpret_cast<const float4*>(cst + 32 + 8 * g + 4 * lh);
    const float4 ch = *reinterpret_cast<const float4*>(cst + 64 + 8 * g + 4 * lh);
    v.x += cb.x; v.y += cb.y; v.z += cb.z; v.w += cb.w;
    if (a.relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    if (!PAIR && a.add != nullptr) {
      v.x += fmaf(addv[g].x, cs.x, ch.x);
      v.y += fmaf(addv[g].y, cs.y, ch.y);
      v.z += fmaf(addv[g].z, cs.z, ch.z);
      v.w += fmaf(addv[g].w, cs.w, ch.w);
    }
    int64_t opix = pix;
    int ocol = c;
    if (scat) {
      const int ab = c / p.Cout;
      ocol = c - ab * p.Cout;
      opix = ((int64_t)img * a.Hout + 2 * gy + (ab >> 1)) * a.Wout + 2 * gx + (ab & 1);
    }
    float* dst = a.y + (opix * a.ldy + ocol);
    if (a.accumulate) {
      const float4 o = *reinterpret_cast<const float4*>(dst);
      v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
    }
    if (!PAIR && a.out_scale != nullptr) {               // the consumer's BatchNorm: fma(v, scale, shift), what its affine on load computes
      const float4 os = *reinterpret_cast<const float4*>(cst + 128 + 8 * g + 4 * lh);
      const float4 oh = *reinterpret_cast<const float4*>(cst + 160 + 8 * g + 4 * lh);
      v.x = fmaf(v.x, os.x, oh.x); v.y = fmaf(v.y, os.y, oh.y); v.z = fmaf(v.z, os.z, oh.z); v.w = fmaf(v.w, os.w, oh.w);
    }
    *reinterpret_cast<float4*>(dst) = v;
    if constexpr (PAIR) {
      const float4 c3 = *reinterpret_cast<const float4*>(cst + 96 + 8 * g + 4 * lh);
      float4 o;
      if (p.x3_one) {                                    // (direct_conv_kernel<1,1,1>: acc = bias; acc = fma(x, w, acc))
        o.x = fmaf(x3v, w3v[g].x, c3.x); o.y = fmaf(x3v, w3v[g].y, c3.y); o.z = fmaf(x3v, w3v[g].z, c3.z); o.w = fmaf(x3v, w3v[g].w, c3.w);
      } else {
        o.x = v2.x + c3.x; o.y = v2.y + c3.y; o.z = v2.z + c3.z; o.w = v2.w + c3.w;
      }
      o.x += fmaf(v.x, cs.x, ch.x);
      o.y += fmaf(v.y, cs.y, ch.y);
      o.z += fmaf(v.z, cs.z, ch.z);
      o.w += fmaf(v.w, cs.w, ch.w);
      *reinterpret_cast<float4*>(p.y2 + ((int64_t)pix * p.ldy2 + c)) = o;
    }
  };
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ksplit == 1) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
      finish_group(g, make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]),
                   PAIR ? make_float4(acc2[4 * g], acc2[4 * g + 1], acc2[4 * g + 2], acc2[4 * g + 3]) : zero4);
  } else if (ksplit <= 4) {
    const float* base = sm + ((grp << ksh) * NACC * 16) * 64 + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if ((g & (ksplit - 1)) != kw) continue;
      float v[4] = {0.f, 0.f, 0.f, 0.f}, v2[4] = {0.f, 0.f, 0.f, 0.f};
      for (int w_ = 0; w_ < ksplit; ++w_) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += base[(w_ * NACC * 16 + 4 * g + j) * 64];
        if constexpr (PAIR) {
#pragma unroll
          for (int j = 0; j < 4; ++j) v2[j] += base[(w_ * NACC * 16 + 16 + 4 * g + j) * 64];
        }
      }
      finish_group(g, make_float4(v[0], v[1], v[2], v[3]), make_float4(v2[0], v2[1], v2[2], v2[3]));
    }
  } else {
    const float* base = sm + lane;
    const int g = kw >> 1, half = kw & 1;
    float v[4] = {0.f, 0.f, 0.f, 0.f}, v2[4] = {0.f, 0.f, 0.f, 0.f};
    for (int w_ = 4 * half; w_ < 4 * half + 4; ++w_) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += base[(w_ * NACC * 16 + 4 * g + j) * 64];
      if constexpr (PAIR) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v2[j] += base[(w_ * NACC * 16 + 16 + 4 * g + j) * 64];
      }
    }
    __syncthreads();
    if (half) {
#pragma unroll
      for (int j = 0; j < 4; ++j) sm[(g * 8 + j) * 64 + lane] = v[j];
      if constexpr (PAIR) {
#pragma unroll
        for (int j = 0; j < 4; ++j) sm[(g * 8 + 4 + j) * 64 + lane] = v2[j];
      }
    }
    __syncthreads();
    if (!half) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += sm[(g * 8 + j) * 64 + lane];
      if constexpr (PAIR) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v2[j] += sm[(g * 8 + 4 + j) * 64 + lane];
      }
#pragma unroll
      for (int gg = 0; gg < 4; ++gg)
        if (gg == g) finish_group(gg, make_float4(v[0], v[1], v[2], v[3]), make_float4(v2[0], v2[1], v2[2], v2[3]));
    }
  }
}

// K-sliced pairs: the pair epilogue on the sums of both products' slices (a.partial: [2][splits][M][Ntot]); one thread: 4 channels of a pixel
__global__ void __launch_bounds__(256) convs32_pair_finish_kernel(const ConvS32 p) {
  const dfl_conv_args& a = p.a;
  const int nq = a.Ntot >> 2;
  const int idx = (int)blockIdx.x * 256 + (int)threadIdx.x;
  if (idx >= p.Mtot * nq) return;
  const int pix = idx / nq, c = (idx - pix * nq) * 4;
  const int64_t slice = (int64_t)p.Mtot * a.Ntot;
  const float* p1 = a.partial + (int64_t)pix * a.Ntot + c;
  const float* p2 = p1 + (int64_t)p.splits * slice;
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f), o = v;
  for (int s = 0; s < p.splits; ++s) {
    const float4 t1 = *reinterpret_cast<const float4*>(p1 + (int64_t)s * slice), t2 = *reinterpret_cast<const float4*>(p2 + (int64_t)s * slice);
    v.x += t1.x; v.y += t1.y; v.z += t1.z; v.w += t1.w;
    o.x += t2.x; o.y += t2.y; o.z += t2.z; o.w += t2.w;
  }
  if (a.bias != nullptr) { v.x += a.bias[c]; v.y += a.bias[c + 1]; v.z += a.bias[c + 2]; v.w += a.bias[c + 3]; }
  if (a.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
  *reinterpret_cast<float4*>(a.y + ((int64_t)pix * a.ldy + c)) = v;
  if (p.bias3 != nullptr) { o.x += p.bias3[c]; o.y += p.bias3[c + 1]; o.z += p.bias3[c + 2]; o.w += p.bias3[c + 3]; }
  o.x += fmaf(v.x, p.add_scale[c], p.add_shift[c]);
  o.y += fmaf(v.y, p.add_scale[c + 1], p.add_shift[c + 1]);
  o.z += fmaf(v.z, p.add_scale[c + 2], p.add_shift[c + 2]);
  o.w += fmaf(v.w, p.add_scale[c + 3], p.add_shift[c + 3]);
  *reinterpret_cast<float4*>(p.y2 + ((int64_t)pix * p.ldy2 + c)) = o;
}

// ---- host side ------------------------------------------------------------------------------------------------------

static bool convs32_switch() {
  static const bool on = [] {
    const char* e = getenv("DFL_CONVS");               // 0: the latency form is never taken (A/B against the GEMM kernels)
    return e == nullptr || atoi(e) != 0;
  }();
  return on;
}

// Validates like dfl::prepare of conv_gemm.hip does for what it uses, and plans; false = not a problem for this form
static bool convs32_plan(const dfl_conv_args* a, ConvS32* p, int force_splits) {
  if (a == nullptr || !a->latency_form || !convs32_switch() || a->x_bf16 || a->y_bf16) return false;
  const int mm = math_mode();
  if (mm != 0 && mm != 1) return false;
  if (a->x == nullptr || a->w == nullptr || a->y == nullptr || a->N <= 0 || a->Hin <= 0 || a->Win <= 0 || a->Cin <= 0 || a->Ntot <= 0) return false;
  if (a->KH <= 0 || a->KW <= 0 || a->KH * a->KW > 16 || a->stride <= 0 || a->pad < 0) return false;
  if (a->x_split || a->x_mode != 0 || a->x_out != nullptr || a->stat_partials != nullptr || a->stat_totals != nullptr || a->stat_other != nullptr) return false;
  if (a->in_tot != nullptr || a->add_tot != nullptr) return false;
  if ((a->in_scale == nullptr) != (a->in_shift == nullptr) || (a->add_scale == nullptr) != (a->add_shift == nullptr)) return false;
  if ((a->out_scale == nullptr) != (a->out_shift == nullptr) || (a->out_scale != nullptr && a->accumulate)) return false;
  if (a->w_split != 0 && !(a->w_split == 1 && mm == 1)) return false;
  const int cpk = a->Cin / 16;
  if (a->Cin % 16 != 0 || cpk < 1 || (cpk & (cpk - 1)) != 0 || a->Cin > 1024) return false;
  if (a->ldx % 4 != 0 || a->ldx < a->Cin || !aligned16(a->x) || !aligned16(a->w) || a->ldy % 4 != 0 || !aligned16(a->y) || a->Ntot % 8 != 0) return false;
  if (a->add != nullptr && (a->ldadd % 4 != 0 || !aligned16(a->add))) return false;
  memset(p, 0, sizeof(*p));
  p->a = *a;
  p->math = mm;
  if (a->scatter2x2) {
    if (a->KH != 1 || a->KW != 1 || a->stride != 1 || a->pad != 0 || a->Ntot % 4 != 0 || a->Hout < 2 * a->Hin || a->Wout < 2 * a->Win || a->add != nullptr) return false;
    p->Hg = a->Hin;
    p->Wg = a->Win;
    p->Cout = a->Ntot / 4;
  } else {
    const int ho = (a->Hin + 2 * a->pad - a->KH) / a->stride + 1, wo = (a->Win + 2 * a->pad - a->KW) / a->stride + 1;
    if (ho != a->Hout || wo != a->Wout) return false;
    p->Hg = a->Hout;
    p->Wg = a->Wout;
    p->Cout = a->Ntot;
  }
  if (a->ldy < p->Cout || p->Cout % 4 != 0) return false;
  const int64_t M = (int64_t)a->N * p->Hg * p->Wg;
  if (M > (1 << 16)) return false;
  p->Mtot = (int)M;
  p->T = a->KH * a->KW;
  if (2.0 * M * a->Ntot * (double)(p->T * a->Cin) > 1.5e9) return false;
  const int64_t xb = (((int64_t)a->N * a->Hin * a->Win - 1) * a->ldx + a->Cin) * 4;
  const int64_t wb = (int64_t)(p->T * a->Cin / 4) * a->Ntot * 16;
  const int64_t lim = (1ll << 31) - 4096;
  if (xb >= lim || wb >= lim) return false;
  p->x_bytes = (uint32_t)xb;
  p->w_bytes = (uint32_t)wb;
  // work split (convs_bf16.hip: convs_plan): 8 k-steps in flight per wave here
  p->mt = (int)ceil_div(M, 32);
  p->nt = (int)ceil_div(a->Ntot, 32);
  int sh = 0;
  while ((1 << sh) < cpk) ++sh;
  p->cpk_shift = sh;
  p->ksteps = p->T * cpk;
  const int tiles = p->mt * p->nt;
  int want = 2048 / tiles;
  if (want > p->ksteps / 4) want = p->ksteps / 4;
  if (want < 1) want = 1;
  int zs = 1;
  if (force_splits > 0) zs = force_splits;
  else if (want > 8 && p->ksteps > 8 * 2 * FU) zs = want / 8;
  if (zs > 16) zs = 16;
  if (zs > p->ksteps) zs = p->ksteps;
  const int per = want / zs;
  int ksh = 0;
  while (ksh < 3 && (2 << ksh) <= per) ++ksh;
  p->ksplit_shift = ksh;
  p->kper = (int)ceil_div(p->ksteps, (int64_t)zs << ksh);
  if (force_splits <= 0) zs = (int)ceil_div(p->ksteps, (int64_t)p->kper << ksh);
  p->splits = zs;
  p->waves = ksh == 3 ? 8 : 4;
  p->grid = (int)ceil_div(tiles, p->waves >> ksh);
  return true;
}

bool convs32_eligible(const dfl_conv_args* a) {
  ConvS32 p;
  return convs32_plan(a, &p, a != nullptr && a->splits > 1 ? a->splits : 1);
}

int convs32_suggest_splits(const dfl_conv_args* a) {
  ConvS32 p;
  return convs32_plan(a, &p, 0) ? p.splits : 0;
}

template <int MATH, bool PAIR>
static int convs32_launch_t(const ConvS32& p, hipStream_t s) {
  dim3 grid((unsigned)p.grid, (unsigned)p.splits);
  const size_t lds = (size_t)(p.waves * (PAIR ? 2 : 1) * 16 * 64 + p.waves * FCONST_PER_WAVE + 2 * p.a.Cin) * 4;
  const bool aff = p.a.in_scale != nullptr;
#define DFL_CS32(AFF_, W_)                                                                                                         \
  {                                                                                                                                \
    auto k = convs32_kernel<MATH, AFF_, PAIR, W_>;                                                                                 \
    DFL_LDS_OPT_IN(k, 96 * 1024, "dfl_conv2d (fp32, latency form)") \
    hipLaunchKernelGGL(k, grid, dim3(64 * W_), lds, s, p);                                                                         \
  }
  if (p.waves == 8) {
    if (aff) DFL_CS32(true, 8) else DFL_CS32(false, 8)
  } else {
    if (aff) DFL_CS32(true, 4) else DFL_CS32(false, 4)
  }
#undef DFL_CS32
  return DFL_OK;
}

// The convolution itself; *splits_out = K slices the caller has to finish (conv_finish_kernel of conv_gemm.hip)
int convs32_launch(const dfl_conv_args* a, hipStream_t s, int* splits_out) {
  ConvS32 p;
  const bool ok = convs32_plan(a, &p, a->splits > 1 ? a->splits : 1);
  DFL_REQUIRE(ok, "dfl_conv2d (fp32 tensors, latency form): not eligible");
  DFL_REQUIRE(p.splits <= 1 || a->partial != nullptr, "dfl_conv2d: splits > 1 needs the partial buffer");
  const int rc = p.math == 0 ? convs32_launch_t<0, false>(p, s) : convs32_launch_t<1, false>(p, s);
  if (rc != DFL_OK) return rc;
  *splits_out = p.splits;
  return check_launch("dfl_conv2d (fp32 tensors, latency form)");
}

static bool pair32_ok(const dfl_conv_args* a, const dfl_conv_args* b, const ConvS32& pa) {
  if (pa.splits > 1 && (b->Cin == 1 || a->partial == nullptr)) return false;
  if (a->scatter2x2 || a->accumulate || a->add != nullptr || a->out_scale != nullptr) return false;
  if (!b->latency_form || b->x_bf16 || b->y_bf16 || b->KH != 1 || b->KW != 1 || b->stride != 1 || b->pad != 0 || b->scatter2x2 || b->accumulate || b->relu) return false;
  if (b->in_scale != nullptr || b->in_tot != nullptr || b->add_tot != nullptr || b->x_mode != 0 || b->x_out != nullptr || b->out_scale != nullptr || b->x_split) return false;
  if (b->stat_partials != nullptr || b->stat_totals != nullptr || b->stat_other != nullptr || b->splits > 1) return false;
  if (b->add != a->y || b->ldadd != a->ldy || b->add_scale == nullptr || b->add_shift == nullptr) return false;
  if (b->N != a->N || b->Hout != a->Hout || b->Wout != a->Wout || b->Hin != a->Hout || b->Win != a->Wout || b->Ntot != a->Ntot) return false;
  if (b->y == a->y || b->y == nullptr || b->ldy % 4 != 0 || !aligned16(b->y) || b->x == nullptr || b->w == nullptr) return false;
  if (b->Cin == 1) return b->w_split == 0;
  if (b->Cin % 16 != 0 || b->ldx % 4 != 0 || !aligned16(b->x) || !aligned16(b->w)) return false;
  if (b->w_split != 0 && !(b->w_split == 1 && pa.math == 1)) return false;
  const int64_t xb = (((int64_t)b->N * b->Hin * b->Win - 1) * b->ldx + b->Cin) * 4, wb = (int64_t)(b->Cin / 4) * b->Ntot * 16;
  return xb < (1ll << 31) - 4096 && wb < (1ll << 31) - 4096;
}

int convs32_pair_ok(const dfl_conv_args* a, const dfl_conv_args* b) {
  ConvS32 pa;
  if (a == nullptr || b == nullptr || !convs32_plan(a, &pa, a->splits > 1 ? a->splits : 1)) return 0;
  if (!pair32_ok(a, b, pa)) return 0;
  return pa.splits > 1 ? 2 : 1;
}

int convs32_pair_launch(const dfl_conv_args* a, const dfl_conv_args* b, hipStream_t s) {
  ConvS32 p;
  const bool ok = convs32_plan(a, &p, a->splits > 1 ? a->splits : 1) && pair32_ok(a, b, p);
  DFL_REQUIRE(ok, "dfl_conv2d_pair (fp32 tensors): these two convolutions do not form a pair");
  p.x3 = b->x;
  p.w3 = b->w;
  p.bias3 = b->bias;
  p.add_scale = b->add_scale;
  p.add_shift = b->add_shift;
  p.y2 = b->y;
  p.ldx3 = b->ldx;
  p.ldy2 = b->ldy;
  p.x3_one = b->Cin == 1 ? 1 : 0;
  p.w3_split = b->w_split;
  if (!p.x3_one) {
    p.ksteps2 = b->Cin / 16;
    p.kper2 = (int)ceil_div(p.ksteps2, (int64_t)p.splits << p.ksplit_shift);
    p.x3_bytes = (uint32_t)((((int64_t)b->N * b->Hin * b->Win - 1) * b->ldx + b->Cin) * 4);
    p.w3_bytes = (uint32_t)((int64_t)(b->Cin / 4) * b->Ntot * 16);
  }
  const int rc = p.math == 0 ? convs32_launch_t<0, true>(p, s) : convs32_launch_t<1, true>(p, s);
  if (rc != DFL_OK) return rc;
  if (p.splits > 1) {
    const int rc = check_launch("dfl_conv2d_pair (fp32 tensors)");
    if (rc != DFL_OK) return rc;
    hipLaunchKernelGGL(convs32_pair_finish_kernel, dim3((unsigned)ceil_div((int64_t)p.Mtot * (a->Ntot / 4), 256)), dim3(256), 0, s, p);
  }
  return check_launch("dfl_conv2d_pair (fp32 tensors)");
}

}  // namespace dfl
