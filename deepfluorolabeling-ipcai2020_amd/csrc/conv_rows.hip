// Row-tiled 3x3 convolution (stride 1, pad 1) for the wide, narrow-channel levels of the U-Net.
//
// Same GEMM view, operands, arithmetic modes and epilogue as conv_gemm.hip (reference: nn.Conv2d 3x3 of
// train_test_code/unet.py:207-218 and its data gradient), but the tile is BM consecutive pixels of ONE image row
// (W % BM == 0) and K is walked as (dy, channel chunk, dx): the BM + 2 pixels x 16 channels a kernel row needs are staged
// in LDS once and the three dx taps read them at row offsets 0 / 1 / 2, instead of being gathered three times.  With 32 or
// 64 output channels the gathered operand dominates the load path (the loop's most contended resource, DESIGN.md
// section 5), so this removes ~40 % of the vector loads and LDS writes per matrix instruction.  Tiles: 192 x 32 and
// 96 x 64 (three waves): at 192 x 192 x 16 that is 3072 / 1536 workgroups = exactly 12 / 6 per CU.
//
// LDS: two stages of [A: 2 planes x (BM + 4) rows of 32 bytes][B: 3 taps x 2 planes x BN rows], layout and unit swap as
// in conv_gemm.hip (the swap by bit 3 of the row stays conflict-free for rows shifted by 1 and 2).  One register set:
// an iteration carries 9 matrix-instruction groups per tile, enough to cover the loads of the next one.
#include <stdlib.h>

#include "common.h"
#include "conv_epilogue.h"
#include "conv_rows.h"

namespace dfl {

template <int WM, int WN, int TM, int TN, bool AFF, int MATH>
__global__ void __launch_bounds__(WM* WN * 64, 4) conv_rows_kernel(const ConvK p) {
  static_assert(MATH == 0 || MATH == 3 || MATH == 4, "fp32, or bf16x3 with pre-split weights (3) / both operands (4)");
  constexpr bool XPRE = (MATH == 4);
  static_assert(!(XPRE && AFF), "a split input cannot take an affine on load");
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int EA = BM + 4;                               // staged rows: BM + 2, rounded up to whole 4-row write groups
  constexpr int PLA = EA * 8 + 16, PLB = BN * 8 + 16;      // plane strides (words)
  constexpr int IMA = 2 * PLA, IMB = 2 * PLB;
  constexpr int STAGE = IMA + 3 * IMB;
  constexpr int RPP = NT / 4;
  static_assert(RPP % 16 == 0 && BM % RPP == 0 && (BM & 15) == 0, "row mapping");
  constexpr int QA = BM / RPP;
  constexpr int NQB = 4 * BN;
  constexpr int QB = (NQB + NT - 1) / NT;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  const dfl_conv_args& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int H = a.Hin, W = a.Win, Cin = a.Cin, Ntot = a.Ntot;
  const int aq = tid & 3;
  float* Ssc = smem + 2 * STAGE;   // AFF: [Cin] scale, [Cin] shift
  float* Ssh = Ssc + Cin;

  // the tile: pixels x0 .. x0 + BM - 1 of image row (n, y)
  const int x0 = m0 % W;
  const int yn = m0 / W;
  const int y = yn % H;

  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)p.w_bytes, 0x00020000);

  // staged row e <-> input pixel x0 - 1 + e of row y + dy - 1.  Rows 0 .. BM-1 by the regular (row, quad) mapping, rows
  // BM and BM + 1 by the first 8 threads.
  uint32_t a_rowb[QA + 1];
  uint32_t xokm = 0;
#pragma unroll
  for (int r = 0; r <= QA; ++r) {
    const int e = (r < QA) ? (tid >> 2) + r * RPP : BM + (tid >> 2);
    const int x = x0 - 1 + e;
    const bool ok = (r < QA || tid < 8) && (unsigned)x < (unsigned)W;
    xokm |= ok ? (1u << r) : 0u;
    a_rowb[r] = (uint32_t)((((int64_t)(yn - 1) * W + x) * a.ldx + 4 * aq) * 4);
  }
  uint32_t b_voff[QB];
#pragma unroll
  for (int r = 0; r < QB; ++r) {
    const int idx = tid + r * NT;
    const int kq = (idx >> 2) & 3, nn = ((idx >> 4) << 2) | (idx & 3);
    b_voff[r] = (idx < NQB && n0 + nn < Ntot) ? (uint32_t)(((int64_t)kq * Ntot + n0 + nn) * 16) : OOB;
  }
  if constexpr (AFF) {
    for (int c = tid; c < Cin; c += NT) {
      Ssc[c] = a.in_scale[c];
      Ssh[c] = a.in_shift[c];
    }
    __syncthreads();
  }

  const int nit = 3 * (Cin / KC);   // (dy, channel chunk) pairs; even (Cin % 32 == 0)
  float4 ra[QA + 1];
  float4 rb[3][QB];
  float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t okA = 0;
  int cur_dy = 0, cur_c0 = 0;       // wave-uniform cursor of the iteration to load next

  auto load = [&](int it) {
    const bool live = it < nit;
    const bool yok = live && (unsigned)(y + cur_dy - 1) < (unsigned)H;
    const uint32_t tapb = (uint32_t)((cur_dy * W * a.ldx + cur_c0) * 4);
    okA = yok ? xokm : 0u;
#pragma unroll
    for (int r = 0; r <= QA; ++r) ra[r] = buf_load4(rsA, ((okA >> r) & 1u) ? a_rowb[r] + tapb : OOB, 0);
    if constexpr (AFF) {
      const int cofs = (live ? cur_c0 : 0) + 4 * aq;
      sc4 = *reinterpret_cast<const float4*>(Ssc + cofs);
      sh4 = *reinterpret_cast<const float4*>(Ssh + cofs);
    }
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const uint32_t soff = live ? (uint32_t)(((cur_dy * 3 + dx) * Cin + cur_c0) >> 2) * 16u * (uint32_t)Ntot : 0u;
#pragma unroll
      for (int r = 0; r < QB; ++r) rb[dx][r] = buf_load4(rsB, live ? b_voff[r] : OOB, soff);
    }
    cur_c0 += KC;
    if (cur_c0 >= Cin) {
      cur_c0 = 0;
      ++cur_dy;
    }
  };

  auto store = [&](int stage) {
    float* Ab = smem + stage * STAGE;
#pragma unroll
    for (int r = 0; r <= QA; ++r) {
      if (r == QA && tid >= 8) break;
      const int e = (r < QA) ? (tid >> 2) + r * RPP : BM + (tid >> 2);
      const int fA = (e >> 3) & 1;
      float4 v = ra[r];
      if constexpr (AFF) {   // zero padding applies AFTER the BatchNorm affine: the data already is 0 there, mask the shift
        const bool ok = (okA >> r) & 1u;
        v.x = fmaf(v.x, sc4.x, ok ? sh4.x : 0.f);
        v.y = fmaf(v.y, sc4.y, ok ? sh4.y : 0.f);
        v.z = fmaf(v.z, sc4.z, ok ? sh4.z : 0.f);
        v.w = fmaf(v.w, sc4.w, ok ? sh4.w : 0.f);
      }
      if constexpr (MATH != 0) {
        uint2 parts[2];
        if constexpr (XPRE) {
          parts[0] = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y));
          parts[1] = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w));
        } else {
          split_bf16<2>(v, parts);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) *reinterpret_cast<uint2*>(Ab + q * PLA + e * 8 + 4 * ((aq >> 1) ^ fA) + 2 * (aq & 1)) = parts[q];
      } else {
        *reinterpret_cast<float4*>(Ab + (aq >> 1) * PLA + e * 8 + 4 * ((aq & 1) ^ fA)) = v;
      }
    }
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      float* Bb = smem + stage * STAGE + IMA + dx * IMB;
#pragma unroll
      for (int r = 0; r < QB; ++r) {
        const int idx = tid + r * NT;
        if (NQB % NT == 0 || idx < NQB) {
          const int kq = (idx >> 2) & 3, nn = ((idx >> 4) << 2) | (idx & 3);
          const int fB = (nn >> 3) & 1;
          const float4 v = rb[dx][r];
          if constexpr (MATH != 0) {   // weights arrive as split quads (hi4 | lo4)
            *reinterpret_cast<uint2*>(Bb + nn * 8 + 4 * ((kq >> 1) ^ fB) + 2 * (kq & 1)) = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y));
            *reinterpret_cast<uint2*>(Bb + PLB + nn * 8 + 4 * ((kq >> 1) ^ fB) + 2 * (kq & 1)) = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w));
          } else {
            *reinterpret_cast<float4*>(Bb + (kq >> 1) * PLB + nn * 8 + 4 * ((kq & 1) ^ fB)) = v;
          }
        }
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int unB = 4 * (lh ^ ((li >> 3) & 1));
  auto compute = [&](int stage) {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int un = 4 * (lh ^ (((li + dx) >> 3) & 1));   // unit of row li + dx after the swap
      const float* Ab = smem + stage * STAGE + (wm * (TM * 32) + li + dx) * 8 + un;
      const float* Bb = smem + stage * STAGE + IMA + dx * IMB + (wn * (TN * 32) + li) * 8 + unB;
      if constexpr (MATH != 0) {
        bf16x8_t ap[TM][2], bp[TN][2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
#pragma unroll
          for (int i = 0; i < TM; ++i) ap[i][q] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Ab + i * 256 + q * PLA));
#pragma unroll
          for (int j = 0; j < TN; ++j) bp[j][q] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Bb + j * 256 + q * PLB));
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {   // small terms first
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][0], bp[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][1], bp[j][0], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][0], bp[j][0], acc[i][j], 0, 0, 0);
          }
      } else {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          float4 av[TM], bv[TN];
#pragma unroll
          for (int i = 0; i < TM; ++i) av[i] = *reinterpret_cast<const float4*>(Ab + i * 256 + g * PLA);
#pragma unroll
          for (int j = 0; j < TN; ++j) bv[j] = *reinterpret_cast<const float4*>(Bb + j * 256 + g * PLB);
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0);
            }
        }
      }
    }
  };

  load(0);
  store(0);
  __syncthreads();
  for (int it = 0; it < nit; it += 2) {   // two at a time: the stage is a compile-time constant
    load(it + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(0);
    __builtin_amdgcn_sched_barrier(0);
    store(1);
    __syncthreads();
    load(it + 2);
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    __builtin_amdgcn_sched_barrier(0);
    store(0);
    __syncthreads();
  }

  float s1[TN], s2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    s1[j] = 0.f;
    s2[j] = 0.f;
  }
  conv_epilogue_simple<WM, WN, TM, TN>(p, acc, s1, s2, m0, n0, wm, wn, li, lh);
  if (a.stat_partials != nullptr) conv_stats_tail<WM, WN, TM, TN>(p, s1, s2, smem, tid, n0, wm, wn, li, lh);
}

// ---- host side ------------------------------------------------------------------------------------------------------

static bool rows_enabled() {
  static const bool on = [] {
    const char* e = getenv("DFL_CONV_ROWS");
    return e == nullptr || atoi(e) != 0;
  }();
  return on;
}

// 0 = not eligible; otherwise the tile's pixel count (192: 192 x 32 tile, 96: 96 x 64 tile)
int conv_rows_tile(const ConvK& k) {
  const dfl_conv_args& a = k.a;
  if (!rows_enabled() || !k.fast) return 0;
  if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.scatter2x2 || a.add != nullptr || a.accumulate) return 0;
  if (a.splits > 1 || a.Cin % 32 != 0 || a.Hout != a.Hin || a.Wout != a.Win) return 0;
  const int mode = math_mode();
  if (mode == 0) {
    if (a.w_split || a.x_split) return 0;
  } else if (mode == 1) {
    if (a.w_split != 1) return 0;                        // bf16x3 with the weights split by the pack kernel
    if (a.x_split && a.in_scale != nullptr) return 0;
  } else {
    return 0;
  }
  if (a.in_scale != nullptr && (size_t)2 * a.Cin * sizeof(float) > 16 * 1024) return 0;
  if (a.Ntot <= 32 && a.Win % 192 == 0) return 192;
  if (a.Ntot <= 64 && a.Win % 96 == 0) return 96;
  return 0;
}

template <int WM, int WN, int TM, int TN, bool AFF, int MATH>
static int rows_launch(const ConvK& k, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int stage = 2 * ((BM + 4) * 8 + 16) + 3 * 2 * (BN * 8 + 16);
  size_t lds = (size_t)2 * stage * sizeof(float);
  if (AFF) lds += (size_t)2 * k.a.Cin * sizeof(float);
  dim3 grid((unsigned)(k.Mtot / BM), (unsigned)ceil_div(k.a.Ntot, BN), 1);
  hipLaunchKernelGGL((conv_rows_kernel<WM, WN, TM, TN, AFF, MATH>), grid, dim3(WM * WN * 64), lds, s, k);
  return check_launch("dfl_conv2d (row tiles)");
}

template <int WM, int WN, int TM, int TN>
static int rows_dispatch(const ConvK& k, hipStream_t s) {
  const bool aff = k.a.in_scale != nullptr;
  if (math_mode() == 0) return aff ? rows_launch<WM, WN, TM, TN, true, 0>(k, s) : rows_launch<WM, WN, TM, TN, false, 0>(k, s);
  if (k.a.x_split) return rows_launch<WM, WN, TM, TN, false, 4>(k, s);
  return aff ? rows_launch<WM, WN, TM, TN, true, 3>(k, s) : rows_launch<WM, WN, TM, TN, false, 3>(k, s);
}

int conv_rows_launch(const ConvK& k, hipStream_t s) {
  const int t = conv_rows_tile(k);
  DFL_REQUIRE(t != 0 && k.Mtot % t == 0, "dfl_conv2d: internal: row-tiled kernel called for an ineligible layer");
  return t == 192 ? rows_dispatch<3, 1, 2, 1>(k, s) : rows_dispatch<3, 1, 1, 2>(k, s);
}

}  // namespace dfl
