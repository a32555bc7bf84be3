// Row-tiled 3x3 convolution (stride 1, pad 1) for the wide, narrow-channel levels of the U-Net.
//
// Same GEMM view, operands and epilogue as conv_gemm.hip (reference: nn.Conv2d 3x3 of train_test_code/unet.py:207-218
// and its data gradient), split-bf16 products only (math modes 1 and 3: the modes whose loop is bound by the load path), but the tile is BM consecutive pixels that form whole row
// segments -- a piece of ONE image row (W % BM == 0) or R = BM / W complete image rows (BM % W == 0) -- and K is walked
// as (dy, channel chunk, dx): the (segment + 2) pixels x 16 channels a kernel row needs are staged in LDS once per
// segment and the three dx taps read them at row offsets 0 / 1 / 2, instead of being gathered three times.  With 32 or
// 64 output channels the gathered operand dominates the load path (the loop's most contended resource, DESIGN.md
// section 5), so this removes ~40 % of the vector loads and LDS writes per matrix instruction.  Tiles: 192 x 32 and
// 96 x 64 (three waves): every level of the 192 x 192 x 16 network divides into them -- 3072 / 1536 / 768 workgroups
// at levels 0 / 1 / 2, and 384 / 192 / 96 tiles x 2 / 4 / 8 K slices = 768 at levels 3-5 (slices = ranges of
// (dy, channel chunk) pairs, partial sums finished by conv_finish_kernel): whole multiples of the 256 CUs, where the
// 64 x 64 tiles of the generic kernel leave 4.5 workgroups per CU.
//
// LDS: two stages of [A: 2 planes x (BM + 4) rows of 32 bytes][B: 3 taps x 2 planes x BN rows], layout and unit swap as
// in conv_gemm.hip (the swap by bit 3 of the row stays conflict-free for rows shifted by 1 and 2).  One register set:
// an iteration carries 9 matrix-instruction groups per tile, enough to cover the loads of the next one.
#include <stdlib.h>

#include "common.h"
#include "conv_epilogue.h"
#include "conv_rows.h"

namespace dfl {

#ifndef DFL_ROWS_OCC
#define DFL_ROWS_OCC 4   // waves per SIMD the register allocation leaves room for (2 and 3 measured: equal within noise; 5 spills: -20 %)
#endif
template <int WM, int WN, int TM, int TN, bool AFF, int MATH>
__global__ void __launch_bounds__(WM* WN * 64, DFL_ROWS_OCC) conv_rows_kernel(const ConvK p) {
  // MATH: 3 = bf16x3, weights pre-split; 4 = bf16x3, both operands pre-split; 5 / 6 = the same two with plain bf16
  // products (math mode 3: hi parts only, one matrix instruction per 16 k-values)
  static_assert(MATH >= 3 && MATH <= 6, "bf16x3 or bf16 products with pre-split weights");
  constexpr bool XPRE = (MATH == 4 || MATH == 6);
  constexpr int NP = (MATH <= 4) ? 2 : 1;                  // bf16 parts per value that are staged and multiplied
  static_assert(!(XPRE && AFF), "a split input cannot take an affine on load");
  constexpr int NT = WM * WN * 64;
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  constexpr int PLB = BN * 8 + 16, IMB = 2 * PLB;          // plane stride / image of a weight chunk (words)
  constexpr int RPP = NT / 4;
  static_assert(RPP % 16 == 0 && BM % RPP == 0 && (BM & 15) == 0, "row mapping");
  constexpr int QA = BM / RPP;
  static_assert(RPP >= 32, "the extra pass must cover the 2 * 16 halo rows of the narrowest images");
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
  // the tile: R segments of Wt pixels = pixels x0 .. x0 + Wt - 1 of the image rows yn .. yn + R - 1 (row index over n * H)
  const int Wt = (W >= BM) ? BM : W;
  const int R = BM / Wt;
  const int E = BM + 2 * R;                                // staged rows: every segment with its two halo pixels
  const int PLA = ((E + 3) & ~3) * 8 + 16, IMA = 2 * PLA;  // plane stride / image of the staged pixels (words)
  const int STAGE = IMA + 3 * IMB;
  const int x0 = (W >= BM) ? m0 % W : 0;
  const int yn = m0 / W;
  float* Ssc = smem + 2 * STAGE;   // AFF: [Cin] scale, [Cin] shift
  float* Ssh = Ssc + Cin;

  __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)p.w_bytes, 0x00020000);

  // staged row e = s * (Wt + 2) + c <-> input pixel x0 - 1 + c of image row yn + s + dy - 1.  Rows by the regular
  // (row, quad) mapping; QA passes cover BM of them, one more pass the rest (at most 32).  okm: bit 3 r + dy.
  uint32_t a_rowb[QA + 1];
  uint32_t okm = 0;
#pragma unroll
  for (int r = 0; r <= QA; ++r) {
    const int e = (tid >> 2) + r * RPP;
    const int sg = e / (Wt + 2), x = x0 - 1 + (e - sg * (Wt + 2));
    const int yi = (yn + sg) % H;                          // row inside its image
    const bool ok = e < E && (unsigned)x < (unsigned)W;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) okm |= (ok && (unsigned)(yi + dy - 1) < (unsigned)H) ? (1u << (3 * r + dy)) : 0u;
    a_rowb[r] = (uint32_t)((((int64_t)(yn + sg - 1) * W + x) * a.ldx + 4 * aq) * 4);
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
      if (a.in_tot != nullptr) {        // live statistics (include/dfl_hip.h)
        bn_live_affine(a.in_tot, a.in_gamma, a.in_beta, a.in_count, a.bn_eps, Cin, c, Ssc + c, Ssh + c);
      } else {
        Ssc[c] = a.in_scale[c];
        Ssh[c] = a.in_shift[c];
      }
    }
    __syncthreads();
  }

  const int cpr = Cin / KC;                                 // channel chunks per kernel row
  const int it_begin = blockIdx.z * p.cps;                  // split-K: p.cps (dy, channel chunk) pairs per slice, an even number
  const int nit = min(it_begin + p.cps, 3 * cpr);
  float4 ra[QA + 1];
  float4 rb[3][QB];
  float4 sc4 = make_float4(1.f, 1.f, 1.f, 1.f), sh4 = make_float4(0.f, 0.f, 0.f, 0.f);
  uint32_t okA = 0;
  int cur_dy = it_begin / cpr, cur_c0 = (it_begin - cur_dy * cpr) * KC;   // wave-uniform cursor of the iteration to load next

  auto load = [&](int it) {
    const bool live = it < nit;
    const uint32_t tapb = (uint32_t)((cur_dy * W * a.ldx + cur_c0) * 4);
    okA = 0;
#pragma unroll
    for (int r = 0; r <= QA; ++r) okA |= (live && ((okm >> (3 * r + cur_dy)) & 1u)) ? (1u << r) : 0u;
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
      const int e = (tid >> 2) + r * RPP;
      if (r == QA && e >= E) break;
      const int fA = (e >> 3) & 1;
      float4 v = ra[r];
      if constexpr (AFF) {   // zero padding applies AFTER the BatchNorm affine: the data already is 0 there, mask the shift
        const bool ok = (okA >> r) & 1u;
        v.x = fmaf(v.x, sc4.x, ok ? sh4.x : 0.f);
        v.y = fmaf(v.y, sc4.y, ok ? sh4.y : 0.f);
        v.z = fmaf(v.z, sc4.z, ok ? sh4.z : 0.f);
        v.w = fmaf(v.w, sc4.w, ok ? sh4.w : 0.f);
      }
      uint2 parts[2];
      if constexpr (XPRE) {   // the producer already left hi4 | lo4 in the slot
        parts[0] = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y));
        parts[1] = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w));
      } else {
        split_bf16<NP>(v, parts);
      }
#pragma unroll
      for (int q = 0; q < NP; ++q) *reinterpret_cast<uint2*>(Ab + q * PLA + e * 8 + 4 * ((aq >> 1) ^ fA) + 2 * (aq & 1)) = parts[q];
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
          // weights arrive as split quads (hi4 | lo4)
          *reinterpret_cast<uint2*>(Bb + nn * 8 + 4 * ((kq >> 1) ^ fB) + 2 * (kq & 1)) = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y));
          if constexpr (NP > 1) *reinterpret_cast<uint2*>(Bb + PLB + nn * 8 + 4 * ((kq >> 1) ^ fB) + 2 * (kq & 1)) = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w));
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
  int erow[TM];                                            // staged row of this lane's pixel in tile i, tap dx = 0
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int t = wm * (TM * 32) + i * 32 + li, sg = t / Wt;
    erow[i] = t + 2 * sg;
  }
  auto compute = [&](int stage) {
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const float* Ab = smem + stage * STAGE;
      const float* Bb = smem + stage * STAGE + IMA + dx * IMB + (wn * (TN * 32) + li) * 8 + unB;
      int aoff[TM];                                        // row erow + dx, unit after the swap by bit 3 of the row
#pragma unroll
      for (int i = 0; i < TM; ++i) aoff[i] = (erow[i] + dx) * 8 + 4 * (lh ^ (((erow[i] + dx) >> 3) & 1));
      bf16x8_t ap[TM][2], bp[TN][2];
#pragma unroll
      for (int q = 0; q < NP; ++q) {
#pragma unroll
        for (int i = 0; i < TM; ++i) ap[i][q] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Ab + aoff[i] + q * PLA));
#pragma unroll
        for (int j = 0; j < TN; ++j) bp[j][q] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const uint4*>(Bb + j * 256 + q * PLB));
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {   // small terms first
          if constexpr (NP > 1) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][0], bp[j][1], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][1], bp[j][0], acc[i][j], 0, 0, 0);
          }
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[i][0], bp[j][0], acc[i][j], 0, 0, 0);
        }
    }
  };

  // Between the matrix instructions and the staging of the next iteration: a scheduling barrier when that staging
  // carries VALU work (affine, splitting) -- mixed into the matrix block it costs 10-15 % --, none when it is pure copies
  // of pre-split operands, which the scheduler then tucks between the matrix instructions (+1-5 %, measured).
#define ROWS_SB2()                                              \
  {                                                             \
    if constexpr (!XPRE) __builtin_amdgcn_sched_barrier(0);     \
  }
  load(it_begin);
  store(0);
  __syncthreads();
  for (int it = it_begin; it < nit; it += 2) {   // two at a time: the stage is a compile-time constant
    load(it + 1);
    __builtin_amdgcn_sched_barrier(0);
    compute(0);
    ROWS_SB2();
    store(1);
    __syncthreads();
    load(it + 2);
    __builtin_amdgcn_sched_barrier(0);
    compute(1);
    ROWS_SB2();
    store(0);
    __syncthreads();
  }

  if (p.splits > 1) {   // raw partial sums; conv_finish_kernel applies the epilogue
    float* part = a.partial + (int64_t)blockIdx.z * p.Mtot * Ntot;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int n = n0 + wn * (TN * 32) + j * 32 + li;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = m0 + wm * (TM * 32) + i * 32 + mfma32_row(r, lane);
          if (n < Ntot) part[(int64_t)m * Ntot + n] = acc[i][j][r];
        }
      }
    }
    return;
  }
  float s1[TN], s2[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    s1[j] = 0.f;
    s2[j] = 0.f;
  }
  conv_epilogue_simple<WM, WN, TM, TN>(p, acc, s1, s2, m0, n0, wm, wn, li, lh);
  if (a.stat_partials != nullptr || a.stat_totals != nullptr) conv_stats_tail<WM, WN, TM, TN>(p, s1, s2, smem, tid, n0, wm, wn, li, lh);
}

// ---- host side ------------------------------------------------------------------------------------------------------

static int g_rows_min_tiles = 512;   // fewer workgroups than this: the generic kernel (it can cut K) fills the chip better

static bool rows_enabled() {
  static const bool on = [] {
    const char* e = getenv("DFL_CONV_ROWS");
    return e == nullptr || atoi(e) != 0;
  }();
  return on;
}

// K slices for a layer with `tiles` row tiles: 1 when those fill the chip, else the smallest count that gives >= 768
// workgroups with an even number (>= 4) of (dy, channel chunk) pairs per slice; 0 = no such count
static int rows_splits(int tiles, int Cin) {
  if (tiles >= g_rows_min_tiles) return 1;
  const int nit = 3 * (Cin / KC);
  for (int s = 2; s <= 16; ++s)
    if (nit % (2 * s) == 0 && nit / s >= 4 && (int64_t)tiles * s >= 768) return s;
  return 0;
}

// 0 = not eligible; otherwise the tile (ROWS_192x32 / ROWS_96x64 / ROWS_96x32).  a.splits: 0 / 1 = asks
// what the kernel would choose (conv_rows_splits), > 1 = the caller's slice count must be one this kernel can take.
int conv_rows_tile(const ConvK& k) {
  const dfl_conv_args& a = k.a;
  if (!rows_enabled() || !k.fast) return 0;
  if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.scatter2x2) return 0;
  if (a.splits <= 1 && (a.add != nullptr || a.accumulate)) return 0;     // one pass: simple epilogue only (slices: the finish kernel has them all)
  if (a.Cin % 32 != 0 || a.Hout != a.Hin || a.Wout != a.Win) return 0;
  // bf16x3 / bf16 products with the weights split by the pack kernel.  (fp32 products are bound by the matrix pipe, not by
  // the load path: measured there, these tiles lose 3-5 % of the step against the generic kernel's, so that mode keeps it.)
  if ((math_mode() != 1 && math_mode() != 3) || a.w_split != 1) return 0;
  const bool has_aff = a.in_scale != nullptr || a.in_tot != nullptr;
  if (a.x_split && has_aff) return 0;
  if (has_aff && (size_t)2 * a.Cin * sizeof(float) > 16 * 1024) return 0;
  // whole row segments per tile: a piece of one image row, or up to 16 complete image rows; enough tiles for 256 CUs
  // (there is no split-K form of this kernel)
  auto fits = [&](int bm, int bn) {
    const bool seg = (a.Win % bm == 0) || (bm % a.Win == 0 && bm / a.Win <= 16);
    if (!seg || k.Mtot % bm != 0) return false;
    const int tiles = (int)((int64_t)(k.Mtot / bm) * ceil_div(a.Ntot, bn));
    if (a.splits > 1) {
      const int nit = 3 * (a.Cin / KC);
      return nit % (2 * a.splits) == 0 && nit / a.splits >= 2;
    }
    return rows_splits(tiles, a.Cin) == 1;
  };
  if (a.Ntot <= 32 && fits(192, 32)) return ROWS_192x32;
  if (a.Ntot <= 32 && fits(96, 32)) return ROWS_96x32;       // (e.g. 1440-pixel rows: 15 x 96)
  if (fits(96, 64)) return ROWS_96x64;
  return 0;
}

template <int WM, int WN, int TM, int TN, bool AFF, int MATH>
static int rows_launch(const ConvK& k, hipStream_t s) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const int Wt = k.a.Win >= BM ? BM : k.a.Win;
  const int E = BM + 2 * (BM / Wt);
  const int stage = 2 * (((E + 3) & ~3) * 8 + 16) + 3 * 2 * (BN * 8 + 16);
  size_t lds = (size_t)2 * stage * sizeof(float);
  if (AFF) lds += (size_t)2 * k.a.Cin * sizeof(float);
  ConvK q = k;
  q.splits = k.a.splits > 1 ? k.a.splits : 1;
  q.cps = 3 * (k.a.Cin / KC) / q.splits;      // (dy, channel chunk) pairs per slice
  dim3 grid((unsigned)(k.Mtot / BM), (unsigned)ceil_div(k.a.Ntot, BN), (unsigned)q.splits);
  hipLaunchKernelGGL((conv_rows_kernel<WM, WN, TM, TN, AFF, MATH>), grid, dim3(WM * WN * 64), lds, s, q);
  return check_launch("dfl_conv2d (row tiles)");
}

template <int WM, int WN, int TM, int TN>
static int rows_dispatch(const ConvK& k, hipStream_t s) {
  const bool aff = k.a.in_scale != nullptr || k.a.in_tot != nullptr;
  if (math_mode() == 3) {
    if (k.a.x_split) return rows_launch<WM, WN, TM, TN, false, 6>(k, s);
    return aff ? rows_launch<WM, WN, TM, TN, true, 5>(k, s) : rows_launch<WM, WN, TM, TN, false, 5>(k, s);
  }
  if (k.a.x_split) return rows_launch<WM, WN, TM, TN, false, 4>(k, s);
  return aff ? rows_launch<WM, WN, TM, TN, true, 3>(k, s) : rows_launch<WM, WN, TM, TN, false, 3>(k, s);
}

// slice count dfl_conv_suggest_splits should answer for this layer (0 = the layer is not for these kernels)
int conv_rows_splits(const ConvK& k) {
  ConvK q = k;
  q.a.splits = 1;
  q.a.add = nullptr;          // with slices every epilogue is the finish kernel's
  q.a.accumulate = 0;
  const dfl_conv_args& a = q.a;
  if (!rows_enabled() || !q.fast || a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1 || a.scatter2x2) return 0;
  for (int pass = 0; pass < 3; ++pass) {
    const int code = pass == 0 ? ROWS_192x32 : (pass == 1 ? ROWS_96x32 : ROWS_96x64);
    const int bm = conv_rows_bm(code), bn = pass == 2 ? 64 : 32;
    if (pass < 2 && a.Ntot > 32) continue;
    const bool seg = (a.Win % bm == 0) || (bm % a.Win == 0 && bm / a.Win <= 16);
    if (!seg || q.Mtot % bm != 0 || a.Cin % 32 != 0) continue;
    const int s = rows_splits((int)((int64_t)(q.Mtot / bm) * ceil_div(a.Ntot, bn)), a.Cin);
    if (s == 0) continue;
    q.a.splits = s;
    if (s > 1 || (k.a.add == nullptr && !k.a.accumulate)) return conv_rows_tile(s > 1 ? q : k) == code ? s : 0;
  }
  return 0;
}

int conv_rows_set_min_tiles(int n) {
  const int old = g_rows_min_tiles;
  g_rows_min_tiles = n;
  return old;
}

int conv_rows_launch(const ConvK& k, hipStream_t s) {
  const int t = conv_rows_tile(k);
  DFL_REQUIRE(t != 0 && k.Mtot % conv_rows_bm(t) == 0, "dfl_conv2d: internal: row-tiled kernel called for an ineligible layer");
  if (t == ROWS_192x32) return rows_dispatch<3, 1, 2, 1>(k, s);
  if (t == ROWS_96x32) return rows_dispatch<3, 1, 1, 1>(k, s);
  return rows_dispatch<3, 1, 1, 2>(k, s);
}

}  // namespace dfl
