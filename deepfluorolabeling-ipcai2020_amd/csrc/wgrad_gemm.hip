// Weight-gradient GEMM on the gfx950 fp32 matrix cores:  dw[cm][cg][t] = sum_m d[m][cm] * G(m, t, cg).
// Reference: torch autograd of nn.Conv2d / nn.ConvTranspose2d at train_test_code/unet.py:93,207,211,218,240
// (triggered by loss.backward(), train.py:422).  Contract: include/dfl_hip.h (dfl_conv2d_wgrad).
//
// The contraction index is the pixel m, the slow dimension of both NHWC operands, so both LDS images are simply
// [16 pixels][channels] as loaded (float4 = 4 channels of one pixel, ds_write_b128) and the MFMA operand reads are
// unit stride over channels.  MFMA rows = cm (dense tensor d), columns = cg (gathered tensor), hence the accumulator
// tile lands in torch's [Cout][Cin][KH][KW] order directly.
// TPB = taps handled by one workgroup: 1 (tap from blockIdx.y; wide layers), or a whole kernel ROW of taps (3 for 3x3,
// 2 for 2x2: narrow layers keep a 32x32 tile per wave with one accumulator per tap, so the d slab staged for a chunk is
// used TPB times while the wave still fits 4 per SIMD).  The pixel range is cut into `splits` slices (blockIdx.z);
// slices write partial[split], which dfl_sum_partials adds.
// Fast path (FAST): buffer loads with hardware bounds checking -- an out-of-image tap, a row past the end of the slice
// or a padded channel gets an out-of-range offset and reads 0 -- so the loop carries no clamping, selects or branches;
// the dense operand's offsets are loop invariant (the chunk advance rides in the scalar offset), the gathered operand
// keeps a pixel cursor advanced without divisions.  The general path covers odd channel counts (first layer, heads).
#include <stdlib.h>

#include "common.h"
#include "convp.h"
#include "direct_small.h"

namespace dfl {

constexpr int KP = 16;
constexpr uint32_t WOOB = 0x80000000u;

struct WgK {
  dfl_wgrad_args a;
  int Mtot, T, nchunks, cps;
  int vecG, vecD, fast;
  uint32_t g_bytes, d_bytes;
};

typedef unsigned int wu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 wbuf_load4(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
  const wu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

constexpr int WG_WS = 1;   // pixel-interleaved waves per workgroup of the one-wave tiles (2 and 4 measured: slices shrink, kernel slows, step time equal)

constexpr int wg_occ(int tiles, bool fast) { return tiles >= 4 ? 2 : (fast ? 4 : 3); }

// WS > 1 (one-wave tiles only): the workgroup is WS independent waves that take every WS-th chunk of the slice with
// their own LDS images and add their accumulators through LDS at the end -- same residency, 1/WS of the slice traffic.
// MATH 1 ("bf16x3", FAST only): operands are split into hi + lo bf16 when staged (common.h) and multiplied as
// hi*hi + hi*lo + lo*hi with v_mfma_f32_32x32x16_bf16.  That instruction wants 8 consecutive k (= pixels) per lane, while
// the staged image is pixel-major like the tensors; gfx950's transposing LDS read ds_read_b64_tr_b16 bridges the two:
// within 16 lanes, lane i supplies the address of 4 consecutive channels of pixel row i/4 (quad i%4) and lane c receives
// channel c of the 4 rows, i.e. 4 consecutive pixels of ITS channel -- two such reads make one operand.  LDS row of a
// pixel: [hi: channels][lo: channels][32 B pad] (the pad spreads the 4 rows of a read over distinct banks).
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef short s16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ bf16x8_t tr_read8(const float* row0, int row_words) {
  typedef __attribute__((address_space(3))) s16x4_t* lds_p;
  const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(row0));
  const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(row0 + 4 * row_words));
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

// HALO (kernel rows of a 3x3 / stride 1 / pad 1 window): the three taps of a kernel row read the SAME input pixels shifted by
// 0 / 1 / 2, so they are staged once -- every image-row segment of the chunk's 16 output pixels with one halo pixel on each
// side (18 rows when the chunk lies in one image row, up to 24 when it spans four rows of a 6-pixel-wide level) -- and
// tap t of pixel j reads staged row j + 2 seg(j) + t: a third of the gathered operand's loads and LDS writes.
// the same with the two 4-row halves at independent word offsets behind `base`
__device__ __forceinline__ bf16x8_t tr_read8_2(const float* base, int off0, int off1) {
  typedef __attribute__((address_space(3))) s16x4_t* lds_p;
  const s16x4_t a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + off0));
  const s16x4_t b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + off1));
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7));
}

template <int WM, int WN, int TM, int TN, int TPB, bool FAST, int WS, int MATH, bool HALO = false>
__global__ void __launch_bounds__(WM* WN * 64 * WS, wg_occ(TM* TN* TPB, FAST)) wgrad_kernel(const WgK p) {
  static_assert(!HALO || (TPB == 3 && FAST && MATH != 0 && WS == 1), "shared halo: kernel rows of three taps, split products");
  static_assert(WS == 1 || WM * WN == 1, "pixel-interleaved waves only for one-wave tiles");
  static_assert(MATH == 0 || ((MATH == 1 || MATH == 2) && FAST), "split-bf16 products exist for the fast path only");
  constexpr int NPW = MATH == 2 ? 1 : 2;   // MATH 2 = plain bf16 products (mode 3): hi parts only
  constexpr int NT = WM * WN * 64;   // threads of one tile (the "virtual workgroup" all staging indices refer to)
  constexpr int BMc = WM * TM * 32, BNg = WN * TN * 32;
  constexpr int LDD = BMc + (MATH ? 8 : 4), LDG = BNg + (MATH ? 8 : 4);   // words per staged pixel row
  constexpr int DQ = BMc / 4, GQ = BNg / 4;
  constexpr int NQD = KP * DQ, NQG = KP * GQ;
  constexpr int QD = (NQD + NT - 1) / NT, QG = (NQG + NT - 1) / NT;
  static_assert(NT % DQ == 0 && NT % GQ == 0, "thread -> channel quad mapping must be fixed per thread");
  // one-wave workgroups keep a single LDS image (9 KB -> 16 workgroups per CU); the next chunk waits in registers
  constexpr int NBUF = (WM * WN == 1) ? 1 : 2;

  constexpr int HROWS = KP + 8;                        // HALO: 16 pixels + 2 halo pixels for each of at most 4 segments
  constexpr int GROWS = HALO ? HROWS : TPB * KP;       // staged rows of the gathered operand per image
  constexpr int QGH = (HROWS * GQ + NT - 1) / NT;      // HALO: float4 per thread and chunk
  constexpr int LDS_PER = NBUF * (KP * LDD + GROWS * LDG);
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int sub = (WS > 1) ? (int)threadIdx.x / NT : 0;
  float* Ds = smem + sub * LDS_PER;    // [NBUF][KP][LDD]
  float* Gs = Ds + NBUF * KP * LDD;    // [NBUF][TPB][KP][LDG]  (HALO: [NBUF][HROWS][LDG])

  const dfl_wgrad_args& a = p.a;
  const int tid = (WS > 1) ? (int)threadIdx.x % NT : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int T = p.T;
  const int Hin = a.Hin, Win = a.Win, KW = a.KW;
  const int cm0 = blockIdx.x * BMc;
  // blockIdx.y = (cg tile, tap group); a tap group is one tap (TPB == 1) or one kernel row of TPB taps
  const int ngroups = T / TPB;
  const int cg0 = (blockIdx.y / ngroups) * BNg;
  const int tap0 = (blockIdx.y % ngroups) * TPB;
  const int tap_dy = tap0 / KW, tap_dx0 = tap0 - tap_dy * KW;   // TPB > 1: tap_dx0 == 0 and the group spans dx = 0..TPB-1
  const int ch_begin = blockIdx.z * p.cps;
  const int ch_end = min(ch_begin + p.cps, p.nchunks);

  const int gq = tid % GQ;  // this thread's channel quad of the gathered tensor (same for every pass)
  const int gc = cg0 + 4 * gq;
  const bool g1 = gc + 1 < a.Cg, g2 = gc + 2 < a.Cg, g3 = gc + 3 < a.Cg;
  const bool has_aff = a.in_scale != nullptr;
  float4 gsc = make_float4(1.f, 1.f, 1.f, 1.f), gsh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (has_aff) {
    if (gc + 0 < a.Cg) { gsc.x = a.in_scale[gc + 0]; gsh.x = a.in_shift[gc + 0]; }
    if (g1) { gsc.y = a.in_scale[gc + 1]; gsh.y = a.in_shift[gc + 1]; }
    if (g2) { gsc.z = a.in_scale[gc + 2]; gsh.z = a.in_shift[gc + 2]; }
    if (g3) { gsc.w = a.in_scale[gc + 3]; gsh.w = a.in_shift[gc + 3]; }
  }

  float4 rd[QD];
  float4 rg[TPB][QG];
  float4 rgh[QGH];       // HALO: the chunk's staged pixel rows
  int ox_loaded = 0;     // HALO: first output column of the chunk load() fetched last
  uint32_t okGh = 0;
  int h_ox = 0, h_oy = 0, h_n = 0;   // HALO: wave-uniform position of the chunk's first output pixel
  if constexpr (HALO) {
    const int m = ch_begin * KP;
    h_ox = m % a.Wout;
    const int tq = m / a.Wout;
    h_oy = tq % a.Hout;
    h_n = tq / a.Hout;
  }
  uint32_t okG = 0;      // bit (r*TPB + tt): tap tt of gather row r is inside the image (affine needs it: pad stays 0)
  bool okD[QD][4];       // general path only

  // pixel cursor of each gather row (advanced by KP per chunk, no divisions in the loop)
  int g_pix[QG], g_ox[QG], g_oy[QG], g_n[QG];
  bool g_ok0[QG];
#pragma unroll
  for (int r = 0; r < QG; ++r) {
    const int idx = tid + r * NT;
    g_pix[r] = idx / GQ;
    g_ok0[r] = (idx < NQG) && (gc < a.Cg);
    const int m = (ch_begin + sub) * KP + g_pix[r];
    g_ox[r] = m % a.Wout;
    const int tq = m / a.Wout;
    g_oy[r] = tq % a.Hout;
    g_n[r] = tq / a.Hout;
  }
  // FAST: loop-invariant offsets of the dense operand and per-tap byte offsets of the gathered one
  uint32_t d_voff[QD];
  __amdgpu_buffer_rsrc_t rsD, rsG;
  if constexpr (FAST) {
    rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.d), 0, (int)p.d_bytes, 0x00020000);
    rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g), 0, (int)p.g_bytes, 0x00020000);
#pragma unroll
    for (int r = 0; r < QD; ++r) {
      const int idx = tid + r * NT;
      const int pix = idx / DQ, q = idx - pix * DQ;
      const int c = cm0 + 4 * q;
      d_voff[r] = (idx < NQD && c < a.Cm) ? (uint32_t)(((int64_t)pix * a.ldd + c) * 4) : WOOB;
    }
  }

  auto load = [&](int ch) {
    const bool live = ch < ch_end;          // wave uniform: a wave may run one chunk past its share (reads zeros)
    const int mc0 = live ? ch * KP : 0;
    if constexpr (FAST) {
      const uint32_t soff = (uint32_t)mc0 * (uint32_t)a.ldd * 4u;
      const bool tail = !live || mc0 + KP > p.Mtot;   // the scalar offset takes no part in the bounds check
#pragma unroll
      for (int r = 0; r < QD; ++r) {
        const int pix = (tid + r * NT) / DQ;
        rd[r] = wbuf_load4(rsD, (tail && (!live || mc0 + pix >= p.Mtot)) ? WOOB : d_voff[r], soff);
      }
    } else {
#pragma unroll
      for (int r = 0; r < QD; ++r) {
        const int idx = tid + r * NT;
        const int pix = idx / DQ, q = idx - pix * DQ;
        const int m = mc0 + pix, c = cm0 + 4 * q;
        const bool ok = live && (idx < NQD) && (m < p.Mtot) && (c < a.Cm);
        const float* src = a.d + (ok ? ((int64_t)m * a.ldd + c) : 0);
        if (p.vecD) {
          okD[r][0] = okD[r][1] = okD[r][2] = okD[r][3] = ok;
          rd[r] = *reinterpret_cast<const float4*>(src);
        } else {
          okD[r][0] = ok;
          okD[r][1] = ok && (c + 1 < a.Cm);
          okD[r][2] = ok && (c + 2 < a.Cm);
          okD[r][3] = ok && (c + 3 < a.Cm);
          rd[r] = make_float4(src[0], src[okD[r][1] ? 1 : 0], src[okD[r][2] ? 2 : 0], src[okD[r][3] ? 3 : 0]);
        }
      }
    }
    if constexpr (HALO) {
      // The chunk's pixels j = 0..15 start at column h_ox of output row h_oy: segment 0 holds the L0 pixels up to the end
      // of that row, segments 1.. whole rows (the last one what is left).  Segment s is staged at rows
      // start_s + 2 s .. start_s + 2 s + len_s + 1 = its pixels framed by their two neighbours; staged row
      // e = idx / GQ (idx = tid + r * NT) therefore belongs to segment s(e) below and is input pixel
      // (row of segment s - 1 + tap_dy, first column of segment s - 1 + (e - first staged row of s)).
      const int Wd = a.Wout;
      const int L0 = min(KP, Wd - h_ox);
      const int rem = KP - L0;
      const int E = KP + 2 * (1 + (rem > 0) + (rem > Wd) + (rem > 2 * Wd));   // staged rows of this chunk
      ox_loaded = h_ox;
      okGh = 0;
#pragma unroll
      for (int r = 0; r < QGH; ++r) {
        const int e = (tid + r * NT) / GQ;
        const int sg = (e >= L0 + 2) + (e >= L0 + Wd + 4) + (e >= L0 + 2 * Wd + 6);
        const int start = (sg == 0) ? 0 : L0 + (sg - 1) * Wd;            // first chunk pixel of the segment
        const int x = ((sg == 0) ? h_ox : 0) - 1 + (e - start - 2 * sg);
        int oy = h_oy + sg, n = h_n;
        if (oy >= a.Hout) {
          oy -= a.Hout;
          ++n;
        }
        const int iy = oy - 1 + tap_dy;
        const bool in = live && e < E && gc < a.Cg && (unsigned)x < (unsigned)Win && (unsigned)iy < (unsigned)Hin && n < a.N;
        okGh |= in ? (1u << r) : 0u;
        rgh[r] = wbuf_load4(rsG, in ? (uint32_t)(((((int64_t)n * Hin + iy) * Win + x) * a.ldg + gc) * 4) : WOOB, 0);
      }
      h_ox += KP;
      while (h_ox >= Wd) {
        h_ox -= Wd;
        if (++h_oy == a.Hout) {
          h_oy = 0;
          ++h_n;
        }
      }
      return;
    }
    okG = 0;
#pragma unroll
    for (int r = 0; r < QG; ++r) {
      const int m = mc0 + g_pix[r];
      const bool ok = live && g_ok0[r] && (m < p.Mtot);
      const int iy = g_oy[r] * a.stride - a.pad + tap_dy;
      const int ix0 = g_ox[r] * a.stride - a.pad + tap_dx0;
      const bool rowok = ok && (unsigned)iy < (unsigned)Hin;
      const int64_t pix0 = ((int64_t)g_n[r] * Hin + iy) * Win + ix0;
#pragma unroll
      for (int tt = 0; tt < TPB; ++tt) {
        const bool in = rowok && (unsigned)(ix0 + tt) < (unsigned)Win;
        okG |= in ? (1u << (r * TPB + tt)) : 0u;
        if constexpr (FAST) {
          rg[tt][r] = wbuf_load4(rsG, in ? (uint32_t)(((pix0 + tt) * a.ldg + gc) * 4) : WOOB, 0);
        } else {
          const float* src = a.g + (in ? ((pix0 + tt) * a.ldg + gc) : 0);
          if (p.vecG) {
            rg[tt][r] = *reinterpret_cast<const float4*>(src);
          } else {
            rg[tt][r] = make_float4(src[0], src[(in && g1) ? 1 : 0], src[(in && g2) ? 2 : 0], src[(in && g3) ? 3 : 0]);
          }
        }
      }
      g_ox[r] += KP * WS;
      while (g_ox[r] >= a.Wout) {
        g_ox[r] -= a.Wout;
        if (++g_oy[r] == a.Hout) {
          g_oy[r] = 0;
          ++g_n[r];
        }
      }
    }
  };

  auto store = [&](int buf) {
    float* Db = Ds + buf * KP * LDD;
#pragma unroll
    for (int r = 0; r < QD; ++r) {
      const int idx = tid + r * NT;
      if (idx < NQD) {
        const int pix = idx / DQ, q = idx - pix * DQ;
        float4 v = rd[r];
        if constexpr (!FAST) {
          v.x = okD[r][0] ? v.x : 0.f;
          v.y = okD[r][1] ? v.y : 0.f;
          v.z = okD[r][2] ? v.z : 0.f;
          v.w = okD[r][3] ? v.w : 0.f;
        }
        if constexpr (MATH != 0) {
          uint2 parts[2];
          if (a.d_split) {   // the producer already left hi4 | lo4 in the slot (wave uniform)
            parts[0] = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y));
            parts[1] = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w));
          } else {
            split_bf16<NPW>(v, parts);
          }
          *reinterpret_cast<uint2*>(Db + pix * LDD + 2 * q) = parts[0];
          if constexpr (NPW > 1) *reinterpret_cast<uint2*>(Db + pix * LDD + BMc / 2 + 2 * q) = parts[1];
        } else {
          *reinterpret_cast<float4*>(Db + pix * LDD + 4 * q) = v;
        }
      }
    }
    float* Gb = Gs + buf * GROWS * LDG;
    if constexpr (HALO) {
#pragma unroll
      for (int r = 0; r < QGH; ++r) {
        const int e = (tid + r * NT) / GQ;
        if (e < HROWS) {
          float4 v = rgh[r];
          if (has_aff) {   // the hardware returned zeros outside the image: the affine must leave them zero (pad after BN)
            const bool in = (okGh >> r) & 1u;
            v.x = in ? fmaf(v.x, gsc.x, gsh.x) : 0.f;
            v.y = in ? fmaf(v.y, gsc.y, gsh.y) : 0.f;
            v.z = in ? fmaf(v.z, gsc.z, gsh.z) : 0.f;
            v.w = in ? fmaf(v.w, gsc.w, gsh.w) : 0.f;
          }
          uint2 parts[2];
          split_bf16<NPW>(v, parts);
          *reinterpret_cast<uint2*>(Gb + e * LDG + 2 * gq) = parts[0];
          if constexpr (NPW > 1) *reinterpret_cast<uint2*>(Gb + e * LDG + BNg / 2 + 2 * gq) = parts[1];
        }
      }
      return;
    }
#pragma unroll
    for (int r = 0; r < QG; ++r) {
      const int idx = tid + r * NT;
      if (idx < NQG) {
        const int pix = idx / GQ;
#pragma unroll
        for (int tt = 0; tt < TPB; ++tt) {
          float4 v = rg[tt][r];
          const bool in = (okG >> (r * TPB + tt)) & 1u;
          if (!FAST || has_aff) {   // FAST without affine: the hardware already returned zeros where needed
            v.x = in ? fmaf(v.x, gsc.x, gsh.x) : 0.f;
            v.y = (in && (FAST || g1)) ? fmaf(v.y, gsc.y, gsh.y) : 0.f;
            v.z = (in && (FAST || g2)) ? fmaf(v.z, gsc.z, gsh.z) : 0.f;
            v.w = (in && (FAST || g3)) ? fmaf(v.w, gsc.w, gsh.w) : 0.f;
          }
          if constexpr (MATH != 0) {
            uint2 parts[2];
            split_bf16<NPW>(v, parts);
            *reinterpret_cast<uint2*>(Gb + (tt * KP + pix) * LDG + 2 * gq) = parts[0];
            if constexpr (NPW > 1) *reinterpret_cast<uint2*>(Gb + (tt * KP + pix) * LDG + BNg / 2 + 2 * gq) = parts[1];
          } else {
            *reinterpret_cast<float4*>(Gb + (tt * KP + pix) * LDG + 4 * gq) = v;
          }
        }
      }
    }
  };

  f32x16 acc[TPB][TM][TN];
#pragma unroll
  for (int tt = 0; tt < TPB; ++tt)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tt][i][j][r] = 0.f;

  const int nit = (ch_end - ch_begin + WS - 1) / WS;   // workgroup uniform (barriers inside)
  load(ch_begin + sub);
  store(0);
  __syncthreads();
  for (int it = 0; it < nit; ++it) {
    const int ch = ch_begin + sub + it * WS;
    const int buf = (NBUF == 2) ? (it & 1) : 0;
    const bool more = (it + 1) < nit;
    const int ox_comp = ox_loaded;   // HALO: first column of the chunk multiplied in this trip
    if (more) load(ch + WS);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MATH != 0) {
      // this lane's transposing-read address inside a 32-channel tile: pixel row 8*lh + (lane&15)/4 (second read: +4),
      // channels 16*((lane>>4)&1) + 4*(lane&3) .. +3  (bf16: 2 per word)
      const int trow = 8 * lh + ((lane & 15) >> 2), tcw = 8 * ((lane >> 4) & 1) + 2 * (lane & 3);
      const float* Dt = Ds + buf * KP * LDD + trow * LDD + (wm * (TM * 32)) / 2 + tcw;
      const float* Gt = Gs + buf * GROWS * LDG + trow * LDG + (wn * (TN * 32)) / 2 + tcw;
      int gsh0 = 0, gsh1 = 4 * LDG;   // HALO: word offsets of this lane's two pixel rows (trow, trow + 4) behind Gt, tap 0
      if constexpr (HALO) {
        const int Wd = a.Wout, L0 = min(KP, Wd - ox_comp);
        const int j0 = trow, j1 = trow + 4;
        gsh0 = 2 * ((j0 >= L0) + (j0 >= L0 + Wd) + (j0 >= L0 + 2 * Wd)) * LDG;
        gsh1 = (4 + 2 * ((j1 >= L0) + (j1 >= L0 + Wd) + (j1 >= L0 + 2 * Wd))) * LDG;
      }
      bf16x8_t dp[TM][2];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int q = 0; q < NPW; ++q) dp[i][q] = tr_read8(Dt + i * 16 + q * (BMc / 2), LDD);
#pragma unroll
      for (int tt = 0; tt < TPB; ++tt) {
        bf16x8_t gp[TN][2];
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < NPW; ++q)
            gp[j][q] = HALO ? tr_read8_2(Gt + tt * LDG + j * 16 + q * (BNg / 2), gsh0, gsh1)
                            : tr_read8(Gt + tt * KP * LDG + j * 16 + q * (BNg / 2), LDG);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) {   // small terms first
            if constexpr (NPW > 1) {
              acc[tt][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dp[i][1], gp[j][0], acc[tt][i][j], 0, 0, 0);
              acc[tt][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dp[i][0], gp[j][1], acc[tt][i][j], 0, 0, 0);
            }
            acc[tt][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dp[i][0], gp[j][0], acc[tt][i][j], 0, 0, 0);
          }
      }
    }
    const float* Db = Ds + buf * KP * LDD + wm * (TM * 32) + li;
    const float* Gb = Gs + buf * GROWS * LDG + wn * (TN * 32) + li;
#pragma unroll
    for (int kk = 0; kk < (MATH == 0 ? KP / 2 : 0); ++kk) {
      float dv[TM];
#pragma unroll
      for (int i = 0; i < TM; ++i) dv[i] = Db[(2 * kk + lh) * LDD + i * 32];
#pragma unroll
      for (int tt = 0; tt < TPB; ++tt) {
        float gv[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) gv[j] = Gb[(tt * KP + 2 * kk + lh) * LDG + j * 32];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[tt][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(dv[i], gv[j], acc[tt][i][j], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the affine + LDS writes (and their vmcnt waits) behind the MFMA block
    if constexpr (NBUF == 1) __syncthreads();
    if (more) store(NBUF == 2 ? (buf ^ 1) : 0);
    __syncthreads();
  }

  if constexpr (WS > 1) {
    // the other waves hand their accumulators over through LDS ([tile][register][lane], the staging images are dead)
    if (sub > 0) {
      float* dst = smem + (sub - 1) * (TPB * TM * TN * 16 * 64);
#pragma unroll
      for (int tt = 0; tt < TPB; ++tt)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[(((tt * TM + i) * TN + j) * 16 + r) * 64 + lane] = acc[tt][i][j][r];
    }
    __syncthreads();
    if (sub > 0) return;
    for (int s2 = 0; s2 < WS - 1; ++s2) {   // fixed order
      const float* src = smem + s2 * (TPB * TM * TN * 16 * 64);
#pragma unroll
      for (int tt = 0; tt < TPB; ++tt)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[tt][i][j][r] += src[(((tt * TM + i) * TN + j) * 16 + r) * 64 + lane];
    }
  }
  // slices: tap-major [T][Cm][Cg] (each accumulator row is a contiguous 128-byte store); final result: torch order
  const bool sliced = a.splits > 1;
  float* out = sliced ? a.partial + (int64_t)blockIdx.z * a.Cm * a.Cg * T : a.dw;
#pragma unroll
  for (int tt = 0; tt < TPB; ++tt) {
    const int t = tap0 + tt;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int cg = cg0 + wn * (TN * 32) + j * 32 + li;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cm = cm0 + wm * (TM * 32) + i * 32 + mfma32_row(r, lane);
          const int64_t o = sliced ? ((int64_t)t * a.Cm + cm) * a.Cg + cg : ((int64_t)cm * a.Cg + cg) * T + t;
          if (cm < a.Cm && cg < a.Cg) out[o] = acc[tt][i][j][r];
        }
      }
    }
  }
}

// i walks the slices' order ([T][R]); the sum lands at [R][T]
__global__ void sum_partials_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n, int splits,
                                    int T) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t R = n / T;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += src[(int64_t)k * n + i];
    dst[T > 1 ? (i % R) * T + i / R : i] = s;
  }
}

// Many slices of a small tensor (narrow layers cut the pixel range into ~1000 slices): 16 outputs x 16 slice lanes per
// workgroup, 8 independent loads in flight per thread, LDS tree over the lanes.  Same summation order every run.
__global__ void __launch_bounds__(256) sum_partials_wide_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                int64_t n, int splits, int T) {
  __shared__ float red[16][17];
  const int o = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int64_t i = (int64_t)blockIdx.x * 16 + o;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (i < n) {
    int k = sl;
    for (; k + 7 * 16 < splits; k += 8 * 16) {
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[u] += src[(int64_t)(k + u * 16) * n + i];
    }
    for (; k < splits; k += 16) acc[0] += src[(int64_t)k * n + i];
  }
  red[sl][o] = ((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7]));
  __syncthreads();
  if (sl == 0 && i < n) {
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += red[j][o];
    const int64_t R = n / T;
    dst[T > 1 ? (i % R) * T + i / R : i] = s;
  }
}

// ---- host side -------------------------------------------------------------------------------------
enum WgCfg { WG_128 = 0, WG_64, WG_ROW3, WG_ROW2, WG_32, WG_64ROW = 6 };   // (5 = CFG_DIRECT in dfl_wgrad_config)

// the shared-halo form of the kernel-row variant: chunks of 16 output pixels never leave their image row
static bool wg_halo_ok(const dfl_wgrad_args* a) {
  constexpr bool on = true;       // (against three stagings: -11 ... 14 % per launch, round 1)
  // (chunks of 16 output pixels may span up to four image rows)
  return on && a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad == 1 && a->Wout >= 6 &&
         ((int64_t)a->N * a->Hout * a->Wout) % KP == 0 && WG_WS == 1;
}

static WgCfg pick_wg(const dfl_wgrad_args* a) {
  const bool narrow = (a->Cm <= 64 || a->Cg <= 64);
  if (narrow) {
    if (a->KH == 3 && a->KW == 3) return WG_ROW3;
    if (a->KH == 2 && a->KW == 2) return WG_ROW2;
    return WG_32;
  }
  const int T = a->KH * a->KW;
  if (a->Cm >= 256 && a->Cg >= 256 && (int64_t)a->Cm * a->Cg * T >= 128ll * 128 * 1024) return WG_128;
  // 64 x 64 tiles with a kernel row of taps per workgroup and the shared halo (three accumulators per wave): a third of
  // the loads per matrix instruction; the workgroup count is kept by three times the pixel slices (the extra partial-sum
  // traffic costs less than the loads saved: 0.080 -> 0.051 + 0.008 ms at 48 x 48 x 128 x 128).
  constexpr int row64 = 1;
  // Only where chunks stay inside an image row (W % 16 == 0), i.e. the wide levels: below that the layers are the deep
  // ones, whose slices are megabytes each -- measured on the 192 x 192 network the extra partial sums eat the gain there
  // (1876-1897 vs 1890-1918 images/s), while the 768 x 768 configuration gains 4 %.
  if (row64 && math_mode() == 1 && wg_halo_ok(a) && a->Wout % KP == 0) return WG_64ROW;
  return WG_64;
}

static void wg_tile(WgCfg c, int* bm, int* bn, int* tpb) {
  switch (c) {
    case WG_128: *bm = 128; *bn = 128; *tpb = 1; break;
    case WG_64: *bm = 64; *bn = 64; *tpb = 1; break;
    case WG_64ROW: *bm = 64; *bn = 64; *tpb = 3; break;
    case WG_ROW3: *bm = 32; *bn = 32; *tpb = 3; break;
    case WG_ROW2: *bm = 32; *bn = 32; *tpb = 2; break;
    default: *bm = 32; *bn = 32; *tpb = 1; break;
  }
}

static int wg_prepare(const dfl_wgrad_args* a, WgK* k, bool need_out) {
  DFL_REQUIRE(a != nullptr, "dfl_conv2d_wgrad: null args");
  DFL_REQUIRE(a->N > 0 && a->Hin > 0 && a->Win > 0 && a->Cg > 0 && a->Cm > 0, "dfl_conv2d_wgrad: bad sizes");
  DFL_REQUIRE(a->KH > 0 && a->KW > 0 && a->stride > 0 && a->pad >= 0, "dfl_conv2d_wgrad: bad window");
  const int ho = (a->Hin + 2 * a->pad - a->KH) / a->stride + 1;
  const int wo = (a->Win + 2 * a->pad - a->KW) / a->stride + 1;
  DFL_REQUIRE(ho == a->Hout && wo == a->Wout, "dfl_conv2d_wgrad: Hout/Wout (%d,%d) do not match the window (%d,%d)",
              a->Hout, a->Wout, ho, wo);
  DFL_REQUIRE(a->ldg >= a->Cg && a->ldd >= a->Cm, "dfl_conv2d_wgrad: ld < C");
  DFL_REQUIRE((a->in_scale == nullptr) == (a->in_shift == nullptr), "dfl_conv2d_wgrad: in_scale/in_shift go together");
  if (need_out) {
    DFL_REQUIRE(a->g && a->d, "dfl_conv2d_wgrad: g and d are required");
    DFL_REQUIRE(a->splits >= 1, "dfl_conv2d_wgrad: splits >= 1");
    DFL_REQUIRE(a->splits > 1 ? a->partial != nullptr : a->dw != nullptr, "dfl_conv2d_wgrad: output buffer missing");
  }
  const int64_t M = (int64_t)a->N * a->Hout * a->Wout;
  DFL_REQUIRE(M < (1ll << 31) && (int64_t)a->N * a->Hin * a->Win < (1ll << 31), "dfl_conv2d_wgrad: too many pixels");
  k->a = *a;
  k->Mtot = (int)M;
  k->T = a->KH * a->KW;
  k->nchunks = (int)ceil_div(M, KP);
  const int r4g = (a->Cg + 3) / 4 * 4, r4d = (a->Cm + 3) / 4 * 4;
  k->vecG = (a->ldg % 4 == 0) && (a->ldg >= r4g) && aligned16(a->g);
  k->vecD = (a->ldd % 4 == 0) && (a->ldd >= r4d) && aligned16(a->d);
  // Channel counts that are not multiples of 4 (the heads: 7, 39, 21, 14) still take the fast path when the rows are
  // padded (ld >= C rounded up): the pad channels are read as rows / columns of the tile that are never stored.
  const int64_t gb = (((int64_t)a->N * a->Hin * a->Win - 1) * a->ldg + r4g) * 4;
  const int64_t db = ((M - 1) * a->ldd + r4d) * 4;
  const int64_t lim = (1ll << 31) - 4096;
  k->fast = k->vecG && k->vecD && gb < lim && db < lim;
  k->g_bytes = (uint32_t)(gb < lim ? gb : 0);
  k->d_bytes = (uint32_t)(db < lim ? db : 0);
  return DFL_OK;
}

template <int WM, int WN, int TM, int TN, int TPB, bool FAST, int MATH = 0, bool HALO = false>
static int wg_launch(const WgK& k, hipStream_t s) {
  constexpr int BMc = WM * TM * 32, BNg = WN * TN * 32;
  constexpr int NBUF = (WM * WN == 1) ? 1 : 2;
  constexpr int WS = (WM * WN == 1) ? WG_WS : 1;
  constexpr int PADW = MATH ? 8 : 4;
  constexpr int GROWS = HALO ? KP + 8 : TPB * KP;
  size_t lds = (size_t)WS * NBUF * (KP * (BMc + PADW) + GROWS * (BNg + PADW)) * sizeof(float);
  const size_t handover = (size_t)(WS - 1) * TPB * TM * TN * 16 * 64 * sizeof(float);
  if (handover > lds) lds = handover;
  const int tiles_g = (int)ceil_div(k.a.Cg, BNg);
  dim3 grid((unsigned)ceil_div(k.a.Cm, BMc), (unsigned)(tiles_g * (k.T / TPB)), (unsigned)k.a.splits);
  hipLaunchKernelGGL((wgrad_kernel<WM, WN, TM, TN, TPB, FAST, WS, MATH, HALO>), grid, dim3(WM * WN * 64 * WS), lds, s, k);
  return check_launch("dfl_conv2d_wgrad");
}

}  // namespace dfl

extern "C" int dfl_wgrad_config(const dfl_wgrad_args* a) {
  if (a != nullptr && a->g_bf16 && a->d_bf16) return dfl::wgradp_config(a);
  dfl::WgK k;
  int rc = dfl::wg_prepare(a, &k, false);
  if (rc != DFL_OK) return rc;
  if (dfl::direct_wgrad_ok(a)) return dfl::CFG_DIRECT;
  return (int)dfl::pick_wg(a);
}

extern "C" int dfl_wgrad_suggest_splits(const dfl_wgrad_args* a) {
  if (a != nullptr && a->g_bf16 && a->d_bf16) return dfl::wgradp_suggest_splits(a);
  dfl::WgK k;
  int rc = dfl::wg_prepare(a, &k, false);
  if (rc != DFL_OK) return rc;
  if (dfl::direct_wgrad_ok(a)) return dfl::direct_wgrad_splits(a);
  int bm, bn, tpb;
  dfl::wg_tile(dfl::pick_wg(a), &bm, &bn, &tpb);
  const int waves = (bm >= 64) ? 4 : dfl::WG_WS;   // 2x2 waves of the wide tiles, WG_WS interleaved waves of the one-wave tiles
  const int64_t blocks = dfl::ceil_div(a->Cm, bm) * dfl::ceil_div(a->Cg, bn) * (k.T / tpb);
  // aim at ~4 waves per SIMD over the whole chip (4096 waves), every slice at least 128 pixels
  int64_t s = dfl::ceil_div(4096, blocks * waves);   // (3072 / 2048 / 1536 measured with bf16x3 products: slower)
  const int64_t max_by_work = k.nchunks / 8 > 0 ? k.nchunks / 8 : 1;
  if (s > max_by_work) s = max_by_work;
  if (s > 2048) s = 2048;
  if (s < 1) s = 1;
  return (int)s;
}

extern "C" int dfl_conv2d_wgrad(const dfl_wgrad_args* a, dfl_stream_t stream) {
  if (a != nullptr && a->g_bf16 && a->d_bf16) return dfl::wgradp_launch(a, static_cast<hipStream_t>(stream));
  dfl::WgK k;
  int rc = dfl::wg_prepare(a, &k, true);
  if (rc != DFL_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dfl::direct_wgrad_ok(a)) {
    DFL_REQUIRE(!a->d_split, "dfl_conv2d_wgrad: a split d is not defined for the direct small-K kernels");
    return dfl::direct_wgrad_launch(a, s);
  }
  DFL_REQUIRE(!a->g_bf16 && !a->d_bf16, "dfl_conv2d_wgrad: mixed bf16 / fp32 operands only for the direct small-K layers");
  DFL_REQUIRE(a->d_mode == 0, "dfl_conv2d_wgrad: d_mode (fused BatchNorm + ReLU backward operand) is implemented by the bf16 patch kernels and the 1-channel 3x3 row form only");
  k.cps = (int)dfl::ceil_div(k.nchunks, a->splits);
  const bool f = k.fast;
  DFL_REQUIRE(!a->d_split || (f && (dfl::math_mode() == 1 || dfl::math_mode() == 3) && !dfl::direct_wgrad_ok(a)),
              "dfl_conv2d_wgrad: a split d needs math mode 1 or 3 (bf16x3 / bf16) and the fast path");
  if (f && dfl::math_mode() == 3) {
    switch (dfl::pick_wg(a)) {
      case dfl::WG_128: return dfl::wg_launch<2, 2, 2, 2, 1, true, 2>(k, s);
      case dfl::WG_64: return dfl::wg_launch<2, 2, 1, 1, 1, true, 2>(k, s);
      case dfl::WG_ROW3: return dfl::wg_launch<1, 1, 1, 1, 3, true, 2>(k, s);
      case dfl::WG_ROW2: return dfl::wg_launch<1, 1, 1, 1, 2, true, 2>(k, s);
      default: return dfl::wg_launch<1, 1, 1, 1, 1, true, 2>(k, s);
    }
  }
  if (f && dfl::math_mode() == 1) {
    switch (dfl::pick_wg(a)) {
      case dfl::WG_128: return dfl::wg_launch<2, 2, 2, 2, 1, true, 1>(k, s);
      case dfl::WG_64: return dfl::wg_launch<2, 2, 1, 1, 1, true, 1>(k, s);
      case dfl::WG_64ROW: return dfl::wg_launch<2, 2, 1, 1, 3, true, 1, true>(k, s);
      case dfl::WG_ROW3:
        if (dfl::wg_halo_ok(a)) return dfl::wg_launch<1, 1, 1, 1, 3, true, 1, true>(k, s);
        return dfl::wg_launch<1, 1, 1, 1, 3, true, 1>(k, s);
      case dfl::WG_ROW2: return dfl::wg_launch<1, 1, 1, 1, 2, true, 1>(k, s);
      default: return dfl::wg_launch<1, 1, 1, 1, 1, true, 1>(k, s);
    }
  }
  switch (dfl::pick_wg(a)) {
    case dfl::WG_128: return f ? dfl::wg_launch<2, 2, 2, 2, 1, true>(k, s) : dfl::wg_launch<2, 2, 2, 2, 1, false>(k, s);
    case dfl::WG_64: return f ? dfl::wg_launch<2, 2, 1, 1, 1, true>(k, s) : dfl::wg_launch<2, 2, 1, 1, 1, false>(k, s);
    case dfl::WG_ROW3: return f ? dfl::wg_launch<1, 1, 1, 1, 3, true>(k, s) : dfl::wg_launch<1, 1, 1, 1, 3, false>(k, s);
    case dfl::WG_ROW2: return f ? dfl::wg_launch<1, 1, 1, 1, 2, true>(k, s) : dfl::wg_launch<1, 1, 1, 1, 2, false>(k, s);
    default: return f ? dfl::wg_launch<1, 1, 1, 1, 1, true>(k, s) : dfl::wg_launch<1, 1, 1, 1, 1, false>(k, s);
  }
}

extern "C" int dfl_sum_partials(const float* src, float* dst, int64_t n, int32_t splits, int32_t T, dfl_stream_t stream) {
  DFL_REQUIRE(src && dst && n >= 0 && splits >= 1, "dfl_sum_partials: bad args");
  if (T < 1) T = 1;
  DFL_REQUIRE(n % T == 0, "dfl_sum_partials: n must be a multiple of T");
  if (n == 0) return DFL_OK;
  if (splits >= 32 && n <= (1 << 20)) {
    hipLaunchKernelGGL(dfl::sum_partials_wide_kernel, dim3((unsigned)dfl::ceil_div(n, 16)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), src, dst, n, (int)splits, (int)T);
    return dfl::check_launch("dfl_sum_partials");
  }
  int64_t blocks = dfl::ceil_div(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dfl::sum_partials_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     src, dst, n, (int)splits, (int)T);
  return dfl::check_launch("dfl_sum_partials");
}
