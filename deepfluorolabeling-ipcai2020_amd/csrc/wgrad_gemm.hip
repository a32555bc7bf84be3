// Weight-gradient GEMM on the gfx950 fp32 matrix cores:  dw[cm][cg][t] = sum_m d[m][cm] * G(m, t, cg).
// Reference: torch autograd of nn.Conv2d / nn.ConvTranspose2d at train_test_code/unet.py:93,207,211,218,240
// (triggered by loss.backward(), train.py:422).  Contract: include/dfl_hip.h (dfl_conv2d_wgrad).
//
// The contraction index is the pixel m, which is the slow dimension of both NHWC operands, so both LDS images
// are simply [16 pixels][channels] as loaded (float4 = 4 channels of one pixel, ds_write_b128) and the MFMA
// operand reads are unit stride over channels.  MFMA rows = cm (dense tensor d), columns = cg (gathered
// tensor), hence the accumulator tile lands in torch's [Cout][Cin][KH][KW] order directly.
// TPB = taps handled by one workgroup: 1 (tap comes from blockIdx.y, deep/wide layers) or all KH*KW taps
// (narrow layers: the d slab is staged once and used by every tap; 9 accumulator tiles per wave).
// The pixel range is cut into `splits` slices (blockIdx.z); slices write partial[split] which dfl_sum_partials adds.
#include "common.h"

namespace dfl {

constexpr int KP = 16;

struct WgK {
  dfl_wgrad_args a;
  int Mtot, T, nchunks, cps;
  int vecG, vecD;
};

template <int WM, int WN, int TM, int TN, int TPB, bool VEC>
__global__ void __launch_bounds__(WM* WN * 64) wgrad_kernel(const WgK p) {
  constexpr int NT = WM * WN * 64;
  constexpr int BMc = WM * TM * 32, BNg = WN * TN * 32;
  constexpr int LDD = BMc + 4, LDG = BNg + 4;
  constexpr int DQ = BMc / 4, GQ = BNg / 4;
  constexpr int NQD = KP * DQ, NQG = KP * GQ;
  constexpr int QD = (NQD + NT - 1) / NT, QG = (NQG + NT - 1) / NT;
  static_assert(NT % DQ == 0 && NT % GQ == 0, "thread -> channel quad mapping must be fixed per thread");

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* Ds = smem;                 // [2][KP][LDD]
  float* Gs = smem + 2 * KP * LDD;  // [2][TPB][KP][LDG]

  const dfl_wgrad_args& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;
  const int T = p.T;
  const int cm0 = blockIdx.x * BMc;
  int cg0, tap0;
  if (TPB == 1) {
    cg0 = (blockIdx.y / T) * BNg;
    tap0 = blockIdx.y % T;
  } else {
    cg0 = blockIdx.y * BNg;
    tap0 = 0;
  }
  const int ch_begin = blockIdx.z * p.cps;
  const int ch_end = min(ch_begin + p.cps, p.nchunks);
  const int Hin = a.Hin, Win = a.Win, KW = a.KW;

  const int gq = tid % GQ;  // this thread's channel quad of the gathered tensor (same for every pass)
  const int gc = cg0 + 4 * gq;
  float4 gsc = make_float4(1.f, 1.f, 1.f, 1.f), gsh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.in_scale != nullptr) {
    if (VEC) {
      if (gc < a.Cg) {
        gsc = *reinterpret_cast<const float4*>(a.in_scale + gc);
        gsh = *reinterpret_cast<const float4*>(a.in_shift + gc);
      }
    } else {
      if (gc + 0 < a.Cg) { gsc.x = a.in_scale[gc + 0]; gsh.x = a.in_shift[gc + 0]; }
      if (gc + 1 < a.Cg) { gsc.y = a.in_scale[gc + 1]; gsh.y = a.in_shift[gc + 1]; }
      if (gc + 2 < a.Cg) { gsc.z = a.in_scale[gc + 2]; gsh.z = a.in_shift[gc + 2]; }
      if (gc + 3 < a.Cg) { gsc.w = a.in_scale[gc + 3]; gsh.w = a.in_shift[gc + 3]; }
    }
  }

  float4 rd[QD];
  float4 rg[TPB][QG];

  // Unconditional loads from clamped addresses; selects / affine / LDS writes are deferred to store() so that the
  // waits land behind the MFMA block (see the note in conv_gemm.hip).
  bool okD[QD][4];
  bool okG[TPB][QG];
  const bool g1 = gc + 1 < a.Cg, g2 = gc + 2 < a.Cg, g3 = gc + 3 < a.Cg;
  constexpr int TKH = (TPB == 9) ? 3 : (TPB == 4 ? 2 : 1);   // multi-tap variants are square windows
  const int tap_dy = tap0 / KW, tap_dx = tap0 - (tap0 / KW) * KW;
  int g_pix[QG], g_ox[QG], g_oy[QG], g_n[QG];
  bool g_ok0[QG];
#pragma unroll
  for (int r = 0; r < QG; ++r) {
    const int idx = tid + r * NT;
    g_pix[r] = idx / GQ;
    g_ok0[r] = (idx < NQG) && (gc < a.Cg);
    const int m = ch_begin * KP + g_pix[r];
    g_ox[r] = m % a.Wout;
    const int tq = m / a.Wout;
    g_oy[r] = tq % a.Hout;
    g_n[r] = tq / a.Hout;
  }

  auto load = [&](int ch) {
    const int mc0 = ch * KP;
#pragma unroll
    for (int r = 0; r < QD; ++r) {
      const int idx = tid + r * NT;
      const int pix = idx / DQ, q = idx - pix * DQ;
      const int m = mc0 + pix, c = cm0 + 4 * q;
      const bool ok = (idx < NQD) && (m < p.Mtot) && (c < a.Cm);
      const float* src = a.d + (ok ? ((int64_t)m * a.ldd + c) : 0);
      if constexpr (VEC) {
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
#pragma unroll
    for (int r = 0; r < QG; ++r) {
      // pixel cursor (g_ox, g_oy, g_n) of this thread's row: advanced by KP per chunk, no divisions in the loop
      const int m = mc0 + g_pix[r];
      const bool ok = g_ok0[r] && (m < p.Mtot);
      const int iy0 = g_oy[r] * a.stride - a.pad;
      const int ix0 = g_ox[r] * a.stride - a.pad;
      const int64_t base = (((int64_t)g_n[r] * Hin + iy0) * Win + ix0) * a.ldg + gc;
      bool rowok[TKH], colok[TKH];
#pragma unroll
      for (int d = 0; d < TKH; ++d) {
        const int dy = (TPB == 1) ? tap_dy : d, dx = (TPB == 1) ? tap_dx : d;
        rowok[d] = (unsigned)(iy0 + dy) < (unsigned)Hin;
        colok[d] = (unsigned)(ix0 + dx) < (unsigned)Win;
      }
#pragma unroll
      for (int tt = 0; tt < TPB; ++tt) {
        const int dyi = (TPB == 1) ? 0 : tt / TKH, dxi = (TPB == 1) ? 0 : tt % TKH;
        const int dy = (TPB == 1) ? tap_dy : dyi, dx = (TPB == 1) ? tap_dx : dxi;
        const bool in = ok && rowok[dyi] && colok[dxi];
        const float* src = a.g + (in ? (base + ((int64_t)dy * Win + dx) * a.ldg) : 0);
        okG[tt][r] = in;
        if constexpr (VEC) {
          rg[tt][r] = *reinterpret_cast<const float4*>(src);
        } else {
          rg[tt][r] = make_float4(src[0], src[(in && g1) ? 1 : 0], src[(in && g2) ? 2 : 0], src[(in && g3) ? 3 : 0]);
        }
      }
      g_ox[r] += KP;
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
        float4 v;
        v.x = okD[r][0] ? rd[r].x : 0.f;
        v.y = okD[r][1] ? rd[r].y : 0.f;
        v.z = okD[r][2] ? rd[r].z : 0.f;
        v.w = okD[r][3] ? rd[r].w : 0.f;
        *reinterpret_cast<float4*>(Db + pix * LDD + 4 * q) = v;
      }
    }
    float* Gb = Gs + buf * TPB * KP * LDG;
#pragma unroll
    for (int r = 0; r < QG; ++r) {
      const int idx = tid + r * NT;
      if (idx < NQG) {
        const int pix = idx / GQ;
#pragma unroll
        for (int tt = 0; tt < TPB; ++tt) {
          const bool in = okG[tt][r];
          float4 v;
          v.x = in ? fmaf(rg[tt][r].x, gsc.x, gsh.x) : 0.f;
          v.y = (in && (VEC || g1)) ? fmaf(rg[tt][r].y, gsc.y, gsh.y) : 0.f;
          v.z = (in && (VEC || g2)) ? fmaf(rg[tt][r].z, gsc.z, gsh.z) : 0.f;
          v.w = (in && (VEC || g3)) ? fmaf(rg[tt][r].w, gsc.w, gsh.w) : 0.f;
          *reinterpret_cast<float4*>(Gb + (tt * KP + pix) * LDG + 4 * gq) = v;
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

  if (ch_begin < ch_end) {
    load(ch_begin);
    store(0);
  }
  __syncthreads();
  for (int ch = ch_begin; ch < ch_end; ++ch) {
    const int buf = (ch - ch_begin) & 1;
    const bool more = (ch + 1) < ch_end;
    if (more) load(ch + 1);
    __builtin_amdgcn_sched_barrier(0);
    const float* Db = Ds + buf * KP * LDD + wm * (TM * 32) + li;
    const float* Gb = Gs + buf * TPB * KP * LDG + wn * (TN * 32) + li;
#pragma unroll
    for (int kk = 0; kk < KP / 2; ++kk) {
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
    __builtin_amdgcn_sched_barrier(0);  // keep the selects + LDS writes (and their vmcnt waits) behind the MFMA block
    if (more) store(buf ^ 1);
    __syncthreads();
  }

  float* out = (a.splits > 1) ? a.partial + (int64_t)blockIdx.z * a.Cm * a.Cg * T : a.dw;
#pragma unroll
  for (int tt = 0; tt < TPB; ++tt) {
    const int t = (TPB == 1) ? tap0 : tt;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int cg = cg0 + wn * (TN * 32) + j * 32 + li;
#pragma unroll
      for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cm = cm0 + wm * (TM * 32) + i * 32 + mfma32_row(r, lane);
          if (cm < a.Cm && cg < a.Cg) out[((int64_t)cm * a.Cg + cg) * T + t] = acc[tt][i][j][r];
        }
      }
    }
  }
}

__global__ void sum_partials_kernel(const float* __restrict__ src, float* __restrict__ dst, int64_t n, int splits) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float s = 0.f;
    for (int k = 0; k < splits; ++k) s += src[(int64_t)k * n + i];
    dst[i] = s;
  }
}

// Many slices of a small tensor (narrow layers cut the pixel range into ~1000 slices): 16 outputs x 16 slice lanes per
// workgroup, 8 independent loads in flight per thread, LDS tree over the lanes.  Same summation order every run.
__global__ void __launch_bounds__(256) sum_partials_wide_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                int64_t n, int splits) {
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
    dst[i] = s;
  }
}

// ---- host side -------------------------------------------------------------------------------------
enum WgCfg { WG_128 = 0, WG_64, WG_TAPS9, WG_TAPS4, WG_32 };

static WgCfg pick_wg(const dfl_wgrad_args* a, bool vec = true) {
  const int T = a->KH * a->KW;
  const bool narrow = (a->Cm <= 64 || a->Cg <= 64) || !vec;   // the scalar-load variants exist for the narrow tiles only
  if (narrow) {
    if (a->KH == 3 && a->KW == 3) return WG_TAPS9;
    if (a->KH == 2 && a->KW == 2) return WG_TAPS4;
    return WG_32;
  }
  if (a->Cm >= 256 && a->Cg >= 256 && (int64_t)a->Cm * a->Cg * T >= 128ll * 128 * 1024) return WG_128;
  return WG_64;
}

static void wg_tile(WgCfg c, int* bm, int* bn, int* tpb) {
  switch (c) {
    case WG_128: *bm = 128; *bn = 128; *tpb = 1; break;
    case WG_64: *bm = 64; *bn = 64; *tpb = 1; break;
    case WG_TAPS9: *bm = 32; *bn = 32; *tpb = 9; break;
    case WG_TAPS4: *bm = 32; *bn = 32; *tpb = 4; break;
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
  k->vecG = (a->ldg % 4 == 0) && (a->ldg >= r4g) && aligned16(a->g) &&
            (a->in_scale == nullptr || (a->Cg % 4 == 0 && aligned16(a->in_scale) && aligned16(a->in_shift)));
  k->vecD = (a->ldd % 4 == 0) && (a->ldd >= r4d) && aligned16(a->d);
  return DFL_OK;
}

template <int WM, int WN, int TM, int TN, int TPB, bool VEC>
static int wg_launch(const WgK& k, hipStream_t s) {
  constexpr int BMc = WM * TM * 32, BNg = WN * TN * 32;
  const size_t lds = (size_t)(2 * KP * (BMc + 4) + 2 * TPB * KP * (BNg + 4)) * sizeof(float);
  const int tiles_g = (int)ceil_div(k.a.Cg, BNg);
  dim3 grid((unsigned)ceil_div(k.a.Cm, BMc), (unsigned)(tiles_g * (TPB == 1 ? k.T : 1)), (unsigned)k.a.splits);
  hipLaunchKernelGGL((wgrad_kernel<WM, WN, TM, TN, TPB, VEC>), grid, dim3(WM * WN * 64), lds, s, k);
  return check_launch("dfl_conv2d_wgrad");
}

}  // namespace dfl

extern "C" int dfl_wgrad_suggest_splits(const dfl_wgrad_args* a) {
  dfl::WgK k;
  int rc = dfl::wg_prepare(a, &k, false);
  if (rc != DFL_OK) return rc;
  int bm, bn, tpb;
  dfl::wg_tile(dfl::pick_wg(a, k.vecG && k.vecD), &bm, &bn, &tpb);
  const int64_t blocks = dfl::ceil_div(a->Cm, bm) * dfl::ceil_div(a->Cg, bn) * (tpb == 1 ? k.T : 1);
  int64_t s = dfl::ceil_div(1536, blocks);
  const int64_t max_by_work = k.nchunks / 8 > 0 ? k.nchunks / 8 : 1;  // >= 128 pixels per slice
  if (s > max_by_work) s = max_by_work;
  if (s > 2048) s = 2048;
  if (s < 1) s = 1;
  return (int)s;
}

extern "C" int dfl_wgrad_config(const dfl_wgrad_args* a) {
  dfl::WgK k;
  int rc = dfl::wg_prepare(a, &k, false);
  if (rc != DFL_OK) return rc;
  return (int)dfl::pick_wg(a, k.vecG && k.vecD);
}

extern "C" int dfl_conv2d_wgrad(const dfl_wgrad_args* a, dfl_stream_t stream) {
  dfl::WgK k;
  int rc = dfl::wg_prepare(a, &k, true);
  if (rc != DFL_OK) return rc;
  k.cps = (int)dfl::ceil_div(k.nchunks, a->splits);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool vec = k.vecG && k.vecD;
  switch (dfl::pick_wg(a, vec)) {
    case dfl::WG_128: return dfl::wg_launch<2, 2, 2, 2, 1, true>(k, s);
    case dfl::WG_64: return dfl::wg_launch<2, 2, 1, 1, 1, true>(k, s);
    case dfl::WG_TAPS9: return vec ? dfl::wg_launch<1, 1, 1, 1, 9, true>(k, s) : dfl::wg_launch<1, 1, 1, 1, 9, false>(k, s);
    case dfl::WG_TAPS4: return vec ? dfl::wg_launch<1, 1, 1, 1, 4, true>(k, s) : dfl::wg_launch<1, 1, 1, 1, 4, false>(k, s);
    default: return vec ? dfl::wg_launch<1, 1, 1, 1, 1, true>(k, s) : dfl::wg_launch<1, 1, 1, 1, 1, false>(k, s);
  }
}

extern "C" int dfl_sum_partials(const float* src, float* dst, int64_t n, int32_t splits, dfl_stream_t stream) {
  DFL_REQUIRE(src && dst && n >= 0 && splits >= 1, "dfl_sum_partials: bad args");
  if (n == 0) return DFL_OK;
  if (splits >= 32 && n <= (1 << 20)) {
    hipLaunchKernelGGL(dfl::sum_partials_wide_kernel, dim3((unsigned)dfl::ceil_div(n, 16)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), src, dst, n, (int)splits);
    return dfl::check_launch("dfl_sum_partials");
  }
  int64_t blocks = dfl::ceil_div(n, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dfl::sum_partials_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                     src, dst, n, (int)splits);
  return dfl::check_launch("dfl_sum_partials");
}
