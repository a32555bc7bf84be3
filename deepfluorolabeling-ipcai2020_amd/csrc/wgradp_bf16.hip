// Patch-resident weight gradient for bf16 tensors (math mode 4, "bf16 storage"):
//   dw[cm][cg][t] = sum over output pixels m of d[m][cm] * G(m, t, cg)
// with G the layer input gathered as dfl_conv2d gathers it (taps, stride, zero padding AFTER the optional BatchNorm affine).
// Reference: torch autograd of nn.Conv2d / nn.ConvTranspose2d at train_test_code/unet.py:93,207,211,218,240 (train.py:422).
// Contract: include/dfl_hip.h (dfl_conv2d_wgrad with g_bf16 = d_bf16 = 1).
//
// Same idea as convp_bf16.hip.  A workgroup owns a 64 x 64 (or smaller) tile of (cm, cg) for ALL taps and walks a range
// of pixel PATCHES.  Per patch it stages, once, the patch of d ([pixel][cm] bf16) and the patch of the gathered tensor with
// its halo ([pixel][cg] bf16, affine applied, padding written as zeros) in LDS; after that every tap of every k-step (16
// pixels) is an address offset into the same image.  The contraction index is the pixel -- the slow index of both NHWC
// operands -- so the MFMA fragments (8 consecutive pixels of one channel per lane) come out of LDS through gfx950's
// transposing read ds_read_b64_tr_b16 (16 lanes name 4 pixel rows x 16 channels and receive 4 pixels of one channel each).
// A wave owns one kernel ROW of taps (KW accumulator tiles): a d fragment is read once per k-step and used for its taps.  Row pitches are chosen so that the 4 rows x 64 bytes of a transposing read fall on 64
// distinct banks (pitch = 64 or 192 bytes mod 256).
// Waves: with 4 (cm, cg) tile pairs in the workgroup tile each wave owns a pair; with 2 or 1 pairs the waves also split
// the k-steps of a patch (2 or 4 phases) and add up through LDS at the end.  Output: partial[slice][t][cm][cg] fp32 (tap-major:
// full 128-byte rows), summed and transposed to torch's [cm][cg][t] by dfl_reduce_batch; a single slot writes dw itself.
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.h"
#include "convp.h"

namespace dfl {

constexpr uint32_t WPOOB = 0x80000000u;
typedef unsigned int wpu32x4 __attribute__((ext_vector_type(4)));
typedef short ws16x4_t __attribute__((ext_vector_type(4)));

struct WgP {
  dfl_wgrad_args a;
  int Mtot, T;
  int PH, PW, IPP, npy, npx, npatch;   // patch of the d grid (Hout x Wout per image)
  int IH, IW;                          // gathered pixels per patch and image
  int P16;                             // patch pixels rounded up to 16
  int CMT, CGT;                        // workgroup tile (multiples of 32, <= 64)
  int pairs, phases;                   // (cm32, cg32) pairs per workgroup, k-step phases (pairs * phases = 4)
  int zslices, patches_per_slice;      // grid.z, patches each slice walks
  int sd, sg;                          // row pitch of the d / g images in bytes
  int dupp_shift, gupp_shift;          // log2(CMT / 8), log2(CGT / 8)
  int g_off;                           // byte offset of the g image behind the d image
  int tab_off;                         // byte offset of the gathered-pixel offset table (P16 words) behind the g image
  int lds_bytes;
  long long* trace;                    // diagnosis build only (-DDFL_WGP_TRACE, docs/experiments/wgradp_trace.py)
  uint32_t g_bytes, d_bytes;
  uint32_t d2_bytes;                   // extent of the second dense tensor (dfl_wgrad_args.d_mode)
  int xcd_map;                         // workgroup -> (tile, slice) map, see the kernel
};

__device__ __forceinline__ float wbf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float wbf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t wpack_bf2(float a, float b) {
  const bf16x2_t h = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
  return __builtin_bit_cast(uint32_t, h);
}

// 8 consecutive pixels of one channel: two transposing reads (pixel rows at byte addresses r0 and r1 as seen by this lane)
__device__ __forceinline__ bf16x8_t wtr_read8(const unsigned char* base, uint32_t r0, uint32_t r1) {
  typedef __attribute__((address_space(3))) ws16x4_t* lds_p;
  const ws16x4_t x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + r0));
  const ws16x4_t y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + r1));
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7));
}

typedef unsigned int wpu32x2 __attribute__((ext_vector_type(2)));
// 4 consecutive pixel rows of one channel (one transposing read), as two dwords
__device__ __forceinline__ wpu32x2 wtr_read4(const unsigned char* base, uint32_t r0) {
  typedef __attribute__((address_space(3))) ws16x4_t* lds_p;
  return __builtin_bit_cast(wpu32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + r0)));
}

// KH x KW window.  4 * KH waves: wave w owns kernel row w % KH (KW taps, KW accumulator tiles) of "slot" w / KH; a slot is
// a (cm32, cg32) pair of the workgroup tile and, when the tile has fewer than 4 pairs, a phase of the k-steps.  12 waves of
// 48 accumulator registers (3x3) instead of 4 waves of 144: three waves per SIMD cover each other's LDS latency, the patch
// is staged by three times the threads, and the register budget leaves room for the prefetch below.
// Units (16 bytes) of the d / g images a workgroup may hold in flight in registers, per window height: the 3x3 kernel
// (768 threads) fills a CU alone; the 2x2 and 1x1 kernels (512 / 256 threads) keep their register count at 128 so that
// two / four workgroups share a CU and cover each other's barriers and load waits.
// (dbrb: the fused BatchNorm + ReLU backward operand keeps TWO tensors of the d patch in flight: smaller patches pay for its registers)
__host__ __device__ constexpr int wgp_max_d_units(int KH, bool dbrb = false) { return KH == 3 ? (dbrb ? 1536 : 2304) : 1024; }
__host__ __device__ constexpr int wgp_max_g_units(int KH, bool dbrb = false) { return KH == 3 ? (dbrb ? 3072 : 4608) : 2048; }

// DBRB: the dense operand is the BatchNorm + ReLU backward of (d = dy, d2 = r) formed while the patch is written to LDS
// (dfl_wgrad_args.d_mode), and the column sums of it -- the layer's bias gradient -- leave with the slice (bias_partial).
// BIAS: the column sums also leave when d is a plain tensor (the operand materialised by dfl_conv_args.x_out).
template <int KH, int KW, bool AFF, bool DBRB = false, bool BIAS = DBRB>
__global__ void __launch_bounds__(256 * KH, KH == 2 ? 4 : 3) wgradp_kernel(const WgP p) {
  constexpr int NT = 256 * KH;
  constexpr int T = KH * KW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const dfl_wgrad_args& a = p.a;
#ifdef DFL_WGP_TRACE
  const long long tr_entry = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int ty = wave % KH, slot = wave / KH;
  const int pair = slot % p.pairs, phase = slot / p.pairs;
  const int pairs_n = p.CGT >> 5;                       // pairs along cg
  const int pm = pair / pairs_n, pn = pair - pm * pairs_n;
  // ---- which tile, which pixel slice.  Workgroup L of the linear dispatch order runs on XCD L % 8 (observed; convp_bf16.hip),
  //      and each XCD has an L2 of its own.  With the hardware's order (cm tile fastest, then cg tile, then slice) the 8 XCDs
  //      share every slice: an XCD sees one or two cm tiles and ALL cg tiles, so the gathered tensor is fetched by all eight
  //      L2s.  xcd_map = 1 gives every XCD a contiguous run of the (slice, cm tile, cg tile) order instead: with 8 slices or
  //      more an XCD keeps whole slices -- both tensors are fetched once, by the L2 whose workgroups read them.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (p.xcd_map) {
    const int gx = gridDim.x, gy = gridDim.y, nt = gx * gy;
    const int total = nt * (int)gridDim.z, per = total >> 3;
    const int L = bx + gx * (by + gy * bz);
    // (the tail L >= 8 * per keeps its number but is decoded like the rest: with the hardware's decode there the map was
    //  not a bijection for gx, gy > 1 and total % 8 != 0 -- ADVICE r04)
    const int Lp = L < (per << 3) ? (L & 7) * per + (L >> 3) : L;
    bz = Lp / nt;
    const int t = Lp - bz * nt;
    bx = t / gy;                                              // cg tile fastest: the (larger) d patch stays in L2 across them
    by = t - bx * gy;
  }
  const int cm0 = bx * p.CMT, cg0 = by * p.CGT;
  unsigned char* Ds = smem;
  unsigned char* Gs = smem + p.g_off;

  __amdgpu_buffer_rsrc_t rsD = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.d), 0, (int)p.d_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsG = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.g), 0, (int)p.g_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsD2 = rsD;
  if constexpr (DBRB) rsD2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.d2), 0, (int)p.d2_bytes, 0x00020000);

  f32x16 acc[KW];
#pragma unroll
  for (int t = 0; t < KW; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // transposing-read lane geometry inside a 32-channel tile: pixel row 8*lh + (lane & 15) / 4 (second read: + 4),
  // channels 16 * ((lane >> 4) & 1) + 4 * (lane & 3) .. + 3
  const int trow = 8 * lh + ((lane & 15) >> 2);
  const uint32_t tcb = (uint32_t)((16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
  const uint32_t d_col = (uint32_t)(pm * 64) + tcb, g_col = (uint32_t)(pn * 64) + tcb;

  // staging: 16-byte units (8 channels); d image: P16 rows x CMT channels, g image: IPP*IH*IW rows x CGT channels
  const int dupp = p.CMT >> 3, gupp = p.CGT >> 3;
  const int npix_g = p.IPP * p.IH * p.IW;
  const int per_img = p.npy * p.npx;
  uint32_t* gtab = reinterpret_cast<uint32_t*>(smem + p.tab_off);   // LDS offset of the gathered pixel of every patch pixel

  // Patch pipeline: the global loads of patch i + 1 are issued into registers before the k-steps of patch i run and are
  // written to LDS after them.  Everything about a unit that does not depend on WHICH patch is staged is worked out once:
  // its position inside the patch (y | x << 12 | image << 24 | live << 31) and its byte offset from the patch origin;
  // per patch a unit costs a bounds test against the image and one add.
  constexpr int MAXD = wgp_max_d_units(KH, DBRB) / NT, MAXG = wgp_max_g_units(KH, DBRB) / NT;   // host: P16 * dupp and npix_g * gupp stay below these
  wpu32x4 dreg[MAXD], greg[MAXG];
  wpu32x4 d2reg[DBRB ? MAXD : 1];
  float bsum = 0.f;                                               // DBRB: this thread's share of the bias gradient (one channel)
  const bool bias_on = BIAS && a.bias_partial != nullptr && by == 0;
  uint32_t dpos[MAXD], gpos[MAXG];
  uint32_t gok = 0;
  const int ddk = NT >> p.dupp_shift, gdk = NT >> p.gupp_shift;
  const int dcq = tid & (dupp - 1), gcq = tid & (gupp - 1);
  const int dc = cm0 + dcq * 8, gc = cg0 + gcq * 8;
  const uint32_t dpitch = (uint32_t)a.ldd * 2u, gpitch = (uint32_t)a.ldg * 2u, d2pitch = (uint32_t)a.ldd2 * 2u;
  const int nd = (p.P16 * dupp + NT - 1) / NT, ng = (npix_g * gupp + NT - 1) / NT;   // units per thread actually in use (uniform)
  {
    const int PHW = p.PH * p.PW;
#pragma unroll
    for (int u = 0; u < MAXD; ++u) {
      const int k = (tid >> p.dupp_shift) + u * ddk;
      const int img = k / PHW, r = k - img * PHW;
      const int y = r / p.PW, x = r - y * p.PW;
      const bool live = k < p.P16 && img < p.IPP && dc < a.Cm;
      dpos[u] = (uint32_t)y | ((uint32_t)x << 12) | ((uint32_t)(img & 127) << 24) | (live ? 0x80000000u : 0u);
    }
    const int IHW = p.IH * p.IW;
#pragma unroll
    for (int u = 0; u < MAXG; ++u) {
      const int k = (tid >> p.gupp_shift) + u * gdk;
      const int img = k / IHW, r = k - img * IHW;
      const int y = r / p.IW, x = r - y * p.IW;
      const bool live = k < npix_g && gc < a.Cg;
      gpos[u] = (uint32_t)y | ((uint32_t)x << 12) | ((uint32_t)(img & 127) << 24) | (live ? 0x80000000u : 0u);
    }
    for (int j = tid; j < p.P16; j += NT) {
      const int img = j / PHW, r = j - img * PHW;
      const int y = r / p.PW, x = r - y * p.PW;
      // rows beyond the patch: d is zero there, any valid g row will do
      gtab[j] = img < p.IPP ? (uint32_t)((img * p.IH + y * a.stride) * p.IW + x * a.stride) * (uint32_t)p.sg : 0u;
    }
  }
  // BatchNorm affine of the gathered tensor (applied when a unit is written to LDS): this thread's 8 channels, kept in LDS
  // behind the offset table -- 16 registers that the 3x3 kernel does not have
  float* aff = reinterpret_cast<float*>(smem + p.tab_off + p.P16 * 4);      // [2][CGT]
  float* dco = aff + 2 * p.CGT;                                             // [3][CMT]: A, B, C of this tile's d channels
  // issue: the loads of one patch, unit by unit -- issue_setup fixes the patch, issue_d(u) / issue_g(u) request one 16-byte unit each
  int is_nleft = 0, is_oy0 = 0, is_ox0 = 0, is_ybase = 0, is_xbase = 0;
  uint32_t is_dbase = 0, is_dbase2 = 0, is_gbase = 0;
  auto issue_setup = [&](int patch, bool live) {
    const int pg = patch / per_img, pr = patch - pg * per_img;
    const int ppy = pr / p.npx, ppx = pr - ppy * p.npx;
    const int img0 = pg * p.IPP;
    is_oy0 = ppy * p.PH;
    is_ox0 = ppx * p.PW;
    is_nleft = live ? a.N - img0 : 0;                              // images of this patch that exist
    // d: rows = patch pixels in patch order (image, row, column), zeros beyond the patch / image
    is_dbase = (uint32_t)(((img0 * a.Hout + is_oy0) * a.Wout + is_ox0) * a.ldd + dc) * 2u;
    is_dbase2 = DBRB ? (uint32_t)(((img0 * a.Hout + is_oy0) * a.Wout + is_ox0) * a.ldd2 + dc) * 2u : 0u;
    // g: the gathered pixels of the patch with their halo, zero outside the image
    is_ybase = is_oy0 * a.stride - a.pad;
    is_xbase = is_ox0 * a.stride - a.pad;
    is_gbase = (uint32_t)(((img0 * a.Hin + is_ybase) * a.Win + is_xbase) * a.ldg + gc) * 2u;   // may wrap: base + rel is what counts
    gok = 0;
  };
  auto issue_d = [&](int u) {
    if (u < nd) {
      const uint32_t q = dpos[u];
      const int qy = (int)(q & 0xfffu), qx = (int)((q >> 12) & 0xfffu), qi = (int)((q >> 24) & 127u);
      const bool ok = (int)q < 0 && qi < is_nleft && is_oy0 + qy < a.Hout && is_ox0 + qx < a.Wout;
      const uint32_t rel = (uint32_t)((qi * a.Hout + qy) * a.Wout + qx) * dpitch;
      dreg[u] = __builtin_amdgcn_raw_buffer_load_b128(rsD, ok ? is_dbase + rel : WPOOB, 0, 0);
      if constexpr (DBRB) {
        const uint32_t rel2 = (uint32_t)((qi * a.Hout + qy) * a.Wout + qx) * d2pitch;
        d2reg[u] = __builtin_amdgcn_raw_buffer_load_b128(rsD2, ok ? is_dbase2 + rel2 : WPOOB, 0, 0);
      }
    }
  };
  auto issue_g = [&](int u) {
    if (u < ng) {
      const uint32_t q = gpos[u];
      const int qy = (int)(q & 0xfffu), qx = (int)((q >> 12) & 0xfffu), qi = (int)((q >> 24) & 127u);
      const bool ok = (int)q < 0 && qi < is_nleft && (unsigned)(is_ybase + qy) < (unsigned)a.Hin && (unsigned)(is_xbase + qx) < (unsigned)a.Win;
      gok |= ok ? (1u << u) : 0u;
      const uint32_t rel = (uint32_t)((qi * a.Hin + qy) * a.Win + qx) * gpitch;
      greg[u] = __builtin_amdgcn_raw_buffer_load_b128(rsG, ok ? is_gbase + rel : WPOOB, 0, 0);
    }
  };
  auto issue = [&](int patch, bool live) {
    issue_setup(patch, live);
#pragma unroll
    for (int u = 0; u < MAXD; ++u) issue_d(u);
#pragma unroll
    for (int u = 0; u < MAXG; ++u) issue_g(u);
  };
  auto commit = [&]() {
#pragma unroll
    for (int u = 0; u < MAXD; ++u) {
      const int k = (tid >> p.dupp_shift) + u * ddk;
      if (u < nd && k < p.P16) {
        wpu32x4 v = dreg[u];
        if constexpr (DBRB) {     // [r > 0] * (A dy + B r + C); rows outside the patch / image were loaded as zeros: r = 0, value 0
          const wpu32x4 r = d2reg[u];
          const float4 a0 = *reinterpret_cast<const float4*>(dco + dcq * 8), a1 = *reinterpret_cast<const float4*>(dco + dcq * 8 + 4);
          const float4 b0 = *reinterpret_cast<const float4*>(dco + p.CMT + dcq * 8), b1 = *reinterpret_cast<const float4*>(dco + p.CMT + dcq * 8 + 4);
          const float4 c0 = *reinterpret_cast<const float4*>(dco + 2 * p.CMT + dcq * 8), c1 = *reinterpret_cast<const float4*>(dco + 2 * p.CMT + dcq * 8 + 4);
          auto brb = [](float dy, float rv, float A, float B, float Cc) { return rv > 0.f ? fmaf(A, dy, fmaf(B, rv, Cc)) : 0.f; };
          v.x = wpack_bf2(brb(wbf_lo(v.x), wbf_lo(r.x), a0.x, b0.x, c0.x), brb(wbf_hi(v.x), wbf_hi(r.x), a0.y, b0.y, c0.y));
          v.y = wpack_bf2(brb(wbf_lo(v.y), wbf_lo(r.y), a0.z, b0.z, c0.z), brb(wbf_hi(v.y), wbf_hi(r.y), a0.w, b0.w, c0.w));
          v.z = wpack_bf2(brb(wbf_lo(v.z), wbf_lo(r.z), a1.x, b1.x, c1.x), brb(wbf_hi(v.z), wbf_hi(r.z), a1.y, b1.y, c1.y));
          v.w = wpack_bf2(brb(wbf_lo(v.w), wbf_lo(r.w), a1.z, b1.z, c1.z), brb(wbf_hi(v.w), wbf_hi(r.w), a1.w, b1.w, c1.w));
        }
        *reinterpret_cast<wpu32x4*>(Ds + (uint32_t)k * (uint32_t)p.sd + (uint32_t)dcq * 16u) = v;
      }
    }
#pragma unroll
    for (int u = 0; u < MAXG; ++u) {
      const int pix = (tid >> p.gupp_shift) + u * gdk;
      if (u < ng && pix < npix_g) {
        wpu32x4 v = greg[u];
        if constexpr (AFF) {   // zero padding applies AFTER the BatchNorm affine: outside pixels stay 0
          if ((gok >> u) & 1u) {
            const float4 s0 = *reinterpret_cast<const float4*>(aff + gcq * 8), s1 = *reinterpret_cast<const float4*>(aff + gcq * 8 + 4);
            const float4 h0 = *reinterpret_cast<const float4*>(aff + p.CGT + gcq * 8), h1 = *reinterpret_cast<const float4*>(aff + p.CGT + gcq * 8 + 4);
            const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
            v.x = wpack_bf2(fmaf(wbf_lo(v.x), sc[0], sh[0]), fmaf(wbf_hi(v.x), sc[1], sh[1]));
            v.y = wpack_bf2(fmaf(wbf_lo(v.y), sc[2], sh[2]), fmaf(wbf_hi(v.y), sc[3], sh[3]));
            v.z = wpack_bf2(fmaf(wbf_lo(v.z), sc[4], sh[4]), fmaf(wbf_hi(v.z), sc[5], sh[5]));
            v.w = wpack_bf2(fmaf(wbf_lo(v.w), sc[6], sh[6]), fmaf(wbf_hi(v.w), sc[7], sh[7]));
          }
        }
        *reinterpret_cast<wpu32x4*>(Gs + (uint32_t)pix * (uint32_t)p.sg + (uint32_t)gcq * 16u) = v;
      }
    }
  };

  const int pbegin = bz * p.patches_per_slice;
  const int pend = min(pbegin + p.patches_per_slice, p.npatch);
  const int nsteps = p.P16 >> 4;
  const uint32_t gadd = g_col + (uint32_t)(ty * p.IW) * (uint32_t)p.sg;     // this wave's kernel row, this lane's channels
  const uint32_t dstep = (uint32_t)(16 * p.phases) * (uint32_t)p.sd;
  const bool al8 = KW == 3 && a.stride == 1 && (p.PW & 7) == 0;          // (uniform)
#ifdef DFL_WGP_TRACE
  long long tr[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  tr[0] = __builtin_amdgcn_s_memtime();
  tr[6] = __builtin_amdgcn_s_memrealtime();
#define WTR(i, t0) tr[i] += __builtin_amdgcn_s_memtime() - (t0);
#define WT0(name) const long long name = __builtin_amdgcn_s_memtime();
#else
#define WTR(i, t0)
#define WT0(name)
#endif
  issue(pbegin, pbegin < pend);
  // (round 6: the first patch's loads go out BEFORE the tables are derived -- with live statistics a column's coefficients are 16 fp64
  // loads and a square root: a memory round trip that used to stand in front of the first request of every launch)
  if constexpr (AFF) {
    for (int c = tid; c < p.CGT; c += NT) {
      aff[c] = (cg0 + c < a.Cg) ? a.in_scale[cg0 + c] : 1.f;
      aff[p.CGT + c] = (cg0 + c < a.Cg) ? a.in_shift[cg0 + c] : 0.f;
    }
  }
  if constexpr (DBRB) {
    for (int c = tid; c < p.CMT; c += NT) {
      const bool live = a.coef != nullptr && cm0 + c < a.Cm;
      float A = live ? a.coef[cm0 + c] : 1.f, B = live ? a.coef[a.Cm + cm0 + c] : 0.f, Cc = live ? a.coef[2 * a.Cm + cm0 + c] : 0.f;
      if (a.coef_tot != nullptr && cm0 + c < a.Cm)      // live statistics (include/dfl_hip.h): derived here
        bn_live_coef(a.coef_tot, a.bn_gamma, a.bn_mean, a.bn_invstd, a.bn_count, a.Cm, cm0 + c, &A, &B, &Cc);
      dco[c] = A;
      dco[p.CMT + c] = B;
      dco[2 * p.CMT + c] = Cc;
    }
  }
  if constexpr (AFF || DBRB) __syncthreads();             // the first commit() reads these tables
  for (int patch = pbegin; patch < pend; ++patch) {
    WT0(tb0)
    if (patch != pbegin) __syncthreads();               // every wave is done reading the previous images
    WTR(1, tb0)
    WT0(tc0)
    commit();
    WTR(2, tc0)
    WT0(tb1)
    __syncthreads();
    WTR(1, tb1)
    WT0(ti0)
    issue(patch + 1, patch + 1 < pend);
    WTR(3, ti0)
    if constexpr (BIAS) {
      // bias gradient: column sums of the d image as stored -- thread t owns channel t % CMT and every (NT / CMT)-th pixel row
      // (rows beyond the patch are zero); one accumulator register instead of eight in the staging path
      if (bias_on) {
        const int bc = tid & (p.CMT - 1), bstride = NT / p.CMT;
        for (int r = tid / p.CMT; r < p.P16; r += bstride)
          bsum += __uint_as_float((uint32_t)*reinterpret_cast<const unsigned short*>(Ds + (uint32_t)r * (uint32_t)p.sd + (uint32_t)bc * 2u) << 16);
      }
    }
    WT0(tk0)
    // ---- k-steps of this patch (16 pixels each), this wave's phase; this lane's two pixel rows of a step are the
    //      patch pixels j0 = 16 ks + trow and j0 + 4: their d rows are j0 * sd, their g rows come from the offset table
    uint32_t dr0 = (uint32_t)(phase * 16 + trow) * (uint32_t)p.sd + d_col;
    const uint32_t* tp = gtab + phase * 16 + trow;
    int ks = phase;
    auto kstep = [&]() {
      const uint32_t gr0 = tp[0] + gadd, gr1 = tp[4] + gadd;
      const bf16x8_t df = wtr_read8(Ds, dr0, dr0 + 4u * (uint32_t)p.sd);
      bf16x8_t gf[KW];
      if (KW == 3 && al8) {
        // the 8 pixels of this lane's half of the k-step are consecutive pixels of one image row (PW % 8 == 0, stride 1): 12
        // consecutive gathered pixels -- three transposing reads -- hold all three taps of the kernel row; tap 1 is the same
        // registers shifted by one pixel (v_alignbit).  3 LDS reads instead of 6 in a loop that is bound by them (round 5)
        const wpu32x2 g0 = wtr_read4(Gs, gr0), g1 = wtr_read4(Gs, gr1), g2 = wtr_read4(Gs, gr0 + 8u * (uint32_t)p.sg);
        gf[0] = __builtin_bit_cast(bf16x8_t, (wpu32x4){g0.x, g0.y, g1.x, g1.y});
        gf[1] = __builtin_bit_cast(bf16x8_t, (wpu32x4){__builtin_amdgcn_alignbit(g0.y, g0.x, 16), __builtin_amdgcn_alignbit(g1.x, g0.y, 16),
                                                        __builtin_amdgcn_alignbit(g1.y, g1.x, 16), __builtin_amdgcn_alignbit(g2.x, g1.y, 16)});
        gf[KW - 1] = __builtin_bit_cast(bf16x8_t, (wpu32x4){g0.y, g1.x, g1.y, g2.x});
      } else {
#pragma unroll
        for (int t = 0; t < KW; ++t) gf[t] = wtr_read8(Gs, gr0 + (uint32_t)(t * p.sg), gr1 + (uint32_t)(t * p.sg));
      }
#pragma unroll
      for (int t = 0; t < KW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, gf[t], acc[t], 0, 0, 0);
      dr0 += dstep;
      tp += 16 * p.phases;
      ks += p.phases;
    };
    while (ks < nsteps) kstep();
    WTR(4, tk0)
  }
#ifdef DFL_WGP_TRACE
  tr[5] = __builtin_amdgcn_s_memtime();
#endif

  // ---- bias gradient of this pixel slice (DBRB): the threads that staged the same 8 channels add up through LDS, fixed order
  if constexpr (BIAS) {
    if (a.bias_partial != nullptr && by == 0) {
      __syncthreads();
      float* bred = reinterpret_cast<float*>(smem);     // [NT]
      bred[tid] = bsum;
      __syncthreads();
      for (int c = tid; c < p.CMT; c += NT) {
        float t = 0.f;
        for (int th = c; th < NT; th += p.CMT) t += bred[th];
        if (cm0 + c < a.Cm) a.bias_partial[(int64_t)bz * a.Cm + cm0 + c] = t;
      }
    }
  }

  // ---- waves that split the k-steps of the patches (phases > 1) add their accumulators through LDS, tap by tap, in a
  //      fixed order; the waves of phase 0 then own the workgroup's result for their (cm, cg) pair and kernel row
  if (p.phases > 1) {
    float* red = reinterpret_cast<float*>(smem);        // [phases - 1][pairs][KH][16][64]
#pragma unroll
    for (int t = 0; t < KW; ++t) {
      __syncthreads();
      if (phase > 0) {
        float* dst = red + ((((phase - 1) * p.pairs + pair) * KH + ty) * 16) * 64 + lane;
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[r * 64] = acc[t][r];
      }
      __syncthreads();
      if (phase == 0) {
        for (int ph = 1; ph < p.phases; ++ph) {
          const float* src = red + ((((ph - 1) * p.pairs + pair) * KH + ty) * 16) * 64 + lane;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] += src[r * 64];
        }
      }
    }
    if (phase > 0 && p.zslices > 1) return;
  }
#ifdef DFL_WGP_TRACE
  const long long tr_main_end = __builtin_amdgcn_s_memtime();
#endif
  // ---- output
  const bool sliced = p.zslices > 1;
  const int cgl = pn * 32 + li, cg = cg0 + cgl;
  if (sliced) {
    // one partial slot per pixel slice, tap-major [t][cm][cg]: every accumulator row is a 128-byte store
    float* out = a.partial + (int64_t)bz * a.Cm * a.Cg * T;
#pragma unroll
    for (int tx = 0; tx < KW; ++tx) {
      const int t = ty * KW + tx;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int cm = cm0 + pm * 32 + mfma32_row(r, lane);
        if (cm < a.Cm && cg < a.Cg) out[((int64_t)t * a.Cm + cm) * a.Cg + cg] = acc[tx][r];
      }
    }
  } else {
    // a single slot writes dw in torch's order [cm][cg][t] (a 4-byte store every T * 4 bytes from the accumulators: 13 - 17 us for
    // a 64 x 64 x 9 tile, round 5 phase clocks): the tile goes through LDS ([cm][cg][T] floats -- the T * CGT floats of a cm row are
    // contiguous in dw as well) and leaves in 16-byte stores by all threads
    __syncthreads();                                      // images and reduction scratch have been read
    float* tile = reinterpret_cast<float*>(smem);
    if (phase == 0) {
#pragma unroll
      for (int tx = 0; tx < KW; ++tx) {
        const int t = ty * KW + tx;
#pragma unroll
        for (int r = 0; r < 16; ++r) tile[((pm * 32 + mfma32_row(r, lane)) * p.CGT + cgl) * T + t] = acc[tx][r];
      }
    }
    __syncthreads();
    const int row4 = p.CGT * T / 4;                       // float4 per cm row of the tile
    const int valid4 = min(p.CGT, a.Cg - cg0) * T / 4;    // (Cg % 8 == 0: whole float4)
    for (int idx = tid; idx < p.CMT * row4; idx += NT) {
      const int row = idx / row4, c4 = idx - row * row4;
      if (cm0 + row < a.Cm && c4 < valid4)
        *reinterpret_cast<float4*>(a.dw + ((int64_t)(cm0 + row) * a.Cg + cg0) * T + c4 * 4) = *reinterpret_cast<const float4*>(tile + (row * p.CGT) * T + c4 * 4);
    }
  }
#ifdef DFL_WGP_TRACE
  __builtin_amdgcn_s_waitcnt(0);                       // (the stores of this wave have left)
  if (tid == 0 && p.trace != nullptr) {
    long long* sink = p.trace + (int64_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 12;
    for (int i = 0; i < 6; ++i) sink[i] = tr[i];
    sink[6] = tr_main_end;
    sink[7] = tr[6];
    sink[8] = __builtin_amdgcn_s_memrealtime();
    sink[10] = tr_entry;
    sink[11] = __builtin_amdgcn_s_memtime();
    sink[9] = 1;
  }
#endif
}

// ---- host side ------------------------------------------------------------------------------------------------------

static int pitch_for(int channels) {        // bytes per staged pixel: data + pad so that pitch = 64 or 192 (mod 256)
  const int data = channels * 2;
  int pitch = data;
  while (pitch % 256 != 64 && pitch % 256 != 192) pitch += 16;
  return pitch;
}

static int wgp_plan(const dfl_wgrad_args* a, WgP* p, bool need_out) {
  DFL_REQUIRE(a != nullptr, "dfl_conv2d_wgrad: null args");
  DFL_REQUIRE(a->N > 0 && a->Hin > 0 && a->Win > 0 && a->Cg > 0 && a->Cm > 0, "dfl_conv2d_wgrad: bad sizes");
  DFL_REQUIRE(a->KH > 0 && a->KW > 0 && a->stride > 0 && a->pad >= 0, "dfl_conv2d_wgrad: bad window");
  const int T = a->KH * a->KW;
  DFL_REQUIRE((a->KH == 1 && a->KW == 1) || (a->KH == 2 && a->KW == 2) || (a->KH == 3 && a->KW == 3),
              "dfl_conv2d_wgrad (bf16): 1x1, 2x2 and 3x3 windows");
  const int ho = (a->Hin + 2 * a->pad - a->KH) / a->stride + 1;
  const int wo = (a->Win + 2 * a->pad - a->KW) / a->stride + 1;
  DFL_REQUIRE(ho == a->Hout && wo == a->Wout, "dfl_conv2d_wgrad: Hout/Wout (%d,%d) do not match the window (%d,%d)", a->Hout,
              a->Wout, ho, wo);
  DFL_REQUIRE(a->Cg % 8 == 0 && a->Cm % 8 == 0 && a->ldg % 8 == 0 && a->ldd % 8 == 0 && aligned16(a->g) && aligned16(a->d),
              "dfl_conv2d_wgrad (bf16): channel counts and pixel strides must be multiples of 8, tensors 16-byte aligned");
  DFL_REQUIRE((a->in_scale == nullptr) == (a->in_shift == nullptr), "dfl_conv2d_wgrad: in_scale/in_shift go together");
  if (need_out) {
    DFL_REQUIRE(a->g && a->d, "dfl_conv2d_wgrad: g and d are required");
    DFL_REQUIRE(a->splits >= 1, "dfl_conv2d_wgrad: splits >= 1");
    DFL_REQUIRE(a->splits > 1 ? a->partial != nullptr : a->dw != nullptr, "dfl_conv2d_wgrad: output buffer missing");
  }
  memset(p, 0, sizeof(*p));
  p->a = *a;
  p->T = T;
  const int64_t M = (int64_t)a->N * a->Hout * a->Wout;
  DFL_REQUIRE(M < (1ll << 31), "dfl_conv2d_wgrad: too many pixels");
  p->Mtot = (int)M;
  const int64_t gb = (((int64_t)a->N * a->Hin * a->Win - 1) * a->ldg + a->Cg) * 2;
  const int64_t db = ((M - 1) * a->ldd + a->Cm) * 2;
  const int64_t lim = (1ll << 31) - 4096;
  DFL_REQUIRE(gb < lim && db < lim, "dfl_conv2d_wgrad (bf16): tensors must stay below 2 GiB");
  p->g_bytes = (uint32_t)gb;
  p->d_bytes = (uint32_t)db;
  if (a->d_mode != 0) {
    DFL_REQUIRE(a->coef_tot == nullptr || (a->coef == nullptr && a->bn_gamma && a->bn_mean && a->bn_invstd && a->bn_count > 0),
                "dfl_conv2d_wgrad (bf16): coef_tot replaces coef and needs bn_gamma, bn_mean, bn_invstd, bn_count");
    DFL_REQUIRE(a->d_mode == 1 && a->KH == 3 && a->d2 != nullptr && a->ldd2 % 8 == 0 && aligned16(a->d2),
                "dfl_conv2d_wgrad (bf16): d_mode 1 needs a 3x3 window and d2 (16-byte aligned, ldd2 %% 8 == 0)");
    const int64_t d2b = ((M - 1) * a->ldd2 + a->Cm) * 2;
    DFL_REQUIRE(d2b < lim, "dfl_conv2d_wgrad (bf16): tensors must stay below 2 GiB");
    p->d2_bytes = (uint32_t)d2b;
  }
  DFL_REQUIRE(a->bias_partial == nullptr || a->KH == 3, "dfl_conv2d_wgrad (bf16): bias_partial is implemented by the 3x3 kernel");
  // workgroup tile and wave roles
  p->CMT = a->Cm > 32 ? 64 : 32;
  p->CGT = a->Cg > 32 ? 64 : 32;
  {
    // Layers with many (cm, cg) tiles but not enough of them to fill the chip (levels 3-4 of the paper network: 16 ... 128
    // tiles of 64 x 64) used to be cut into 16 ... 2 pixel slices, each leaving a full-size fp32 partial gradient (37.7 MB per
    // 3x3 layer, written here and read back by dfl_reduce_batch: the largest wasted traffic of a step).  With
    // 32 x 32 tiles the same workgroup count needs a quarter of the slices -- none at all from 256 tiles on; the patch of d /
    // g is then staged by twice as many workgroups, which these small, L2-resident activations can afford.
    static const int small_from = [] {
      const char* e = getenv("DFL_WGP_SMALL_TILES");   // use 32 x 32 tiles from this many 64 x 64 tiles on (0: never)
      return e ? atoi(e) : 4;                          // (round 4: 16 -> 4, the same step time with 140 MB less partial sums per step)
    }();
    constexpr int small_from21 = 4;                    // ... for the 2x2 / 1x1 windows (64 x 64 tiles there: kernels 0.437 -> 0.395 ms, batched sums 0.14 -> 0.21: equal, round 4)
    const int64_t tiles64 = ceil_div(a->Cm, 64) * ceil_div(a->Cg, 64);
    const int from = a->KH == 3 ? small_from : small_from21;
    if (from > 0 && a->Cm > 32 && a->Cg > 32 && tiles64 >= from && tiles64 < 256) p->CMT = p->CGT = 32;
  }
  p->pairs = (p->CMT / 32) * (p->CGT / 32);
  p->phases = 4 / p->pairs;
  p->dupp_shift = p->CMT == 64 ? 3 : 2;
  p->gupp_shift = p->CGT == 64 ? 3 : 2;
  p->sd = pitch_for(p->CMT);
  p->sg = pitch_for(p->CGT);
  // Patch: the (images, rows, columns) box of output pixels that costs the least over the whole layer -- bytes staged
  // (halo and the padding of the last 16-pixel step included) plus a fixed charge per patch for its two barriers and the
  // pipeline bubble -- within what a thread can hold in flight (2048 / 4096 sixteen-byte units of d / g per workgroup)
  // and the LDS budget.  Whole images are grouped while they fit.
  const int64_t max_du = wgp_max_d_units(a->KH, a->d_mode != 0), max_gu = wgp_max_g_units(a->KH, a->d_mode != 0);
  const int lds_cap = a->KH == 3 ? 120 * 1024 : (a->KH == 2 ? 72 * 1024 : 52 * 1024);   // (100 / 140 KB for the 3x3 window: flat or slower, round 3)
  auto geom = [&](int ipp_, int ph_, int pw_, int64_t* lds, int64_t* du, int64_t* gu) {
    const int p16 = (ipp_ * ph_ * pw_ + 15) / 16 * 16;
    const int ih = (ph_ - 1) * a->stride + a->KH, iw = (pw_ - 1) * a->stride + a->KW;
    *du = (int64_t)p16 * (p->CMT / 8);
    *gu = (int64_t)ipp_ * ih * iw * (p->CGT / 8);
    *lds = (int64_t)p16 * p->sd + (int64_t)ipp_ * ih * iw * p->sg + (int64_t)p16 * 4 + 512;
  };
  auto fits = [&](int ipp_, int ph_, int pw_) {
    int64_t lds, du, gu;
    geom(ipp_, ph_, pw_, &lds, &du, &gu);
    return lds <= lds_cap && du <= max_du && gu <= max_gu && ipp_ <= 127 && ph_ < 4096 && pw_ < 4096;
  };
  int ipp = 0, ph = 0, pw = 0;
  double best_cost = 1e300;
  // the search depends on few numbers and runs at every launch: remembered per shape
  struct PatchMemo { int key[8]; int ipp, ph, pw; };
  static std::mutex memo_mu;
  static std::vector<PatchMemo> memo;
  const int mkey[8] = {a->N, a->Hout, a->Wout, a->stride, a->KH + 16 * (a->d_mode != 0 ? 1 : 0), a->KW, p->CMT, p->CGT};
  bool found = false;
  {
    std::lock_guard<std::mutex> lock(memo_mu);
    for (const PatchMemo& m : memo)
      if (memcmp(m.key, mkey, sizeof(mkey)) == 0) {
        ipp = m.ipp; ph = m.ph; pw = m.pw;
        found = true;
        break;
      }
  }
  auto consider = [&](int ipp_, int ph_, int pw_) {
    if (ipp_ < 1 || ph_ < 1 || pw_ < 1 || ph_ > a->Hout || pw_ > a->Wout || !fits(ipp_, ph_, pw_)) return;
    int64_t lds, du, gu;
    geom(ipp_, ph_, pw_, &lds, &du, &gu);
    const double npatch = (double)ceil_div(a->N, ipp_) * (double)ceil_div(a->Hout, ph_) * (double)ceil_div(a->Wout, pw_);
    constexpr double charge = 24.0 * 1024.0;          // bytes-equivalent of a patch's fixed costs (barriers, load round trip; 4 ... 128 KB measured: flat, round 3)
    double cost = npatch * ((double)(du + gu) * 16.0 + charge);
    if (a->KH == 3 && a->stride == 1 && pw_ % 8 == 0) cost *= 0.92;   // (runs of 8 pixels: the three taps of a kernel row from one set of reads)
    if (cost < best_cost) {
      best_cost = cost;
      ipp = ipp_;
      ph = ph_;
      pw = pw_;
    }
  };
  if (!found) {
    for (int n = a->N; n >= 1; --n) consider(n, a->Hout, a->Wout);      // whole images
    for (int pw_ = 4; pw_ <= a->Wout; ++pw_) {
      if (pw_ != a->Wout && pw_ % 4 != 0) continue;
      for (int ph_ = 1; ph_ <= a->Hout; ++ph_) {
        if (!fits(1, ph_, pw_)) break;
        consider(1, ph_, pw_);
      }
    }
    if (ipp == 0) consider(1, 1, a->Wout < 4 ? a->Wout : 4);
    DFL_REQUIRE(ipp > 0, "dfl_conv2d_wgrad (bf16): no patch of this layer fits the staging limits");
    PatchMemo m;
    memcpy(m.key, mkey, sizeof(mkey));
    m.ipp = ipp; m.ph = ph; m.pw = pw;
    std::lock_guard<std::mutex> lock(memo_mu);
    if (memo.size() < 4096) memo.push_back(m);
  }
  p->IPP = ipp;
  p->PH = ph;
  p->PW = pw;
  p->P16 = (ipp * ph * pw + 15) / 16 * 16;
  p->npy = (int)ceil_div(a->Hout, ph);
  p->npx = (int)ceil_div(a->Wout, pw);
  p->npatch = (int)ceil_div(a->N, ipp) * p->npy * p->npx;
  p->IH = (ph - 1) * a->stride + a->KH;
  p->IW = (pw - 1) * a->stride + a->KW;
  p->g_off = (p->P16 * p->sd + 255) / 256 * 256;
  p->tab_off = (p->g_off + (ipp * p->IH * p->IW + 2) * p->sg + 255) / 256 * 256;   // (+ 2 pixels: the third read of a row's last run of 8)
  p->lds_bytes = p->tab_off + p->P16 * 4 + 2 * p->CGT * 4 + 3 * p->CMT * 4;
  if ((a->d_mode != 0 || a->bias_partial != nullptr) && p->lds_bytes < 256 * a->KH * 4) p->lds_bytes = 256 * a->KH * 4;   // room for the bias-gradient sums
  const int red_bytes = (p->phases - 1) * p->pairs * a->KH * 16 * 64 * 4;   // room for the cross-phase sums
  if (p->lds_bytes < red_bytes) p->lds_bytes = red_bytes;
  return DFL_OK;
}

// pixel slices so that the chip sees ~256 workgroups (DFL_WGP_WGS; measured 256 / 512 / 1024: 1.50 / 1.82 / 2.19 ms of weight
// gradients per step and 0.43 / 0.74 / 1.10 ms of partial sums)
static int wgp_slices(const WgP& p) {
  const int64_t tiles = ceil_div(p.a.Cm, p.CMT) * ceil_div(p.a.Cg, p.CGT);
  constexpr int target3 = 256;                        // (192 / 128: +1 % / +5 % step time, round 3)
  constexpr int target21 = 512;                       // 2x2 / 1x1 windows: several workgroups per CU
  const int target = p.a.KH == 3 ? target3 : target21;
  int64_t z = ceil_div(target, tiles);
  if (z > p.npatch) z = p.npatch;
  if (z < 1) z = 1;
  // equalise: every slice walks the same number of patches
  const int64_t pps = ceil_div(p.npatch, z);
  z = ceil_div(p.npatch, pps);
  return (int)z;
}

int wgradp_suggest_splits(const dfl_wgrad_args* a) {
  WgP p;
  int rc = wgp_plan(a, &p, false);
  if (rc != DFL_OK) return rc;
  return wgp_slices(p);
}

int wgradp_config(const dfl_wgrad_args* a) {
  WgP p;
  int rc = wgp_plan(a, &p, false);
  if (rc != DFL_OK) return rc;
  return 16 + (p.T == 9 ? 0 : (p.T == 4 ? 1 : 2));
}

template <int KH, int KW>
static int wgp_launch_t(const WgP& p_in, hipStream_t s) {
  WgP p = p_in;
  dim3 grid((unsigned)ceil_div(p.a.Cm, p.CMT), (unsigned)ceil_div(p.a.Cg, p.CGT), (unsigned)p.zslices);
  p.xcd_map = (grid.x * grid.y > 1 && grid.x * grid.y * grid.z >= 16) ? 1 : 0;   // (against the hardware's order: 1.330 -> 1.315 ms of weight gradients per step, round 4)
  const size_t lds = (size_t)p.lds_bytes;
#define DFL_WGP_LAUNCH(AFF_, DBRB_, BIAS_)                                                                                       \
  {                                                                                                                              \
    auto k = wgradp_kernel<KH, KW, AFF_, DBRB_, BIAS_>;                                                                          \
    DFL_LDS_OPT_IN(k, 150 * 1024, "dfl_conv2d_wgrad (bf16)") \
    hipLaunchKernelGGL(k, grid, dim3(256 * KH), lds, s, p);                                                                      \
  }
  const bool aff = p.a.in_scale != nullptr;
  if constexpr (KH == 3) {
    if (p.a.d_mode != 0) {
      if (aff) DFL_WGP_LAUNCH(true, true, true) else DFL_WGP_LAUNCH(false, true, true)
      return check_launch("dfl_conv2d_wgrad (bf16)");
    }
    if (p.a.bias_partial != nullptr) {               // the operand is a tensor of its own (dfl_conv_args.x_out)
      if (aff) DFL_WGP_LAUNCH(true, false, true) else DFL_WGP_LAUNCH(false, false, true)
      return check_launch("dfl_conv2d_wgrad (bf16)");
    }
  }
  if (aff) DFL_WGP_LAUNCH(true, false, false) else DFL_WGP_LAUNCH(false, false, false)
#undef DFL_WGP_LAUNCH
  return check_launch("dfl_conv2d_wgrad (bf16)");
}

int wgradp_launch(const dfl_wgrad_args* a, hipStream_t s) {
  WgP p;
  int rc = wgp_plan(a, &p, true);
  if (rc != DFL_OK) return rc;
  p.zslices = a->splits;                     // partial slots = pixel slices
  p.patches_per_slice = (int)ceil_div(p.npatch, p.zslices);
  if (p.zslices == 1 && p.lds_bytes < p.CMT * p.CGT * p.T * 4) p.lds_bytes = p.CMT * p.CGT * p.T * 4;   // the output tile of the torch-order store
#ifdef DFL_WGP_TRACE
  {
    const char* e = getenv("DFL_WGP_TRACE_PTR");
    p.trace = e ? reinterpret_cast<long long*>(strtoull(e, nullptr, 0)) : nullptr;
    fprintf(stderr, "wgradp: tile %dx%d pairs %d phases %d patch %dx%dx%d (P16 %d) gathered %dx%d slices %d patches/slice %d lds %d\n", p.CMT, p.CGT, p.pairs,
            p.phases, p.IPP, p.PH, p.PW, p.P16, p.IH, p.IW, p.zslices, p.patches_per_slice, p.lds_bytes);
  }
#endif
  switch (p.T) {
    case 9: return wgp_launch_t<3, 3>(p, s);
    case 4: return wgp_launch_t<2, 2>(p, s);
    default: return wgp_launch_t<1, 1>(p, s);
  }
}

}  // namespace dfl
