// Latency form of the bf16 convolution (round 5): the kernel dfl_conv2d takes for the SMALL problems of a batch-1 inference
// forward (dfl_conv_args.latency_form; reference: the per-image loops of train_test_code/util.py:116-165 and :318-356 call
// UNet.forward, unet.py:161-193, on one image at a time).
//
// Why a second kernel for the same contract.  At batch 1 and 192x192 every layer of the paper network is 0.01 - 1 GFLOP: the
// forward is a chain of 44 dependent convolutions whose time is not throughput but the LENGTH OF EACH KERNEL'S DEPENDENCY
// CHAIN.  The patch-resident kernel (convp_bf16.hip) is built for throughput -- geometry tables, patch image staged through
// LDS (global -> registers -> LDS -> barrier), weight ring, accumulators -> LDS -> rows -> stores -- and takes 5 - 15 us per
// launch on these shapes where a dependent launch boundary costs 1.5 (docs/experiments/infer192_r05: 58 kernels, 0.48 ms;
// a grid barrier inside one persistent launch costs 4.2 us without and 12 us with agent-scope fences on this chip, so one
// launch for the whole forward is no way out either).  This kernel keeps the chain as short as the hardware allows:
//   * no LDS staging, no barrier in front of the matrix instructions: a wave owns ONE 32-pixel x 32-channel output tile and
//     a range of k-steps; both MFMA operands come straight from global memory (L2) in the fragment layout -- the pixel
//     fragment is the lane's own 16 bytes (8 channels) of its pixel at the tap's offset (zero padding = out-of-range buffer
//     offset), the weight fragment is one coalesced 1 KiB run of the [k/16][n][16] layout convp uses;
//   * all loads of up to 16 k-steps are in flight before the first matrix instruction (two register sets of 8 k-steps);
//   * D = W * X orientation: a lane ends up with 4 consecutive channels of ITS pixel per accumulator group, so the epilogue
//     (bias, ReLU, + BN(add), 2x2 scatter) runs on the accumulators and stores 8 bytes per group -- no transpose;
//   * K is split over the 8 waves of a workgroup (partial tiles meet in LDS, summed in a fixed order) so that a layer with
//     18 ... 576 k-steps still is a few k-steps deep per wave, and over workgroups (fp32 partial slices + convp_finish_kernel)
//     only for the weight-heavy levels whose 5 - 19 MB of weights need every CU's memory pipe.
// Same arithmetic as convp_kernel: bf16 products, fp32 accumulation (another summation order), values rounded to bf16 once.
#include "common.h"
#include "convp.h"

namespace dfl {

constexpr uint32_t SOOB = 0x80000000u;
typedef unsigned int su32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int su32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float sbf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float sbf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t spack_bf2(float a, float b) {
  const bf16x2_t h = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);
  return __builtin_bit_cast(uint32_t, h);
}

constexpr int SU = 8;              // k-steps per register set
constexpr int SWAVES = 8;          // waves per workgroup
constexpr int SRED_FLOATS = SWAVES * 16 * 64;

template <bool AFF>
__global__ void __launch_bounds__(64 * SWAVES, 1) convs_kernel(const ConvP p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];      // [8][16][64] partial tiles, then [2][Cin] scale / shift
  const dfl_conv_args& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int ksh = p.s_ksplit_shift, ksplit = 1 << ksh;
  const int grp = wave >> ksh, kw = wave & (ksplit - 1);
  const int tile = (int)blockIdx.x * (SWAVES >> ksh) + grp;
  const bool tok = tile < p.s_mt * p.s_nt;
  const int mi = tok ? tile / p.s_nt : 0, ni = tok ? tile - mi * p.s_nt : 0;

  // this lane's pixel of the gather grid and the input pixel of its tap (0, 0)
  const int pix = mi * 32 + li;
  const bool pok = tok && pix < p.Mtot;
  const int HW = p.Hg * p.Wg;
  const int img = pix / HW, rem = pix - img * HW;
  const int gy = rem / p.Wg, gx = rem - gy * p.Wg;
  const int iy0 = gy * a.stride - a.pad, ix0 = gx * a.stride - a.pad;
  const int pbase = (img * a.Hin + iy0) * a.Win + ix0;
  const uint32_t ldx2 = (uint32_t)a.ldx * 2u, lh16 = (uint32_t)lh * 16u;
  // this lane's weight column
  const int n = ni * 32 + li;
  const bool nok = tok && n < a.Ntot;
  const uint32_t wrow = (uint32_t)a.Ntot * 32u, wcol = (uint32_t)n * 32u + lh16;

  __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)p.w_bytes, 0x00020000);

  // k-steps of this wave: slice (blockIdx.y, kw) of the layer's T * Cin / 16 steps, s = tap * cpk + chunk
  const int cpk_sh = p.s_cpk_shift, cpk = 1 << cpk_sh;
  const int KW = a.KW, kw_magic = (256 + KW - 1) / KW;
  const int slice = (int)blockIdx.y * ksplit + kw;
  const int s_begin = slice * p.s_kper;
  const int s_end = min(p.s_ksteps, s_begin + p.s_kper);

  su32x4 xb[2][SU], wb[2][SU];
  uint32_t okm[2] = {0u, 0u};
  auto load_group = [&](int buf, int s0) {
    uint32_t m = 0;
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int s = s0 + u;                              // (wave-uniform)
      const bool live = s < s_end;
      const int tap = s >> cpk_sh, cc = s & (cpk - 1);
      const int ty = (tap * kw_magic) >> 8, tx = tap - ty * KW;
      const bool ok = pok && live && (unsigned)(iy0 + ty) < (unsigned)a.Hin && (unsigned)(ix0 + tx) < (unsigned)a.Win;
      const uint32_t xo = (uint32_t)(pbase + ty * a.Win + tx) * ldx2 + (uint32_t)(cc * 32) + lh16;
      xb[buf][u] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? xo : SOOB, 0, 0);
      wb[buf][u] = __builtin_amdgcn_raw_buffer_load_b128(rsW, (live && nok) ? (uint32_t)s * wrow + wcol : SOOB, 0, 0);
      m |= ok ? (1u << u) : 0u;
    }
    okm[buf] = m;
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float* tab = sm + SRED_FLOATS;                         // [2][Cin]
  auto compute_group = [&](int buf, int s0) {
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      su32x4 x = xb[buf][u];
      if constexpr (AFF) {                               // BatchNorm affine of the input; zero padding applies AFTER it
        const int cc = (s0 + u) & (cpk - 1);
        const float* sc = tab + cc * 16 + lh * 8;
        const float* sh = sc + a.Cin;
        const float4 s0v = *reinterpret_cast<const float4*>(sc), s1v = *reinterpret_cast<const float4*>(sc + 4);
        const float4 h0v = *reinterpret_cast<const float4*>(sh), h1v = *reinterpret_cast<const float4*>(sh + 4);
        su32x4 y;
        y.x = spack_bf2(fmaf(sbf_lo(x.x), s0v.x, h0v.x), fmaf(sbf_hi(x.x), s0v.y, h0v.y));
        y.y = spack_bf2(fmaf(sbf_lo(x.y), s0v.z, h0v.z), fmaf(sbf_hi(x.y), s0v.w, h0v.w));
        y.z = spack_bf2(fmaf(sbf_lo(x.z), s1v.x, h1v.x), fmaf(sbf_hi(x.z), s1v.y, h1v.y));
        y.w = spack_bf2(fmaf(sbf_lo(x.w), s1v.z, h1v.z), fmaf(sbf_hi(x.w), s1v.w, h1v.w));
        const bool ok = (okm[buf] >> u) & 1u;
        x.x = ok ? y.x : 0u; x.y = ok ? y.y : 0u; x.z = ok ? y.z : 0u; x.w = ok ? y.w : 0u;
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, wb[buf][u]), __builtin_bit_cast(bf16x8_t, x), acc, 0, 0, 0);
    }
  };

  // ---- everything the epilogue needs from memory is requested before the k-steps as well: this wave finishes the accumulator
  //      groups g = kw, kw + ksplit, ... (< 4) of its tile: channels ni*32 + 8 g + 4 lh + 0..3 of the lane's pixel
  const bool sliced = p.splits > 1;
  const bool scat = a.scatter2x2 != 0;
  float4 cbias[4], casc[4], cash[4];
  su32x2 addv[4];
  const unsigned short* addp = reinterpret_cast<const unsigned short*>(a.add);
  // (AFF) the scale / shift table is requested FIRST: loads return in order, so the wait in front of its LDS copy and barrier does
  // not wait for the fragments behind it -- they land while the table is written
  float tsc[2] = {1.f, 1.f}, tsh[2] = {0.f, 0.f};
  if constexpr (AFF) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c = tid + q * 64 * SWAVES;
      if (c < a.Cin) {
        tsc[q] = a.in_scale[c];
        tsh[q] = a.in_shift[c];
      }
    }
  }
  load_group(0, s_begin);
  if (s_begin + SU < s_end) load_group(1, s_begin + SU);
  // (which wave finishes which group: one wave of the tile -> all four; two -> g = kw, kw + 2; four -> g = kw; eight -> the even
  //  wave 2g finishes group g, see below)
  auto owns = [&](int g) { return ksplit == 8 ? ((kw >> 1) == g && (kw & 1) == 0) : ((g & (ksplit - 1)) == kw); };
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c = ni * 32 + 8 * g + 4 * lh;
    const bool on = !sliced && owns(g) && pok && c < a.Ntot;
    const int cco = scat ? c % p.Cout : c;
    cbias[g] = (on && a.bias != nullptr) ? *reinterpret_cast<const float4*>(a.bias + cco) : make_float4(0.f, 0.f, 0.f, 0.f);
    casc[g] = (on && addp != nullptr && a.add_scale != nullptr) ? *reinterpret_cast<const float4*>(a.add_scale + c) : make_float4(1.f, 1.f, 1.f, 1.f);
    cash[g] = (on && addp != nullptr && a.add_scale != nullptr) ? *reinterpret_cast<const float4*>(a.add_shift + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    addv[g] = (on && addp != nullptr) ? *reinterpret_cast<const su32x2*>(addp + ((uint32_t)pix * (uint32_t)a.ldadd + (uint32_t)c)) : (su32x2){0u, 0u};
  }
  if constexpr (AFF) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c = tid + q * 64 * SWAVES;
      if (c < a.Cin) {
        tab[c] = tsc[q];
        tab[a.Cin + c] = tsh[q];
      }
    }
    __syncthreads();
  }

  // ---- k-steps
  {
    int s0 = s_begin;
    while (true) {
      compute_group(0, s0);
      s0 += SU;
      if (s0 >= s_end) break;
      if (s0 + SU < s_end) load_group(0, s0 + SU);
      compute_group(1, s0);
      s0 += SU;
      if (s0 >= s_end) break;
      if (s0 + SU < s_end) load_group(1, s0 + SU);
    }
  }

  // ---- the waves of a tile add up through LDS (fixed order), each finishing its share of the accumulator groups
  if (ksplit > 1) {
    float* mine = sm + (wave * 16) * 64 + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[r * 64] = acc[r];
    __syncthreads();
  }
  unsigned short* yp = reinterpret_cast<unsigned short*>(a.y);
  auto finish_group = [&](int g, float4 v) {
    const int c = ni * 32 + 8 * g + 4 * lh;
    if (!pok || c >= a.Ntot) return;
    if (sliced) {                                        // raw sums of this K slice: convp_finish_kernel does the rest
      *reinterpret_cast<float4*>(a.partial + ((int64_t)blockIdx.y * p.Mtot + pix) * a.Ntot + c) = v;
      return;
    }
    v.x += cbias[g].x; v.y += cbias[g].y; v.z += cbias[g].z; v.w += cbias[g].w;
    if (a.relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    if (addp != nullptr) {
      v.x += fmaf(sbf_lo(addv[g].x), casc[g].x, cash[g].x);
      v.y += fmaf(sbf_hi(addv[g].x), casc[g].y, cash[g].y);
      v.z += fmaf(sbf_lo(addv[g].y), casc[g].z, cash[g].z);
      v.w += fmaf(sbf_hi(addv[g].y), casc[g].w, cash[g].w);
    }
    uint32_t opix = (uint32_t)pix;
    int ocol = c;
    if (scat) {
      const int ab = c / p.Cout;
      ocol = c - ab * p.Cout;
      opix = (uint32_t)((img * a.Hout + 2 * gy + (ab >> 1)) * a.Wout + 2 * gx + (ab & 1));
    }
    unsigned short* dst = yp + (opix * (uint32_t)a.ldy + (uint32_t)ocol);
    if (a.accumulate) {
      const su32x2 o = *reinterpret_cast<const su32x2*>(dst);
      v.x += sbf_lo(o.x); v.y += sbf_hi(o.x); v.z += sbf_lo(o.y); v.w += sbf_hi(o.y);
    }
    su32x2 w;
    w.x = spack_bf2(v.x, v.y);
    w.y = spack_bf2(v.z, v.w);
    *reinterpret_cast<su32x2*>(dst) = w;
  };
  if (ksplit == 1) {
#pragma unroll
    for (int g = 0; g < 4; ++g) finish_group(g, make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]));
  } else if (ksplit <= 4) {
    // wave kw of the tile's ksplit waves takes the groups g = kw, kw + ksplit, ...
    const float* base = sm + ((grp << ksh) * 16) * 64 + lane;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if ((g & (ksplit - 1)) != kw) continue;
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      for (int q = 0; q < ksplit; ++q)
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += base[(q * 16 + 4 * g + j) * 64];
      finish_group(g, make_float4(v[0], v[1], v[2], v[3]));
    }
  } else {
    // eight waves: the even ones finish one group each (g = kw / 2); the sums are formed by ALL lanes of waves 2g and 2g + 1 --
    // wave 2g adds the partial tiles 0..3, wave 2g + 1 the tiles 4..7 -- and meet in LDS once more
    const float* base = sm + lane;
    const int g = kw >> 1, half = kw & 1;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int q = 4 * half; q < 4 * half + 4; ++q)
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += base[(q * 16 + 4 * g + j) * 64];
    __syncthreads();                                     // every partial tile has been read
    if (half) {
#pragma unroll
      for (int j = 0; j < 4; ++j) sm[(g * 4 + j) * 64 + lane] = v[j];
    }
    __syncthreads();
    if (!half) {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] += sm[(g * 4 + j) * 64 + lane];
#pragma unroll
      for (int gg = 0; gg < 4; ++gg)                     // (static register indices: a run-time g would put the epilogue constants in scratch)
        if (gg == g) finish_group(gg, make_float4(v[0], v[1], v[2], v[3]));
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------

static bool convs_switch() {
  static const bool on = [] {
    const char* e = getenv("DFL_CONVS");               // 0: the latency form is never taken (A/B against the patch kernels)
    return e == nullptr || atoi(e) != 0;
  }();
  return on;
}

// Can these arguments take the latency form?  (the caller has validated them as convp_plan_search does)
bool convs_eligible(const dfl_conv_args& a, const ConvP& p) {
  if (!a.latency_form || !convs_switch()) return false;
  if (a.x_mode != 0 || a.x_out != nullptr || a.stat_partials != nullptr || a.stat_totals != nullptr || a.stat_other != nullptr) return false;
  if (a.in_tot != nullptr || a.add_tot != nullptr) return false;
  const int cpk = a.Cin / 16;
  if (cpk < 1 || (cpk & (cpk - 1)) != 0 || a.Cin > 1024) return false;
  if (a.KW > 16 || a.Ntot % 8 != 0 || p.Mtot > (1 << 16)) return false;
  if (p.Cout % 4 != 0) return false;
  // a latency problem, not a throughput problem: the largest layer of a 192x192 image is 1.4 GFLOP (operands are not reused
  // across tiles here: beyond this the patch-resident kernels win)
  if (2.0 * p.Mtot * a.Ntot * (double)(p.T * a.Cin) > 1.5e9) return false;
  // measured (tools/kbench_infer.py 192 1, both forms in one call): the 3x3 layers of the two widest levels with 64 or more input
  // channels -- 36 or more k-steps on 288+ pixel tiles, i.e. three dependent load rounds per wave here -- stay with the patch kernels
  // (14 -> 17, 13 -> 15, 15 -> 17 us with this form); everything else of a 192x192 forward is 1 - 7 us shorter
  if (p.Mtot >= 9216 && p.T * (a.Cin / 16) >= 36) return false;
  if (a.bias != nullptr && (reinterpret_cast<uintptr_t>(a.bias) & 15u) != 0) return false;
  if (a.add != nullptr && a.add_scale != nullptr && ((reinterpret_cast<uintptr_t>(a.add_scale) & 15u) != 0 || (reinterpret_cast<uintptr_t>(a.add_shift) & 15u) != 0)) return false;
  if (a.ldadd % 4 != 0 || a.ldy % 4 != 0) return false;
  return true;
}

// Work split: 32 x 32 tiles, `ksplit` waves of a workgroup per tile, `splits` K slices over workgroups.  Aim: about one wave
// per SIMD-slot pair of the chip (2048 waves) and at least 6 k-steps per wave; cross-workgroup slices only when the eight waves
// of a workgroup leave more than ~36 k-steps each.
void convs_plan(const dfl_conv_args& a, ConvP* p, int force_splits) {
  p->s_mt = (int)ceil_div(p->Mtot, 32);
  p->s_nt = (int)ceil_div(a.Ntot, 32);
  const int cpk = a.Cin / 16;
  int sh = 0;
  while ((1 << sh) < cpk) ++sh;
  p->s_cpk_shift = sh;
  p->s_ksteps = p->T * cpk;
  const int tiles = p->s_mt * p->s_nt;
  int want = 2048 / tiles;
  if (want > p->s_ksteps / 6) want = p->s_ksteps / 6;
  if (want < 1) want = 1;
  int zs = 1;
  if (force_splits > 0) zs = force_splits;
  else if (want > 8) zs = want / 8;
  if (zs > 16) zs = 16;
  if (zs > p->s_ksteps) zs = p->s_ksteps;
  int per = want / zs;
  int ksh = 0;
  while (ksh < 3 && (2 << ksh) <= per) ++ksh;
  p->s_ksplit_shift = ksh;
  p->s_kper = (int)ceil_div(p->s_ksteps, (int64_t)zs << ksh);
  if (force_splits <= 0) zs = (int)ceil_div(p->s_ksteps, (int64_t)p->s_kper << ksh);      // (no empty slices when the choice is free)
  p->splits = zs;
  p->tile = CONVS_TILE;
  p->grid = (int)ceil_div(tiles, SWAVES >> ksh);
  p->lds_bytes = SRED_FLOATS * 4 + 2 * a.Cin * 4;
}

int convs_launch(const ConvP& p, hipStream_t s) {
  dim3 grid((unsigned)p.grid, (unsigned)p.splits);
  const size_t lds = (size_t)p.lds_bytes;
  if (p.a.in_scale != nullptr) {
    hipLaunchKernelGGL(convs_kernel<true>, grid, dim3(64 * SWAVES), lds, s, p);
  } else {
    hipLaunchKernelGGL(convs_kernel<false>, grid, dim3(64 * SWAVES), lds, s, p);
  }
  return check_launch("dfl_conv2d (bf16, latency form)");
}

}  // namespace dfl
