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
//
// PAIR form (dfl_conv2d_pair): the last 3x3 convolution of a residual block and the block's 1x1 convolution -- y1 = ReLU(conv(x) +
// bias), y2 = conv1x1(x3) + bias3 + BN(y1) (unet.py:218-231) -- as ONE launch: the waves of a tile take the k-steps of the second
// product behind their share of the first (a second accumulator tile), the epilogue rounds y1 to bf16, stores it and forms y2 from
// the rounded value exactly as the two launches do.  11 of the 44 launches of a forward.
#include <stdlib.h>
#include <string.h>

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

constexpr int SU = 9;              // k-steps per register set
constexpr int SU2 = 4;             // k-steps per wave of a pair's second product (registers of their own)
// Waves per workgroup, a template parameter: 8 (512 threads, one workgroup per CU) where eight waves share a tile's k-steps, 4 (256
// threads, two workgroups per CU) otherwise -- a layer of 1152 wave tasks then is 288 workgroups over all 256 CUs instead of 144 on 144
// (the fragments of this form are not reused across waves: a CU's L1 fill is what a 36-k-step layer waits for).
constexpr int SCONST_PER_WAVE = 6 * 32;            // per wave: bias, add_scale, add_shift, bias3, out_scale, out_shift of its tile's 32 columns

// second convolution of a pair (1x1, stride 1, no affine on load): y2 = x3 * w3 + bias3 + add_scale * y1 + add_shift
struct ConvPair {
  const void* x3;
  const void* w3;
  const float* bias3;
  const float* add_scale;
  const float* add_shift;
  void* y2;
  int ldx3, ldy2, ksteps2, kper2;
  int x3_fp32;                     // the network's first block: x3 is the 1-channel fp32 image, w3 the fp32 quad-packed weights
  uint32_t x3_bytes, w3_bytes;
};

__global__ void convs_pair_finish_kernel(const ConvP p, const ConvPair q);       // (K-sliced pairs, below)

template <bool AFF, bool PAIR, int SWAVES>
__global__ void __launch_bounds__(64 * SWAVES, SWAVES == 8 ? 1 : 2) convs_kernel(const ConvP p, const ConvPair q) {
  constexpr int NACC = PAIR ? 2 : 1;
  constexpr int SCONST_FLOATS = SWAVES * SCONST_PER_WAVE;
  constexpr int TABQ = 1024 / (64 * SWAVES);            // table entries per thread (Cin <= 1024)
  constexpr int SRED = SWAVES * NACC * 16 * 64;          // floats: partial tiles of the workgroup's waves
  extern __shared__ __attribute__((aligned(16))) float sm[];      // [8][NACC*16][64] partial tiles, [8][6][32] constants, [2][Cin] scale / shift
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
  __amdgpu_buffer_rsrc_t rsX3 = rsX, rsW3 = rsW;
  if constexpr (PAIR) {
    rsX3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.x3), 0, (int)q.x3_bytes, 0x00020000);
    rsW3 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(q.w3), 0, (int)q.w3_bytes, 0x00020000);
  }

  // k-steps of this wave: slice (blockIdx.y, kw) of the layer's T * Cin / 16 steps, s = tap * cpk + chunk, walked as v = 0 .. n1 - 1.
  // The second product of a PAIR has at most SU2 k-steps per wave (host): their fragments get registers of their own, are requested
  // with the first ones and used last -- no branch per step in the main loop
  const int cpk_sh = p.s_cpk_shift, cpk = 1 << cpk_sh;
  const int KW = a.KW, kw_magic = (256 + KW - 1) / KW;
  const int slice = (int)blockIdx.y * ksplit + kw;
  const int s_begin = slice * p.s_kper;
  const int n1 = max(0, min(p.s_ksteps, s_begin + p.s_kper) - s_begin);
  const int t_begin = PAIR ? slice * q.kper2 : 0;       // (the second product's k-steps are shared out over all K slices and waves as well)
  const int n2 = PAIR ? max(0, min(q.ksteps2, t_begin + q.kper2) - t_begin) : 0;
  const int nv = n1;
  const uint32_t x3off = PAIR ? (uint32_t)pix * (uint32_t)q.ldx3 * 2u + lh16 : 0u;

  su32x4 xb[2][SU], wb[2][SU];
  uint32_t okm[2] = {0u, 0u};
  auto load_group = [&](int buf, int v0) {
    uint32_t m = 0;
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      const int v = v0 + u;                              // (wave-uniform)
      const int s = s_begin + v;
      const bool live = v < n1;
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
  su32x4 x2b[PAIR ? SU2 : 1], w2b[PAIR ? SU2 : 1];

  f32x16 acc, acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f, acc2[r] = 0.f;
  float* cst = sm + SRED + wave * 192;                   // this wave's [6][32] epilogue constants
  float* tab = sm + SRED + SCONST_FLOATS;                // [2][Cin]
  auto compute_group = [&](int buf, int v0) {
#pragma unroll
    for (int u = 0; u < SU; ++u) {
      su32x4 x = xb[buf][u];
      if constexpr (AFF) {                               // BatchNorm affine of the input; zero padding applies AFTER it
        const int cc = (s_begin + v0 + u) & (cpk - 1);
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

  // ---- what the epilogue needs from memory is requested FIRST (loads return in order: the LDS copies below then wait for these
  //      few loads only, the fragments behind them land meanwhile): per wave the constants of its tile's 32 columns (lane l < 32:
  //      column ni*32 + l) and, AFF, the scale / shift table of the input channels
  const bool sliced = p.splits > 1;
  const bool scat = a.scatter2x2 != 0;
  const unsigned short* addp = reinterpret_cast<const unsigned short*>(a.add);
  const float* asc_p = PAIR ? q.add_scale : a.add_scale;
  const float* ash_p = PAIR ? q.add_shift : a.add_shift;
  float k0 = 0.f, k1 = 1.f, k2 = 0.f, k3 = 0.f, k4 = 1.f, k5 = 0.f;
  {
    const int c = ni * 32 + li;
    const bool on = tok && c < a.Ntot && lh == 0;
    const int cco = scat ? c % p.Cout : c;
    if (on && a.bias != nullptr) k0 = a.bias[cco];
    if (on && asc_p != nullptr) k1 = asc_p[c], k2 = ash_p[c];
    if (PAIR && on && q.bias3 != nullptr) k3 = q.bias3[c];
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
  if (SU < nv) load_group(1, SU);
  if constexpr (PAIR) {
#pragma unroll
    for (int u = 0; u < SU2; ++u) {
      const int t = t_begin + u;
      const bool live = u < n2;
      x2b[u] = __builtin_amdgcn_raw_buffer_load_b128(rsX3, (pok && live) ? x3off + (uint32_t)(t * 32) : SOOB, 0, 0);
      w2b[u] = __builtin_amdgcn_raw_buffer_load_b128(rsW3, (live && nok) ? (uint32_t)t * wrow + wcol : SOOB, 0, 0);
    }
  }
  // (which wave finishes which accumulator group -- channels ni*32 + 8 g + 4 lh + 0..3 of the lane's pixel: one wave of the tile ->
  //  all four; two -> g = kw, kw + 2; four -> g = kw; eight -> the even wave 2g finishes group g, see below)
  auto owns = [&](int g) { return ksplit == 8 ? ((kw >> 1) == g && (kw & 1) == 0) : ((g & (ksplit - 1)) == kw); };
  su32x2 addv[4];
  float x3v = 0.f;
  float4 w3v[4];
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int c = ni * 32 + 8 * g + 4 * lh;
    const bool on = !sliced && owns(g) && pok && c < a.Ntot;
    addv[g] = (!PAIR && on && addp != nullptr) ? *reinterpret_cast<const su32x2*>(addp + ((uint32_t)pix * (uint32_t)a.ldadd + (uint32_t)c)) : (su32x2){0u, 0u};
    w3v[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (PAIR && q.x3_fp32 && on) {                       // fp32 quad-packed weights of a 1-channel 1x1 window: w[n][0]
      const float* wq = reinterpret_cast<const float*>(q.w3) + (int64_t)c * 4;
      w3v[g] = make_float4(wq[0], wq[4], wq[8], wq[12]);
    }
  }
  if (PAIR && q.x3_fp32 && pok) x3v = reinterpret_cast<const float*>(q.x3)[(int64_t)pix * q.ldx3];
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

  // ---- k-steps
  {
    int v0 = 0;
    while (true) {
      compute_group(0, v0);
      v0 += SU;
      if (v0 >= nv) break;
      if (v0 + SU < nv) load_group(0, v0 + SU);
      compute_group(1, v0);
      v0 += SU;
      if (v0 >= nv) break;
      if (v0 + SU < nv) load_group(1, v0 + SU);
    }
  }
  if constexpr (PAIR) {
#pragma unroll
    for (int u = 0; u < SU2; ++u)
      acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w2b[u]), __builtin_bit_cast(bf16x8_t, x2b[u]), acc2, 0, 0, 0);
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
  unsigned short* yp = reinterpret_cast<unsigned short*>(a.y);
  auto finish_group = [&](int g, float4 v, float4 v2) {
    const int c = ni * 32 + 8 * g + 4 * lh;
    if (!pok || c >= a.Ntot) return;
    if (sliced) {                                        // raw sums of this K slice: convp_finish_kernel / convs_pair_finish_kernel do the rest
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
    if (!PAIR && addp != nullptr) {
      v.x += fmaf(sbf_lo(addv[g].x), cs.x, ch.x);
      v.y += fmaf(sbf_hi(addv[g].x), cs.y, ch.y);
      v.z += fmaf(sbf_lo(addv[g].y), cs.z, ch.z);
      v.w += fmaf(sbf_hi(addv[g].y), cs.w, ch.w);
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
    if (!PAIR && a.out_scale != nullptr) {               // the consumer's BatchNorm, applied to the ROUNDED value and rounded as the consumer would
      const float4 os = *reinterpret_cast<const float4*>(cst + 128 + 8 * g + 4 * lh);
      const float4 oh = *reinterpret_cast<const float4*>(cst + 160 + 8 * g + 4 * lh);
      const su32x2 t = w;
      w.x = spack_bf2(fmaf(sbf_lo(t.x), os.x, oh.x), fmaf(sbf_hi(t.x), os.y, oh.y));
      w.y = spack_bf2(fmaf(sbf_lo(t.y), os.z, oh.z), fmaf(sbf_hi(t.y), os.w, oh.w));
    }
    *reinterpret_cast<su32x2*>(dst) = w;
    if constexpr (PAIR) {
      // the second convolution's epilogue, as its own launch performs it: product + bias3, then + BN(y1) of the ROUNDED y1
      const float4 c3 = *reinterpret_cast<const float4*>(cst + 96 + 8 * g + 4 * lh);
      float4 o;
      if (q.x3_fp32) {                                   // (direct_conv_kernel<1,1,1>: acc = bias; acc = fma(x, w, acc))
        o.x = fmaf(x3v, w3v[g].x, c3.x); o.y = fmaf(x3v, w3v[g].y, c3.y); o.z = fmaf(x3v, w3v[g].z, c3.z); o.w = fmaf(x3v, w3v[g].w, c3.w);
      } else {
        o.x = v2.x + c3.x; o.y = v2.y + c3.y; o.z = v2.z + c3.z; o.w = v2.w + c3.w;
      }
      o.x += fmaf(sbf_lo(w.x), cs.x, ch.x);
      o.y += fmaf(sbf_hi(w.x), cs.y, ch.y);
      o.z += fmaf(sbf_lo(w.y), cs.z, ch.z);
      o.w += fmaf(sbf_hi(w.y), cs.w, ch.w);
      su32x2 w2;
      w2.x = spack_bf2(o.x, o.y);
      w2.y = spack_bf2(o.z, o.w);
      *reinterpret_cast<su32x2*>(reinterpret_cast<unsigned short*>(q.y2) + ((uint32_t)pix * (uint32_t)q.ldy2 + (uint32_t)c)) = w2;
    }
  };
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ksplit == 1) {
#pragma unroll
    for (int g = 0; g < 4; ++g)
      finish_group(g, make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]),
                   PAIR ? make_float4(acc2[4 * g], acc2[4 * g + 1], acc2[4 * g + 2], acc2[4 * g + 3]) : zero4);
  } else if (ksplit <= 4) {
    // wave kw of the tile's ksplit waves takes the groups g = kw, kw + ksplit, ...
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
    // eight waves: the even ones finish one group each (g = kw / 2); the sums are formed by ALL lanes of waves 2g and 2g + 1 --
    // wave 2g adds the partial tiles 0..3, wave 2g + 1 the tiles 4..7 -- and meet in LDS once more
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
    __syncthreads();                                     // every partial tile has been read
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
      for (int gg = 0; gg < 4; ++gg)                     // (static register indices: a run-time g would put per-group registers in scratch)
        if (gg == g) finish_group(gg, make_float4(v[0], v[1], v[2], v[3]), make_float4(v2[0], v2[1], v2[2], v2[3]));
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
  if (a.out_scale != nullptr && a.accumulate) return false;
  const int cpk = a.Cin / 16;
  if (cpk < 1 || (cpk & (cpk - 1)) != 0 || a.Cin > 1024) return false;
  if (a.KW > 16 || a.Ntot % 8 != 0 || p.Mtot > (1 << 16)) return false;
  if (p.Cout % 4 != 0) return false;
  // a latency problem, not a throughput problem: the largest layer of a 192x192 image is 1.4 GFLOP (operands are not reused
  // across tiles here: beyond this the patch-resident kernels win)
  if (2.0 * p.Mtot * a.Ntot * (double)(p.T * a.Cin) > 1.5e9) return false;
  // (round 5, measured with tools/kbench_infer.py 192 1 and docs/experiments/infer192_r05: with 8 k-steps per register set and the
  //  BatchNorm affine on load the 3x3 layers of the two widest levels were 2 - 3 us slower in this form; with 9 k-steps and the affine
  //  moved into the producer every layer of a 192x192 forward is faster here: 0.335 -> 0.320 ms without an exception)
  return true;
}

static int convs_waves(int ksplit_shift) { return ksplit_shift == 3 ? 8 : 4; }      // (always eight waves: 0.320 instead of 0.3125 ms per forward)

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
  // K slices over workgroups (a finish launch, 4.7 us) only where the eight waves of a tile would be left with more than two
  // register sets of k-steps each
  int zs = 1;
  if (force_splits > 0) zs = force_splits;
  else if (want > 8 && p->s_ksteps > 8 * 2 * SU) zs = want / 8;
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
  const int W = convs_waves(ksh);
  p->grid = (int)ceil_div(tiles, W >> ksh);
  p->lds_bytes = (W * 16 * 64 + W * SCONST_PER_WAVE + 2 * a.Cin) * 4;
}

int convs_launch(const ConvP& p, hipStream_t s) {
  dim3 grid((unsigned)p.grid, (unsigned)p.splits);
  const size_t lds = (size_t)p.lds_bytes;
  ConvPair none;
  memset(&none, 0, sizeof(none));
  const int W = convs_waves(p.s_ksplit_shift);
  if (p.a.in_scale != nullptr) {
    if (W == 8) hipLaunchKernelGGL((convs_kernel<true, false, 8>), grid, dim3(512), lds, s, p, none);
    else hipLaunchKernelGGL((convs_kernel<true, false, 4>), grid, dim3(256), lds, s, p, none);
  } else {
    if (W == 8) hipLaunchKernelGGL((convs_kernel<false, false, 8>), grid, dim3(512), lds, s, p, none);
    else hipLaunchKernelGGL((convs_kernel<false, false, 4>), grid, dim3(256), lds, s, p, none);
  }
  return check_launch("dfl_conv2d (bf16, latency form)");
}

// ---- pairs (dfl_conv2d_pair): conditions under which (a, b) run as one launch; pa = a's plan in latency form
static bool pair_ok(const dfl_conv_args* a, const dfl_conv_args* b, const ConvP& pa) {
  if (pa.tile != CONVS_TILE) return false;
  if (pa.splits > 1 && (!b->x_bf16 || a->partial == nullptr)) return false;
  if (a->scatter2x2 || a->accumulate || a->add != nullptr || a->out_scale != nullptr || a->in_scale != nullptr) return false;
  if (b->x_bf16 && ceil_div(b->Cin / 16, (int64_t)pa.splits << pa.s_ksplit_shift) > SU2) return false;
  if (!b->latency_form || !b->y_bf16 || b->KH != 1 || b->KW != 1 || b->stride != 1 || b->pad != 0 || b->scatter2x2 || b->accumulate || b->relu) return false;
  if (b->in_scale != nullptr || b->in_tot != nullptr || b->add_tot != nullptr || b->x_mode != 0 || b->x_out != nullptr || b->out_scale != nullptr) return false;
  if (b->stat_partials != nullptr || b->stat_totals != nullptr || b->stat_other != nullptr || b->splits > 1) return false;
  if (b->add != a->y || b->ldadd != a->ldy || b->add_scale == nullptr || b->add_shift == nullptr) return false;
  if (b->N != a->N || b->Hout != a->Hout || b->Wout != a->Wout || b->Hin != a->Hout || b->Win != a->Wout || b->Ntot != a->Ntot) return false;
  if (b->y == a->y || b->y == nullptr || b->ldy % 4 != 0 || b->x == nullptr || b->w == nullptr) return false;
  if (b->x_bf16) {
    if (b->w_split != 2 || b->Cin % 16 != 0 || b->ldx % 8 != 0 || !aligned16(b->x) || !aligned16(b->w)) return false;
    const int64_t xb = (((int64_t)b->N * b->Hin * b->Win - 1) * b->ldx + b->Cin) * 2, wb = (int64_t)b->Cin * b->Ntot * 2;
    if (xb >= (1ll << 31) - 4096 || wb >= (1ll << 31) - 4096) return false;
  } else {
    if (b->Cin != 1 || b->w_split != 0 || b->x_split != 0) return false;          // the network's first block: 1-channel fp32 image
  }
  return true;
}

int convs_pair_ok(const dfl_conv_args* a, const dfl_conv_args* b) {
  if (a == nullptr || b == nullptr || !a->x_bf16 || !a->latency_form) return 0;
  ConvP pa;
  if (convp_plan(a, &pa, a->splits > 1 ? a->splits : 1) != DFL_OK) return 0;
  if (!pair_ok(a, b, pa)) return 0;
  return pa.splits > 1 ? 2 : 1;            // 2: K slices -- a->partial holds [2][splits][M][Ntot] floats
}

// a then b as ONE launch (the caller has checked convs_pair_ok)
int convs_pair_launch(const dfl_conv_args* a, const dfl_conv_args* b, hipStream_t s) {
  ConvP p;
  int rc = convp_plan(a, &p, a->splits > 1 ? a->splits : 1);
  if (rc != DFL_OK) return rc;
  DFL_REQUIRE(pair_ok(a, b, p), "dfl_conv2d_pair: these two convolutions do not form a pair");
  ConvPair q;
  memset(&q, 0, sizeof(q));
  q.x3 = b->x;
  q.w3 = b->w;
  q.bias3 = b->bias;
  q.add_scale = b->add_scale;
  q.add_shift = b->add_shift;
  q.y2 = b->y;
  q.ldx3 = b->ldx;
  q.ldy2 = b->ldy;
  q.x3_fp32 = b->x_bf16 ? 0 : 1;
  if (b->x_bf16) {
    q.ksteps2 = b->Cin / 16;
    q.kper2 = (int)ceil_div(q.ksteps2, (int64_t)p.splits << p.s_ksplit_shift);
    q.x3_bytes = (uint32_t)((((int64_t)b->N * b->Hin * b->Win - 1) * b->ldx + b->Cin) * 2);
    q.w3_bytes = (uint32_t)((int64_t)b->Cin * b->Ntot * 2);
  }
  dim3 grid((unsigned)p.grid, (unsigned)p.splits);
  const int W = convs_waves(p.s_ksplit_shift);
  const size_t lds = (size_t)(W * 2 * 16 * 64 + W * SCONST_PER_WAVE + 2 * a->Cin) * 4;
  if (W == 8) {
    auto k = convs_kernel<false, true, 8>;
    DFL_LDS_OPT_IN(k, 96 * 1024, "dfl_conv2d (bf16, latency form)")
    hipLaunchKernelGGL(k, grid, dim3(512), lds, s, p, q);
  } else {
    hipLaunchKernelGGL((convs_kernel<false, true, 4>), grid, dim3(256), lds, s, p, q);
  }
  if (p.splits > 1) {
    rc = check_launch("dfl_conv2d_pair");
    if (rc != DFL_OK) return rc;
    hipLaunchKernelGGL(convs_pair_finish_kernel, dim3((unsigned)ceil_div((int64_t)p.Mtot * (a->Ntot / 4), 256)), dim3(256), 0, s, p, q);
  }
  return check_launch("dfl_conv2d_pair");
}


// K-sliced pairs: y1 = ReLU(sum of the first product's slices + bias), y2 = sum of the second product's slices + bias3 + BN(y1) -- the pair
// epilogue of convs_kernel on the sums (a.partial: [2][splits][M][Ntot] fp32).  One thread: 4 consecutive channels of a pixel.
__global__ void __launch_bounds__(256) convs_pair_finish_kernel(const ConvP p, const ConvPair q) {
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
  su32x2 w;
  w.x = spack_bf2(v.x, v.y);
  w.y = spack_bf2(v.z, v.w);
  *reinterpret_cast<su32x2*>(reinterpret_cast<unsigned short*>(a.y) + ((uint32_t)pix * (uint32_t)a.ldy + (uint32_t)c)) = w;
  if (q.bias3 != nullptr) { o.x += q.bias3[c]; o.y += q.bias3[c + 1]; o.z += q.bias3[c + 2]; o.w += q.bias3[c + 3]; }
  o.x += fmaf(sbf_lo(w.x), q.add_scale[c], q.add_shift[c]);
  o.y += fmaf(sbf_hi(w.x), q.add_scale[c + 1], q.add_shift[c + 1]);
  o.z += fmaf(sbf_lo(w.y), q.add_scale[c + 2], q.add_shift[c + 2]);
  o.w += fmaf(sbf_hi(w.y), q.add_scale[c + 3], q.add_shift[c + 3]);
  su32x2 w2;
  w2.x = spack_bf2(o.x, o.y);
  w2.y = spack_bf2(o.z, o.w);
  *reinterpret_cast<su32x2*>(reinterpret_cast<unsigned short*>(q.y2) + ((uint32_t)pix * (uint32_t)q.ldy2 + (uint32_t)c)) = w2;
}

// ---- the network's first convolution in the same spirit (1-channel fp32 image, 3x3 window, stride 1: unet.py:211 with in_channels = 1):
// direct_conv3_rows_kernel walks bands of rows, three per workgroup -- 32 workgroups for one 192x192 image, 13 us.  Here a thread owns
// one pixel and 8 channels (a wave: 64 consecutive pixels of one channel group): 9 loads of the image, the 9 x 8 weights from LDS,
// the multiply-adds in direct_conv3_rows_kernel's order (bit-identical results), one 16-byte store.
template <bool BF>
__global__ void __launch_bounds__(256) convs_first_kernel(const dfl_conv_args a, int M) {
  __shared__ __attribute__((aligned(16))) float wl[9 * 64 + 3 * 64];          // [9][Ntot] weights, bias, out_scale, out_shift
  const int ncg = a.Ntot >> 3;                            // channel groups of 8
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wpb = 4;                                      // waves per block
  const int gw = (int)blockIdx.x * wpb + wave;            // global wave: (pixel block of 64, channel group)
  const int cg = gw % ncg, pb = gw / ncg;
  const int pix = pb * 64 + lane;
  const bool pok = pix < M;
  const int HW = a.Hout * a.Wout;
  const int img = pix / HW, rem = pix - img * HW;
  const int oy = rem / a.Wout, ox = rem - oy * a.Wout;
  float xv[9];
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int iy = oy - a.pad + dy, ix = ox - a.pad + dx;
      const bool ok = pok && (unsigned)iy < (unsigned)a.Hin && (unsigned)ix < (unsigned)a.Win;
      xv[dy * 3 + dx] = ok ? a.x[((int64_t)(img * a.Hin + iy) * a.Win + ix) * a.ldx] : 0.f;
    }
  const int Nt = a.Ntot;
  for (int i = threadIdx.x; i < 9 * Nt; i += 256) {
    const int k = i / Nt, nn = i - k * Nt;
    wl[i] = a.w[((int64_t)(k >> 2) * Nt + nn) * 4 + (k & 3)];                 // quad-packed operand
  }
  for (int i = threadIdx.x; i < Nt; i += 256) {
    wl[9 * Nt + i] = a.bias != nullptr ? a.bias[i] : 0.f;
    wl[10 * Nt + i] = a.out_scale != nullptr ? a.out_scale[i] : 1.f;
    wl[11 * Nt + i] = a.out_scale != nullptr ? a.out_shift[i] : 0.f;
  }
  __syncthreads();
  const int n0 = cg * 8;
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = wl[9 * Nt + n0 + j];
#pragma unroll
  for (int k = 0; k < 9; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = fmaf(xv[k], wl[k * Nt + n0 + j], acc[j]);
  if (a.relu) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = fmaxf(acc[j], 0.f);
  }
  if constexpr (!BF) {                                    // fp32 tensors: nothing is rounded, the output affine is the consumer's fma
    if (a.out_scale != nullptr) {
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = fmaf(acc[j], wl[10 * Nt + n0 + j], wl[11 * Nt + n0 + j]);
    }
    if (pok) {
      float* dst = a.y + ((int64_t)pix * a.ldy + n0);
      *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
      *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
    return;
  }
  su32x4 w;
  w.x = spack_bf2(acc[0], acc[1]);
  w.y = spack_bf2(acc[2], acc[3]);
  w.z = spack_bf2(acc[4], acc[5]);
  w.w = spack_bf2(acc[6], acc[7]);
  if (a.out_scale != nullptr) {
    const float* os = wl + 10 * Nt + n0;
    const float* oh = wl + 11 * Nt + n0;
    const su32x4 t = w;
    w.x = spack_bf2(fmaf(sbf_lo(t.x), os[0], oh[0]), fmaf(sbf_hi(t.x), os[1], oh[1]));
    w.y = spack_bf2(fmaf(sbf_lo(t.y), os[2], oh[2]), fmaf(sbf_hi(t.y), os[3], oh[3]));
    w.z = spack_bf2(fmaf(sbf_lo(t.z), os[4], oh[4]), fmaf(sbf_hi(t.z), os[5], oh[5]));
    w.w = spack_bf2(fmaf(sbf_lo(t.w), os[6], oh[6]), fmaf(sbf_hi(t.w), os[7], oh[7]));
  }
  if (pok) *reinterpret_cast<su32x4*>(reinterpret_cast<unsigned short*>(a.y) + ((int64_t)pix * a.ldy + n0)) = w;
}

bool convs_first_ok(const dfl_conv_args* a) {
  if (!a->latency_form || !convs_switch()) return false;
  if (a->x_bf16 || a->Cin != 1 || a->KH != 3 || a->KW != 3 || a->stride != 1 || a->w_split != 0 || a->x_split != 0) return false;
  if (a->Ntot % 8 != 0 || a->Ntot > 64 || a->ldy % 8 != 0 || !aligned16(a->y)) return false;
  if (a->add != nullptr || a->accumulate || a->scatter2x2 || a->splits > 1 || a->in_scale != nullptr || a->in_tot != nullptr) return false;
  if (a->stat_partials != nullptr || a->stat_totals != nullptr || a->stat_other != nullptr || a->x_mode != 0) return false;
  const int ho = a->Hin + 2 * a->pad - 2, wo = a->Win + 2 * a->pad - 2;
  if (ho != a->Hout || wo != a->Wout) return false;
  return (int64_t)a->N * a->Hout * a->Wout <= (1 << 16);
}

int convs_first_launch(const dfl_conv_args* a, hipStream_t s) {
  const int M = a->N * a->Hout * a->Wout;
  const int waves = (int)ceil_div(M, 64) * (a->Ntot / 8);
  if (a->y_bf16) hipLaunchKernelGGL(convs_first_kernel<true>, dim3((unsigned)ceil_div(waves, 4)), dim3(256), 0, s, *a, M);
  else hipLaunchKernelGGL(convs_first_kernel<false>, dim3((unsigned)ceil_div(waves, 4)), dim3(256), 0, s, *a, M);
  return check_launch("dfl_conv2d (first layer, latency form)");
}

}  // namespace dfl
