// Patch-resident convolution for bf16 tensors (math mode 4, "bf16 storage": BASELINE configs[1] as named -- bf16
// activations and weights in HBM, fp32 accumulation / statistics / master weights).
//
// Serves what conv_gemm.hip / conv_rows.hip serve for fp32 tensors -- nn.Conv2d 3x3 / 1x1 / 2x2-stride-2,
// nn.ConvTranspose2d(k2,s2) (scatter epilogue) and, with re-packed weights, every data gradient (reference:
// train_test_code/unet.py:93,207,211,218,240; torch autograd at train.py:422) -- with a different decomposition, built
// around what bf16 changes: the matrix pipe is 16x faster than for fp32 products, so the loop must not touch LDS or the
// vector-load return path once per tap.
//
//   * A workgroup owns a PATCH of output pixels (PH x PW pixels of IPP images, WM*TM*32 GEMM rows) and BN = WN*TN*32 output
//     columns.  The input pixels the patch needs for ALL taps -- the patch with its halo -- are staged in LDS ONCE per
//     block of CK input channels (BatchNorm affine applied there, zero padding written as zeros), [pixel][CK] bf16 with a
//     pixel pitch of 2*CK + 16 bytes.  Every tap then is an address offset: lane l of a 32-row tile reads its pixel's 16
//     bytes (8 channels) with one ds_read_b128 at base(l) + tap offset + chunk offset.  The pitch (odd multiple of 16
//     bytes) spreads the 16 pixels of a b128 lane group over all 64 banks: no conflicts for consecutive pixels.  The loop
//     has NO barrier (the image is static while it runs) and no per-tap gather, masks or bounds logic at all.
//   * Weights never touch LDS: they are packed [k/16][n][16 k] bf16 (dfl_pack_job.split = 2), so the MFMA B fragment of a
//     wave -- 32 columns x 16 k -- is one fully coalesced 1 KiB buffer load straight into registers.  Waves of a
//     workgroup that share output columns re-read those lines from L1/L2; waves split the N dimension first (WN), so this
//     happens only for the 32/64-channel layers whose whole weight tensor is 18-147 KiB.  Fragments are prefetched two
//     groups of four k-steps ahead (a ring of three register sets).
//   * v_mfma_f32_32x32x16_bf16, fp32 accumulators, TM x TN tiles per wave.
//   * Epilogue as dfl_conv2d defines it (bias, ReLU, + BN(other), accumulate, NHWC / 2x2-scatter store, per-channel
//     statistics); values are rounded to bf16 once, the statistics are taken from the ROUNDED values (what consumers
//     normalise).  K slices (blockIdx.z = ranges of channel blocks) leave fp32 partial sums for convp_finish_kernel.
//
// Geometry (patch shape, resident channels, K slices, tile configuration) is chosen on the host per layer
// (convp_plan): candidates are scored by matrix work x rounds over the 256 CUs, including the fill of the last tiles.
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>

#include "common.h"
#include "convp.h"

namespace dfl {

constexpr uint32_t POOB = 0x80000000u;
#ifndef DFL_BRB_U
#define DFL_BRB_U 4
#endif
typedef unsigned int pu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf2(float a, float b) {
  const bf16x2_t h = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);   // round to nearest even (v_cvt_pk_bf16_f32)
  return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ float round_bf(float v) { return (float)(__bf16)v; }
// q / d for the small row indices of a patch (q < 65536, d < 65536): one multiply-high with m = ceil(2^32 / d) from the host
// (the epilogue decodes a patch row into (image, y, x) for every row it stores: two 35-instruction divisions each before)
__device__ __forceinline__ int pdiv(int q, uint32_t m, int d) { return d == 1 ? q : (int)__umulhi((uint32_t)q, m); }
__device__ __forceinline__ float ld_bf(const __bf16* p) { return (float)*p; }

// KS = 2 ("two k-groups"): 512 threads -- the same 4-wave tile layout twice.  Both groups share the staged patch image and
// take alternate ring groups of k-steps; their accumulators meet in LDS before the epilogue.  Two waves per SIMD instead of
// one cover each other's fragment-read and weight-load latencies in the k loop (a batch-16 layer of the deep levels has one
// workgroup per CU, i.e. ONE wave per SIMD: its k loop runs at 35-41 % of the matrix rate), the patch is staged by twice
// the threads, and -- unlike more K slices -- no partial sums go through HBM.
// AFF: what happens to an input unit on its way into the LDS image -- 0: nothing; 1: BatchNorm affine (scale, shift); 2: BatchNorm +
// ReLU backward (dfl_conv_args.x_mode): x is dy, x2 the saved ReLU output r, the staged value [r > 0] * (A dy + B r + C).
#ifndef DFL_CONVP_G1
#define DFL_CONVP_G1 4
#endif
#ifndef DFL_CONVP_PRIME
#define DFL_CONVP_PRIME 0   // 1: request a block's first two weight groups before its image is staged.  Measured (round 3): 20-30
#endif                      // more registers live across the staging, 4.75 -> 4.84 ms/step; six k-steps per ring set: 4.87
template <int WM, int WN, int TM, int TN, int AFF, bool GA, int KS = 1>
#ifndef DFL_CONVP_ADEPTH
#define DFL_CONVP_ADEPTH 1    // k-steps the A fragments run ahead (2, a second register set, measured: no faster)
#endif
#ifndef DFL_CONVP_MINW3
#define DFL_CONVP_MINW3 2
#endif
__global__ void __launch_bounds__(256 * KS, (TM * TN >= 6 || KS == 2) ? 1 : (TM * TN <= 3 ? DFL_CONVP_MINW3 : 2)) convp_kernel(const ConvP p) {
  static_assert(WM * WN == 4, "four waves per k-group");
  static_assert(KS == 1 || (KS == 2 && !GA), "two k-groups: LDS-image form only");
  static_assert(AFF != 2 || !GA, "the fused BatchNorm + ReLU backward operand is staged through LDS");
  constexpr int NT = 256 * KS;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const dfl_conv_args& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3;
  // k-group of this wave: a compile-time 0 for one group, wave-uniform (scalar register) for two -- the k loop's cursors and
  // liveness tests must stay scalar
  const int kg = KS == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid >> 8);
  const int wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
  const int S = p.pix_stride;
  const int PP = p.PH * p.PW;

  // ---- which patch, column tile and K slice.  The grid is linear; workgroup b runs on XCD b % 8 (observed, used for
  //      speed only).  Layers whose weights outweigh their activations (deep levels: 5-19 MB of weights against a 4 MB
  //      L2 per XCD) put all patches of one (column tile, K slice) pair on ONE XCD, so that every XCD streams only its
  //      share of the weights instead of all of them; the others keep patches adjacent (activation reuse in L2).
  int bpatch, btile, bslice;
  {
    const int b = blockIdx.x;
    if (p.xcd_mode == 0) {
      bpatch = b % p.npatch;
      const int r = b / p.npatch;
      btile = r % p.ntiles;
      bslice = r / p.ntiles;
    } else {
      const int x = b & 7, r = b >> 3;
      int s;
      if (p.xcd_mode == 1) {                        // pairs round-robin over the XCDs
        s = x + 8 * (r / p.npatch);
        bpatch = r % p.npatch;
      } else {                                      // fewer pairs than XCDs: 8 / pairs XCDs share one pair's patches
        const int pairs = p.ntiles * p.splits;
        s = x % pairs;
        bpatch = x / pairs + (8 / pairs) * r;
      }
      if (s >= p.ntiles * p.splits || bpatch >= p.npatch) return;
      btile = s % p.ntiles;
      bslice = s / p.ntiles;
    }
  }
  const int per_img = p.npy * p.npx;
  const int pg = bpatch / per_img, pr = bpatch - pg * per_img;
  const int ppy = pr / p.npx, ppx = pr - ppy * p.npx;
  const int img0 = pg * p.IPP, gy0 = ppy * p.PH, gx0 = ppx * p.PW;
  const int n0 = btile * (WN * TN * 32);

  // ---- LDS base of each tile row of this lane (tap (0,0), channel chunk 0, this lane's k half)
  uint32_t a_base[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    int q = (wm * TM + i) * 32 + li;
    int img = pdiv(q, p.mPP, PP);
    int r = q - img * PP;
    int py = pdiv(r, p.mPW, p.PW), px = r - py * p.PW;
    if (img >= p.IPP) img = 0, py = 0, px = 0;   // padding rows of the last tile: any valid address, results are dropped
    a_base[i] = (uint32_t)(((img * p.IH + py * a.stride) * p.IW + px * a.stride) * S + lh * 16);
  }

  __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)p.w_bytes, 0x00020000);
  uint32_t b_voff[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + (wn * TN + j) * 32 + li;
    b_voff[j] = n < a.Ntot ? (uint32_t)(n * 32 + lh * 16) : POOB;
  }

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- "live" BatchNorm statistics (dfl_conv_args.in_tot / add_tot): scale / shift are derived here, one channel per thread,
  //      into two small LDS tables behind everything else the kernel keeps in LDS ([2][128] for the input channels of the
  //      resident block, [2][BN] for the epilogue's "+ BN(add)")
  float* in_tab = reinterpret_cast<float*>(smem + p.tab_off);
  float* add_tab = in_tab + 384;                      // (in_tab: [3][128] -- scale, shift, or the three backward coefficients)
  // The epilogue's per-column constants -- bias, scale and shift of "+ BN(add)" (given, or derived from the live totals) -- go into
  // the table [3][BN] HERE, at kernel start (round 6).  They used to be fetched from global memory in front of the row loop, where the
  // whole workgroup then waited one memory round trip for them: 1.4-2 us of every launch by the phase clocks of convq_bf16.hip,
  // whose epilogue is this one -- exposed wherever a layer has one workgroup per CU.
  {
    constexpr int BN0 = WN * TN * 32;
    const bool scat0 = a.scatter2x2 != 0;
    for (int col = tid; col < BN0; col += NT) {
      const int n = n0 + col;
      float sc_ = 1.f, sh_ = 0.f, b_ = 0.f;
      if (n < a.Ntot) {
        if (a.bias != nullptr) b_ = a.bias[scat0 ? n % p.Cout : n];
        if (a.add != nullptr) {
          if (a.add_scale != nullptr) sc_ = a.add_scale[n], sh_ = a.add_shift[n];
          else if (a.add_tot != nullptr) bn_live_affine(a.add_tot, a.add_gamma, a.add_beta, a.add_count, a.bn_eps, a.Ntot, n, &sc_, &sh_);
        }
      }
      add_tab[col] = sc_;
      add_tab[BN0 + col] = sh_;
      add_tab[2 * BN0 + col] = b_;
    }
  }

  // ---- staging geometry: 16-byte units (8 channels) of the patch image, `upp` per pixel; a thread keeps its channel
  //      group, its pixel advances by 256 / upp per pass
  const int upp = p.CK >> 3, upp_sh = p.upp_shift;
  const int cg = tid & (upp - 1);
  const int dpix = NT >> upp_sh;
  const int dpix_y = dpix / p.IW, dpix_x = dpix - dpix_y * p.IW;
  const int npix = p.IPP * p.IH * p.IW;
  const int CKC = p.CK >> 4;                       // 16-channel chunks per resident block (a power of two)
  const int ckc_sh = p.upp_shift - 1;
  const int S_steps = p.T * CKC;                   // k-steps per block
  const int cin_chunks = a.Cin >> 4;
  const int KW = a.KW;
  // tap / KW without a division: (tap * ceil(256 / KW)) >> 8 is exact for tap < 16 (at most 16 taps), KW <= 16.  The k loop asks for
  // it once per k-step; the scalar division the compiler emits for it (22 dependent scalar instructions) sat in the chain
  // cursor -> LDS address -> fragment read -> matrix instruction and made that chain longer than the three matrix instructions
  // of a k-step (rocprofv3: 15-18 scalar instructions per matrix instruction in these kernels).
  const int kw_magic = (256 + KW - 1) / KW;

#ifdef DFL_CONVP_TRACE   // diagnosis build (docs/experiments/convp_trace.py): shader-clock stamps of wave 0 at the phase boundaries
  long long tr_t[10];
  tr_t[6] = tr_t[7] = tr_t[8] = 0;
  tr_t[0] = __builtin_amdgcn_s_memtime();
  tr_t[5] = __builtin_amdgcn_s_memrealtime();
  tr_t[1] = tr_t[2] = 0;
#define TR(i) tr_t[i] = __builtin_amdgcn_s_memtime();
#define TRACC(i, t0) tr_t[i] += __builtin_amdgcn_s_memtime() - (t0);
#else
#define TR(i)
#define TRACC(i, t0)
#endif
  if constexpr (GA) {
    // ---- Windows that do not overlap (1x1 / stride 1, 2x2 / stride 2; no padding, no affine on load), "global A": every
    //      input pixel belongs to ONE GEMM row, so there is nothing to share through LDS and the A fragment of a lane -- 8
    //      channels of one pixel of its row's window -- is one 16-byte load straight from the tensor, like the B fragment.  No LDS image, no staging round trip before the first matrix instruction, no barrier before the
    //      epilogue; a wave keeps two groups of k-steps (A and B) in flight.  These layers are HBM streams (K = Cin is
    //      2 ... 64 k-steps): what they need is bytes in flight, not the patch reuse the LDS image exists for.
    uint32_t a_goff[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int q = (wm * TM + i) * 32 + li;
      const int img = pdiv(q, p.mPP, PP);
      const int r = q - img * PP;
      const int py = pdiv(r, p.mPW, p.PW), px = r - py * p.PW;
      const int n = img0 + img, gy = gy0 + py, gx = gx0 + px;
      const bool ok = img < p.IPP && n < a.N && gy < p.Hg && gx < p.Wg;
      a_goff[i] = ok ? (uint32_t)(((n * a.Hin + gy * a.stride) * a.Win + gx * a.stride) * a.ldx) * 2u + (uint32_t)lh * 16u : POOB;
    }
    constexpr int GG = 2;
    const int chunks = a.Cin >> 4;
    const int steps = p.T * chunks;                    // k = tap * Cin + c, as the weights are packed
    const int ngr = (steps + GG - 1) / GG;
    pu32x4 areg[3][GG][TM], bqreg[3][GG][TN];
    int ls = 0, lcc = 0, ltx = 0;
    uint32_t lao = 0;                                  // byte offset of (tap, chunk) from the window's first pixel
    const uint32_t bstep = (uint32_t)a.Ntot * 32u;
    const uint32_t pixb = (uint32_t)a.ldx * 2u;
    const uint32_t tap_x = pixb - (uint32_t)chunks * 32u, tap_y = (uint32_t)(a.Win - a.KW) * pixb;
    auto load_gr = [&](int set) {
#pragma unroll
      for (int e = 0; e < GG; ++e) {
        const bool live = ls < steps;
        const uint32_t ao = live ? lao : 0u, bo = live ? (uint32_t)ls * bstep : 0u;
#pragma unroll
        for (int i = 0; i < TM; ++i) areg[set][e][i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, live ? a_goff[i] : POOB, ao, 0);
#pragma unroll
        for (int j = 0; j < TN; ++j) bqreg[set][e][j] = __builtin_amdgcn_raw_buffer_load_b128(rsW, live ? b_voff[j] : POOB, bo, 0);
        ++ls;
        lao += 32u;
        if (++lcc == chunks) {
          lcc = 0;
          lao += tap_x;
          if (++ltx == a.KW) {
            ltx = 0;
            lao += tap_y;
          }
        }
      }
    };
    auto compute_gr = [&](int set) {
#pragma unroll
      for (int e = 0; e < GG; ++e)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, bqreg[set][e][j]);
#pragma unroll
          for (int i = 0; i < TM; ++i)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, areg[set][e][i]), bf, acc[i][j], 0, 0, 0);
        }
    };
    load_gr(0);
    load_gr(1);
    for (int g = 0; g < ngr; g += 3) {
      load_gr(2);
      compute_gr(0);
      if (g + 1 < ngr) {
        load_gr(0);
        compute_gr(1);
      }
      if (g + 2 < ngr) {
        load_gr(1);
        compute_gr(2);
      }
    }
  } else {
  // B fragments ride a ring of three register sets, loaded two groups (2 G k-steps) ahead of their use.
  constexpr int G = TN == 1 ? DFL_CONVP_G1 : 2;    // k-steps per ring set (the ring holds 3 * G * TN fragments)
  const int ngroups_all = (S_steps + G - 1) / G;
  // k-group kg takes the ring groups kg, kg + KS, ...: local group gl is global group gl * KS + kg
  const int ngroups = (ngroups_all - kg + KS - 1) / KS;
  pu32x4 breg[3][G][TN];
  auto load_group = [&](int blk, int gl, int set) {
    const int g = gl * KS + kg;
#pragma unroll
    for (int e = 0; e < G; ++e) {
      const int s = g * G + e;
      const bool live = gl < ngroups && s < S_steps;
      const int tap = s >> ckc_sh, cc = s & (CKC - 1);
      const uint32_t soff = live ? (uint32_t)((tap * cin_chunks + blk * CKC + cc)) * (uint32_t)a.Ntot * 32u : 0u;
#pragma unroll
      for (int j = 0; j < TN; ++j) breg[set][e][j] = __builtin_amdgcn_raw_buffer_load_b128(rsW, live ? b_voff[j] : POOB, soff, 0);
    }
  };
  const int blk_begin = bslice * p.blk_per_slice;
  const int blk_end = min(blk_begin + p.blk_per_slice, p.nblk);
  for (int blk = blk_begin; blk < blk_end; ++blk) {
    const int c0 = blk * p.CK;
    if (DFL_CONVP_PRIME) {
      load_group(blk, 0, 0);
      load_group(blk, 1, 1);
    }
    // ================================================================ stage the patch image of channels [c0, c0 + CK)
    if (blk != blk_begin) __syncthreads();         // every wave is done reading the previous image
#ifdef DFL_CONVP_TRACE
    const long long tb0 = __builtin_amdgcn_s_memtime();
#endif
    {
      float sc[8], sh[8], sq[8];
      if constexpr (AFF == 1) {
        if (a.in_tot != nullptr) {                   // live statistics: this block's channels, one per thread, through LDS
          if (tid < p.CK) bn_live_affine(a.in_tot, a.in_gamma, a.in_beta, a.in_count, a.bn_eps, a.Cin, c0 + tid, in_tab + tid, in_tab + 128 + tid);
          __syncthreads();
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            sc[e] = in_tab[cg * 8 + e];
            sh[e] = in_tab[128 + cg * 8 + e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            sc[e] = a.in_scale[c0 + cg * 8 + e];
            sh[e] = a.in_shift[c0 + cg * 8 + e];
          }
        }
      }
      __amdgpu_buffer_rsrc_t rsR = rsX, rsO = rsX;
      bool store_on = false;                       // x_out: this workgroup writes the interior of its patch (column tile 0 only)
      if constexpr (AFF == 2) {                    // d(pre-activation) = [r > 0] * (sc dy + sh r + sq); no BatchNorm: 1, 0, 0
        rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x2), 0, (int)p.x2_bytes, 0x00020000);
        store_on = a.x_out != nullptr && btile == 0;
        if (store_on) rsO = __builtin_amdgcn_make_buffer_rsrc(a.x_out, 0, (int)p.xo_bytes, 0x00020000);
        if (a.in_tot != nullptr) {                   // live statistics: A, B, C of this block's channels, one per thread, through LDS
          if (tid < p.CK)
            bn_live_coef(a.in_tot, a.in_gamma, a.in_mean, a.in_invstd, a.in_count, a.Cin, c0 + tid, in_tab + tid, in_tab + 128 + tid, in_tab + 256 + tid);
          __syncthreads();
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            sc[e] = in_tab[cg * 8 + e];
            sh[e] = in_tab[128 + cg * 8 + e];
            sq[e] = in_tab[256 + cg * 8 + e];
          }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int c = c0 + cg * 8 + e;
            sc[e] = a.in_scale != nullptr ? a.in_scale[c] : 1.f;
            sh[e] = a.in_scale != nullptr ? a.in_scale[a.Cin + c] : 0.f;
            sq[e] = a.in_scale != nullptr ? a.in_scale[2 * a.Cin + c] : 0.f;
          }
        }
      }
      int pix = tid >> upp_sh;
      int img = pix / (p.IH * p.IW);
      int rem = pix - img * (p.IH * p.IW);
      int iy = rem / p.IW, ix = rem - iy * p.IW;
      const int ybase = gy0 * a.stride - a.pad, xbase = gx0 * a.stride - a.pad;
      const uint32_t cbyte = (uint32_t)((c0 + cg * 8) * 2);
      constexpr int U = AFF == 2 ? DFL_BRB_U : 8;  // loads in flight per thread (two tensors in mode 2)
      for (; pix < npix; pix += U * dpix) {
        pu32x4 v[U], v2[AFF == 2 ? U : 1];
        uint32_t offo[AFF == 2 ? U : 1];
        bool ok[U];
        int pixs[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int gy = ybase + iy, gx = xbase + ix, n = img0 + img;
          pixs[u] = pix + u * dpix;
          ok[u] = pixs[u] < npix && n < a.N && (unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win;
          const uint32_t off = (uint32_t)(((n * a.Hin + gy) * a.Win + gx) * a.ldx) * 2u + cbyte;
          v[u] = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok[u] ? off : POOB, 0, 0);
          if constexpr (AFF == 2) {
            const uint32_t off2 = (uint32_t)(((n * a.Hin + gy) * a.Win + gx) * a.ldx2) * 2u + cbyte;
            v2[u] = __builtin_amdgcn_raw_buffer_load_b128(rsR, ok[u] ? off2 : POOB, 0, 0);
            // (stride 1: gathered pixel (iy, ix) is output pixel (iy - pad, ix - pad) of the patch)
            const bool own = store_on && ok[u] && (unsigned)(iy - a.pad) < (unsigned)p.PH && (unsigned)(ix - a.pad) < (unsigned)p.PW;
            offo[u] = own ? (uint32_t)(((n * a.Hin + gy) * a.Win + gx) * a.ldxo) * 2u + cbyte : POOB;
          }
          ix += dpix_x;
          iy += dpix_y;
          if (ix >= p.IW) {
            ix -= p.IW;
            ++iy;
          }
          while (iy >= p.IH) {
            iy -= p.IH;
            ++img;
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
          if (pixs[u] < npix) {
            pu32x4 w = v[u];
            if constexpr (AFF == 1) {              // zero padding applies AFTER the BatchNorm affine: outside pixels stay 0
              if (ok[u]) {
                w.x = pack_bf2(fmaf(bf_lo(w.x), sc[0], sh[0]), fmaf(bf_hi(w.x), sc[1], sh[1]));
                w.y = pack_bf2(fmaf(bf_lo(w.y), sc[2], sh[2]), fmaf(bf_hi(w.y), sc[3], sh[3]));
                w.z = pack_bf2(fmaf(bf_lo(w.z), sc[4], sh[4]), fmaf(bf_hi(w.z), sc[5], sh[5]));
                w.w = pack_bf2(fmaf(bf_lo(w.w), sc[6], sh[6]), fmaf(bf_hi(w.w), sc[7], sh[7]));
              }
            }
            if constexpr (AFF == 2) {              // outside pixels were loaded as zeros: r = 0 there, the value stays 0
              const pu32x4 r = v2[u];
              auto brb = [](float dy, float rv, float A, float B, float Cc) { return rv > 0.f ? fmaf(A, dy, fmaf(B, rv, Cc)) : 0.f; };
              w.x = pack_bf2(brb(bf_lo(w.x), bf_lo(r.x), sc[0], sh[0], sq[0]), brb(bf_hi(w.x), bf_hi(r.x), sc[1], sh[1], sq[1]));
              w.y = pack_bf2(brb(bf_lo(w.y), bf_lo(r.y), sc[2], sh[2], sq[2]), brb(bf_hi(w.y), bf_hi(r.y), sc[3], sh[3], sq[3]));
              w.z = pack_bf2(brb(bf_lo(w.z), bf_lo(r.z), sc[4], sh[4], sq[4]), brb(bf_hi(w.z), bf_hi(r.z), sc[5], sh[5], sq[5]));
              w.w = pack_bf2(brb(bf_lo(w.w), bf_lo(r.w), sc[6], sh[6], sq[6]), brb(bf_hi(w.w), bf_hi(r.w), sc[7], sh[7], sq[7]));
              if (store_on) __builtin_amdgcn_raw_buffer_store_b128(w, rsO, offo[u], 0, 0);     // (outside the interior: out of range, dropped)
            }
            *reinterpret_cast<pu32x4*>(smem + (uint32_t)pixs[u] * (uint32_t)S + (uint32_t)cg * 16u) = w;
          }
        }
      }
    }
    __syncthreads();
    TRACC(1, tb0)
#ifdef DFL_CONVP_TRACE
    const long long tb1 = __builtin_amdgcn_s_memtime();
#endif

    // ================================================================ k-steps of this block: s = tap * CKC + chunk
    // A fragments run DFL_CONVP_ADEPTH (1) k-steps ahead of the matrix instructions that use them: the reads of step s + 1 are
    // issued before the instructions of step s (a fragment read takes 64-128 cycles plus the scalar cursor arithmetic in front of
    // its address, an instruction 32).  With depth 2 two register sets take turns (G is even: a group always starts on set 0).
    static_assert(G % 2 == 0 || DFL_CONVP_ADEPTH == 1, "the A-fragment sets alternate inside a group");
    constexpr int AD = DFL_CONVP_ADEPTH;
    bf16x8_t afq[AD][TM];
    auto fetch_a = [&](int s, int slot) {
      s = s < S_steps ? s : S_steps - 1;           // dead steps of the last group: weights were loaded as zeros
      const int tap = s >> ckc_sh, cc = s & (CKC - 1);
      const int ty = (tap * kw_magic) >> 8, tx = tap - ty * KW;
      const uint32_t aoff = (uint32_t)((ty * p.IW + tx) * S + cc * 32);
#pragma unroll
      for (int i = 0; i < TM; ++i) afq[slot][i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const pu32x4*>(smem + a_base[i] + aoff));
    };
    auto compute_group = [&](int gl, int set) {
      const int g = gl * KS + kg;
#pragma unroll
      for (int e = 0; e < G; ++e) {
        bf16x8_t af[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) af[i] = afq[e % AD][i];
        // the step this k-group takes AD steps from now
        fetch_a(e + AD < G ? g * G + e + AD : (g + KS) * G + (e + AD - G), e % AD);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, breg[set][e][j]);
#pragma unroll
          for (int i = 0; i < TM; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bf, acc[i][j], 0, 0, 0);
        }
      }
    };
#pragma unroll
    for (int d = 0; d < AD; ++d) fetch_a(kg * G + d, d);
    if (!DFL_CONVP_PRIME) {
      load_group(blk, 0, 0);
      load_group(blk, 1, 1);
    }
    for (int g = 0; g < ngroups; g += 3) {
      load_group(blk, g + 2, 2);
      compute_group(g, 0);
      if (g + 1 < ngroups) {
        load_group(blk, g + 3, 0);
        compute_group(g + 1, 1);
      }
      if (g + 2 < ngroups) {
        load_group(blk, g + 4, 1);
        compute_group(g + 2, 2);
      }
    }
    TRACC(2, tb1)
  }
  }
  TR(3)

  // ==================================================================== the two k-groups add up (through LDS, tile row by tile row)
  if constexpr (KS == 2) {
    constexpr int XP = WN * TN * 32 + 4;
    float* xg = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      __syncthreads();                                // the k loop / the previous tile row is done with this LDS region
      if (kg == 1) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) xg[(wm * 32 + mfma32_row(r, lane)) * XP + (wn * TN + j) * 32 + li] = acc[i][j][r];
      }
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] += xg[(wm * 32 + mfma32_row(r, lane)) * XP + (wn * TN + j) * 32 + li];
      }
    }
  }

  // ==================================================================== epilogue
  const bool sliced = p.splits > 1;
  if (sliced) {
    if (KS == 2 && kg == 1) return;
    // K slices: raw fp32 partial sums, row = GEMM row, 128-byte runs per accumulator row; convp_finish_kernel does the rest
    float* part = a.partial + (int64_t)bslice * p.Mtot * a.Ntot;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int q = (wm * TM + i) * 32 + 8 * g + 4 * lh;
        int img = pdiv(q, p.mPP, PP);
        int r = q - img * PP;
        int py = pdiv(r, p.mPW, p.PW), px = r - py * p.PW;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int n = img0 + img, gy = gy0 + py, gx = gx0 + px;
          const bool rok = img < p.IPP && n < a.N && gy < p.Hg && gx < p.Wg;
          const int64_t m = ((int64_t)n * p.Hg + gy) * p.Wg + gx;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + li;
            if (rok && col < a.Ntot) part[m * a.Ntot + col] = acc[i][j][4 * g + rr];
          }
          if (++px == p.PW) {
            px = 0;
            if (++py == p.PH) {
              py = 0;
              ++img;
            }
          }
        }
      }
    }
    return;
  }

  // One pass per tile row i of the waves: the WM x WN waves drop their 32 x (TN*32) accumulator tiles into LDS as a
  // [WM*32 rows][BN columns] fp32 image (lane = column: 128-byte runs per row), then every thread takes 8 consecutive
  // columns of a row -- bias, ReLU, + BN(other), accumulate, statistics on 8 values at a time, ONE 16-byte bf16 store (and
  // 16-byte loads of the partner tensors) instead of eight 2-byte accesses per tensor.
  constexpr int BN = WN * TN * 32;
  // The streamed (global A) form passes no barrier between the fill of add_tab at kernel start and the reads below: a wave that did
  // not take part in the fill can be through a short k loop (K = 64: four k-steps) before the filling wave has derived the live
  // scale / shift (16 fp64 loads + a square root per column) -- round 6 found a 1x1 layer's output changing from run to run once the
  // measured table gave it this form with 16 x 8 patches (docs/experiments/table_bisect.py).  The LDS-image form has its staging barriers.
  if constexpr (GA) __syncthreads();
  constexpr int EP = BN + 4;                          // row pitch in floats (+4: rows 4 apart on different banks)
  constexpr int UPR = BN / 8;                         // 8-column units per row
  constexpr int RPS = (NT / UPR) < WM * 32 ? (NT / UPR) : WM * 32;   // rows per step of the workgroup's threads
  static_assert(NT % UPR == 0 && (WM * 32) % RPS == 0, "row phase mapping");
  const bool rowthread = (RPS * UPR == NT) ? true : (tid < RPS * UPR);   // (two k-groups on a narrow tile: more threads than (row, unit) pairs)
  float* ep = reinterpret_cast<float*>(smem);
  const bool do_stats = a.stat_partials != nullptr || a.stat_totals != nullptr;
  const bool scat = a.scatter2x2 != 0;
  const unsigned short* addp = reinterpret_cast<const unsigned short*>(a.add);
  const unsigned short* sop = reinterpret_cast<const unsigned short*>(a.stat_other);
  unsigned short* yp = reinterpret_cast<unsigned short*>(a.y);
  const int ucol = (tid % UPR) * 8;                   // this thread's 8 columns inside the workgroup's BN
  const int urow = tid / UPR;
  const int ncol = n0 + ucol;                         // first of its GEMM columns
  const bool cok = ncol < a.Ntot;                     // (Ntot % 8 == 0: a unit is inside or outside as a whole)
  const int cab = (scat && cok) ? ncol / p.Cout : 0;
  const int cco = scat ? ncol - cab * p.Cout : ncol;
  float cbias[8], casc[8], cash[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {                       // (from the table filled at kernel start; barriers passed since)
    casc[e] = add_tab[ucol + e];
    cash[e] = add_tab[BN + ucol + e];
    cbias[e] = add_tab[2 * BN + ucol + e];
    s1[e] = 0.f;
    s2[e] = 0.f;
  }
  auto unpack = [](const pu32x4 w, float* f) {
    f[0] = bf_lo(w.x); f[1] = bf_hi(w.x); f[2] = bf_lo(w.y); f[3] = bf_hi(w.y);
    f[4] = bf_lo(w.z); f[5] = bf_hi(w.z); f[6] = bf_lo(w.w); f[7] = bf_hi(w.w);
  };
  // The per-column constants above must have ARRIVED before the row loops: with a load still outstanding at the loop header the
  // compiler covers their first use inside the loop by s_waitcnt vmcnt(0) -- which, on every iteration, also waits for the
  // previous row's STORE to be acknowledged by memory (measured: 3 us of a workgroup's 13 on a 32-column layer).
  __builtin_amdgcn_s_waitcnt(0x0F70);                 // vmcnt(0) only
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#ifdef DFL_CONVP_TRACE
    const long long te0 = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();                                  // the previous pass (or the k loop) is done with this LDS region
    TRACC(6, te0)
#ifdef DFL_CONVP_TRACE
    const long long te1 = __builtin_amdgcn_s_memtime();
#endif
    if (KS == 1 || kg == 0) {
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          ep[(wm * 32 + mfma32_row(r, lane)) * EP + (wn * TN + j) * 32 + li] = acc[i][j][r];
    }
    __syncthreads();
    TRACC(7, te1)
#ifdef DFL_CONVP_TRACE
    const long long te2 = __builtin_amdgcn_s_memtime();
#endif
    for (int rl = rowthread ? urow : WM * 32; rl < WM * 32; rl += RPS) {    // row rl of the image = row (rl / 32) * TM*32 + i*32 + rl % 32 of the patch
      const int q = ((rl >> 5) * TM + i) * 32 + (rl & 31);
      const int img = pdiv(q, p.mPP, PP);
      const int rr = q - img * PP;
      const int py = pdiv(rr, p.mPW, p.PW), px = rr - py * p.PW;
      const int n = img0 + img, gy = gy0 + py, gx = gx0 + px;
      if (!(cok && img < p.IPP && n < a.N && gy < p.Hg && gx < p.Wg)) continue;
      // (32-bit element offsets: the host checks every tensor of the epilogue against 2^32 elements; 64-bit multiplies made this
      // loop the longest stretch of a workgroup's epilogue)
      const uint32_t m = (uint32_t)((n * p.Hg + gy) * p.Wg + gx);
      float v[8];
      const float4 v0 = *reinterpret_cast<const float4*>(ep + rl * EP + ucol);
      const float4 v1 = *reinterpret_cast<const float4*>(ep + rl * EP + ucol + 4);
      v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[e] += cbias[e];
        if (a.relu) v[e] = fmaxf(v[e], 0.f);
      }
      if (addp != nullptr) {
        float o[8];
        unpack(*reinterpret_cast<const pu32x4*>(addp + (m * (uint32_t)a.ldadd + (uint32_t)ncol)), o);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += fmaf(o[e], casc[e], cash[e]);
      }
      // pixel and column of this unit in y (scatter2x2: the 2x2 position its column group stands for)
      const uint32_t opix = scat ? (uint32_t)((n * a.Hout + 2 * gy + (cab >> 1)) * a.Wout + 2 * gx + (cab & 1)) : m;
      const int ocol = scat ? cco : ncol;
      const uint32_t yo = opix * (uint32_t)a.ldy + (uint32_t)ocol;
      if (a.accumulate) {
        float o[8];
        unpack(*reinterpret_cast<const pu32x4*>(yp + yo), o);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += o[e];
      }
      pu32x4 w;
      w.x = pack_bf2(v[0], v[1]);
      w.y = pack_bf2(v[2], v[3]);
      w.z = pack_bf2(v[4], v[5]);
      w.w = pack_bf2(v[6], v[7]);
      *reinterpret_cast<pu32x4*>(yp + yo) = w;
      if (do_stats) {
        float vr[8], u[8];
        unpack(w, vr);                                // statistics of the values as stored
        if (sop != nullptr) {
          unpack(*reinterpret_cast<const pu32x4*>(sop + (opix * (uint32_t)a.ldso + (uint32_t)ocol)), u);
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) u[e] = vr[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          s1[e] += vr[e];
          s2[e] = fmaf(vr[e], u[e], s2[e]);
        }
      }
    }
    TRACC(8, te2)
  }
#ifdef DFL_CONVP_TRACE
  if (tid == 0 && a.partial != nullptr) {
    long long* sink = reinterpret_cast<long long*>(a.partial) + (int64_t)(bpatch + p.npatch * btile) * 12;
    sink[8] = tr_t[6]; sink[9] = tr_t[7]; sink[10] = tr_t[8];
    sink[0] = tr_t[0]; sink[1] = tr_t[1]; sink[2] = tr_t[2]; sink[3] = tr_t[3]; sink[4] = __builtin_amdgcn_s_memtime();
    sink[5] = tr_t[5]; sink[6] = __builtin_amdgcn_s_memrealtime(); sink[7] = 1;
  }
#endif
  if (!do_stats) return;
  // per-column sums of the workgroup -> one row of stat_partials (rows = patches): threads of one column unit add up
  // through LDS in a fixed order
  __syncthreads();
  float* red = reinterpret_cast<float*>(smem);        // [RPS][2][BN]
  if (rowthread) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      red[(urow * 2 + 0) * BN + ucol + e] = s1[e];
      red[(urow * 2 + 1) * BN + ucol + e] = s2[e];
    }
  }
  __syncthreads();
  for (int idx = tid; idx < 2 * BN; idx += NT) {
    const int which = idx / BN, col = idx - which * BN;
    const int n = n0 + col;
    if (n < a.Ntot) {
      float sum = 0.f;
      for (int w = 0; w < RPS; ++w) sum += red[(w * 2 + which) * BN + col];
      if (scat) {                                      // rows = (patch, 2x2 position): [.][2][Cout], the sums of y's Cout channels
        const int ab = n / p.Cout, co = n - ab * p.Cout;
        if (a.stat_totals != nullptr) bn_live_add(a.stat_totals, bpatch * 4 + ab, which, p.Cout, co, sum);
        else a.stat_partials[(((int64_t)bpatch * 4 + ab) * 2 + which) * p.Cout + co] = sum;
      } else if (a.stat_totals != nullptr) {           // live statistics: added to the layer's totals (hardware fp64 atomics)
        bn_live_add(a.stat_totals, bpatch, which, a.Ntot, n, sum);
      } else {
        a.stat_partials[((int64_t)bpatch * 2 + which) * a.Ntot + n] = sum;
      }
    }
  }
}

// K-slice finish: y = epilogue(sum_s partial[s]) -- conv_finish_kernel of conv_gemm.hip for bf16 tensors.
__global__ void __launch_bounds__(256) convp_finish_kernel(const ConvP p, int TX, int rows_per_block) {
  __shared__ float red[2][256];
  const dfl_conv_args& a = p.a;
  const int Ntot = a.Ntot;
  const int TY = 256 / TX;
  const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
  const int n = blockIdx.y * TX + tx;
  const bool nok = n < Ntot;
  int co = n, ab = 0;
  if (a.scatter2x2 && nok) {
    ab = n / p.Cout;
    co = n - ab * p.Cout;
  }
  const float bias = (a.bias != nullptr && nok) ? a.bias[co] : 0.f;
  float asc = 1.f, ash = 0.f;
  if (a.add != nullptr && a.add_scale != nullptr && nok) {
    asc = a.add_scale[n];
    ash = a.add_shift[n];
  } else if (a.add != nullptr && a.add_tot != nullptr && nok) {
    bn_live_affine(a.add_tot, a.add_gamma, a.add_beta, a.add_count, a.bn_eps, Ntot, n, &asc, &ash);
  }
  const __bf16* addp = reinterpret_cast<const __bf16*>(a.add);
  const __bf16* sop = reinterpret_cast<const __bf16*>(a.stat_other);
  __bf16* yp = reinterpret_cast<__bf16*>(a.y);
  const float osc = (a.out_scale != nullptr && nok) ? a.out_scale[co] : 1.f, osh = (a.out_scale != nullptr && nok) ? a.out_shift[co] : 0.f;
  const int64_t slice = (int64_t)p.Mtot * Ntot;
  const int r0 = blockIdx.x * rows_per_block;
  const int r1 = min(r0 + rows_per_block, p.Mtot);
  float s1 = 0.f, s2 = 0.f;
  if (nok) {
    for (int m = r0 + ty; m < r1; m += TY) {
      const float* pp = a.partial + (int64_t)m * Ntot + n;
      float v = 0.f;
      for (int s = 0; s < p.splits; ++s) v += pp[(int64_t)s * slice];
      v += bias;
      if (a.relu) v = fmaxf(v, 0.f);
      if (addp != nullptr) v += fmaf(ld_bf(addp + (int64_t)m * a.ldadd + n), asc, ash);
      int64_t opix = m;
      if (a.scatter2x2) {
        const int jx = m % p.Wg;
        const int t = m / p.Wg;
        const int iy = t % p.Hg;
        const int ni = t / p.Hg;
        opix = ((int64_t)ni * a.Hout + 2 * iy + (ab >> 1)) * a.Wout + 2 * jx + (ab & 1);
      }
      __bf16* dst = yp + opix * a.ldy + co;
      if (a.accumulate) v += (float)*dst;
      __bf16 hv = (__bf16)v;
      if (a.out_scale != nullptr) hv = (__bf16)fmaf((float)hv, osc, osh);       // (latency form with K slices: the consumer's BatchNorm, include/dfl_hip.h)
      *dst = hv;
      const float vr = (float)hv;
      const float u = (sop != nullptr) ? ld_bf(sop + opix * a.ldso + co) : vr;
      s1 += vr;
      s2 = fmaf(vr, u, s2);
    }
  }
  if (a.stat_partials == nullptr && a.stat_totals == nullptr) return;
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  __syncthreads();
  if (ty == 0 && nok) {
    float t1 = 0.f, t2 = 0.f;
    for (int y = 0; y < TY; ++y) {
      t1 += red[0][y * TX + tx];
      t2 += red[1][y * TX + tx];
    }
    if (a.scatter2x2 && a.stat_totals != nullptr) {
      bn_live_add(a.stat_totals, (int)blockIdx.x * 4 + ab, 0, p.Cout, co, t1);
      bn_live_add(a.stat_totals, (int)blockIdx.x * 4 + ab, 1, p.Cout, co, t2);
    } else if (a.scatter2x2) {                         // rows = (row block, 2x2 position), see convp_kernel
      a.stat_partials[(((int64_t)blockIdx.x * 4 + ab) * 2 + 0) * p.Cout + co] = t1;
      a.stat_partials[(((int64_t)blockIdx.x * 4 + ab) * 2 + 1) * p.Cout + co] = t2;
    } else if (a.stat_totals != nullptr) {
      bn_live_add(a.stat_totals, (int)blockIdx.x, 0, Ntot, n, t1);
      bn_live_add(a.stat_totals, (int)blockIdx.x, 1, Ntot, n, t2);
    } else {
      a.stat_partials[((int64_t)blockIdx.x * 2 + 0) * Ntot + n] = t1;
      a.stat_partials[((int64_t)blockIdx.x * 2 + 1) * Ntot + n] = t2;
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------------

struct TileCfg { int WM, WN, TM, TN, GA, KS, Q, P, NL; };   // GA: A fragments straight from global memory (1x1 windows, see the kernel); KS: k-groups (1 or 2);
                                                       // Q: the unrolled 3x3 form of convq_bf16.hip (8 WM x 12 patch, 32 WN columns); P: its persistent form;
                                                       // NL: layout + 1 of the narrow 3x3 form of convn_bf16.hip
// value reported by dfl_conv_config for these kernels = 16 + index
static const TileCfg kTiles[] = {{4, 1, 2, 1, 0, 1}, {4, 1, 1, 1, 0, 1}, {2, 2, 4, 1, 0, 1}, {2, 2, 3, 1, 0, 1}, {2, 2, 2, 1, 0, 1}, {1, 4, 2, 1, 0, 1},
                                 {1, 4, 3, 1, 0, 1}, {1, 4, 4, 1, 0, 1}, {1, 4, 6, 1, 0, 1}, {1, 4, 9, 1, 0, 1}, {2, 2, 1, 1, 0, 1}, {1, 4, 1, 1, 0, 1},
                                 {4, 1, 3, 1, 0, 1}, {4, 1, 4, 1, 0, 1}, {2, 2, 6, 1, 0, 1},
                                 // two column tiles per wave: half the LDS fragment reads per matrix instruction
                                 {2, 2, 2, 2, 0, 1}, {2, 2, 3, 2, 0, 1}, {2, 2, 4, 2, 0, 1}, {4, 1, 2, 2, 0, 1}, {4, 1, 3, 2, 0, 1}, {1, 4, 2, 2, 0, 1}, {1, 4, 3, 2, 0, 1},
                                 // 1x1 windows streamed from global memory (22 ...): only ever chosen through the measured table
                                 {4, 1, 1, 1, 1, 1}, {4, 1, 2, 1, 1, 1}, {2, 2, 1, 1, 1, 1}, {2, 2, 2, 1, 1, 1}, {1, 4, 1, 1, 1, 1}, {1, 4, 2, 1, 1, 1},
                                 {4, 1, 1, 2, 1, 1}, {2, 2, 1, 2, 1, 1},
                                 // two k-groups (512 threads, 30 ...): only ever chosen through the measured table
                                 {1, 4, 3, 1, 0, 2}, {1, 4, 2, 1, 0, 2}, {1, 4, 4, 1, 0, 2}, {2, 2, 3, 1, 0, 2}, {2, 2, 2, 1, 0, 2}, {1, 4, 2, 2, 0, 2},
                                 {2, 2, 2, 2, 0, 2}, {4, 1, 3, 1, 0, 2}, {4, 1, 2, 1, 0, 2},
                                 // (39 = CONVS_TILE: the latency form plans itself, never through this table)
                                 {0, 0, 0, 0, 0, 0, 0},
                                 // unrolled 3x3 form (convq_bf16.hip; 40: one k-group, 41: two): chosen through the measured table or, for
                                 // the layer shapes convq_default() names, by default
                                 {1, 4, 3, 1, 0, 1, 1}, {1, 4, 3, 1, 0, 2, 1},
                                 // ... 42: eight waves on a 16 x 12 patch; 43-45: 64 columns; 46-48: 32 columns (WM x 8 patch rows)
                                 {2, 4, 3, 1, 0, 1, 1}, {2, 2, 3, 1, 0, 1, 1}, {4, 2, 3, 1, 0, 1, 1}, {2, 2, 3, 1, 0, 2, 1},
                                 {4, 1, 3, 1, 0, 1, 1}, {8, 1, 3, 1, 0, 1, 1}, {4, 1, 3, 1, 0, 2, 1},
                                 // 49-57: the same nine layouts, persistent (a workgroup walks several patches, the next one staged ahead)
                                 {1, 4, 3, 1, 0, 1, 1, 1}, {1, 4, 3, 1, 0, 2, 1, 1}, {2, 4, 3, 1, 0, 1, 1, 1}, {2, 2, 3, 1, 0, 1, 1, 1}, {4, 2, 3, 1, 0, 1, 1, 1},
                                 {2, 2, 3, 1, 0, 2, 1, 1}, {4, 1, 3, 1, 0, 1, 1, 1}, {8, 1, 3, 1, 0, 1, 1, 1}, {4, 1, 3, 1, 0, 2, 1, 1},
                                 // 58-63: the narrow 3x3 form (convn_bf16.hip: 32 / 64 output columns, its six layouts); table only
                                 {4, 1, 1, 1, 0, 1, 0, 0, 1}, {4, 1, 1, 1, 0, 1, 0, 0, 2}, {4, 1, 1, 1, 0, 1, 0, 0, 3}, {4, 1, 1, 1, 0, 1, 0, 0, 4},
                                 {4, 1, 1, 1, 0, 1, 0, 0, 5}, {4, 1, 1, 1, 0, 1, 0, 0, 6},
                                 // 64, 65: its persistent form (layouts 4 and 5; 32 -> 32 layers)
                                 {4, 1, 1, 1, 0, 1, 0, 1, 5}, {4, 1, 1, 1, 0, 1, 0, 1, 6}};
constexpr int kNumTiles = (int)(sizeof(kTiles) / sizeof(kTiles[0]));
static_assert(kNumTiles == CONVN_PERS_TILE + 2 && CONVN_PERS_TILE == CONVN_TILE + CONVN_LAYOUTS && CONVN_TILE == CONVQ_TILE + 2 * CONVQ_LAYOUTS && CONVQ_TILE == CONVS_TILE + 1, "convp.h: CONVS_TILE is the index behind the patch-kernel configurations, the unrolled 3x3 forms follow it");
constexpr size_t kLdsSoft = 64 * 1024, kLdsHard = 150 * 1024;

// Row blocks of convp_finish_kernel = rows of stat_partials the following finalize kernel has to read.  Few (<= FIN_ROWS) and
// the finalize launch reads a few hundred KB instead of up to 4.7 MB (one row per 1-2 GEMM rows on the deep levels: its
// 10-12 us there against 5-6 us elsewhere); the finish kernel keeps its parallelism through narrow column blocks (FIN_TX).
constexpr int FIN_ROWS = 128;                  // (measured 2048 / 512 / 256 / 128 rows: 4.89 / 4.88 / 4.865 / 4.85 ms per step)
constexpr int FIN_TX = 32;
static int finish_rows_p(int M, int Ntot) {
  int64_t nb = ceil_div((int64_t)M * Ntot, 1024);
  if (nb > 2048) nb = 2048;
  if (nb > FIN_ROWS) nb = FIN_ROWS;
  if (nb > M) nb = M;
  return nb < 1 ? 1 : (int)nb;
}

int convp_finish_rows(const ConvP& p) { return finish_rows_p(p.Mtot, p.a.Ntot); }

// Fill p's geometry for tile configuration t and a patch of (ipp images, ph x pw pixels); returns false if impossible.
static bool try_geometry(const dfl_conv_args& a, ConvP* p, const TileCfg& t, int ipp, int ph, int pw, int want_splits,
                         double* cost) {
  const int QP = t.WM * t.TM * 32;
  if (t.WM == 0) return false;
  if (t.NL) {
    // narrow 3x3 form: the layout fixes the patch, all input channels pass through one image in blocks of 32, no K slices, one
    // column tile; workgroups are dealt to the XCDs in runs of q_ngroups patches.  Scored only by measurement.
    int nph, npw;
    convn_patch(t.NL - 1, &nph, &npw);
    if (ipp != 1 || ph != nph || pw != npw || want_splits > 1 || !convn_shape_ok(a) || !convn_layout_ok(t.NL - 1, a.Ntot, a.Cin)) return false;
    p->IPP = 1;
    p->PH = ph;
    p->PW = pw;
    p->mPP = (uint32_t)(((1ull << 32) + (uint64_t)(ph * pw) - 1) / (uint64_t)(ph * pw));
    p->mPW = (uint32_t)(((1ull << 32) + (uint64_t)pw - 1) / (uint64_t)pw);
    p->npy = (int)ceil_div(p->Hg, ph);
    p->npx = (int)ceil_div(p->Wg, pw);
    p->npatch = a.N * p->npy * p->npx;
    if (p->npatch >= 65536) return false;              // (the kernel divides patch numbers by multiply-high)
    p->IH = ph + 2;
    p->IW = pw + 2;
    p->CK = 32;
    p->nblk = a.Cin / 32;
    p->splits = 1;
    p->blk_per_slice = p->nblk;
    p->ntiles = 1;
    p->pix_stride = 80;
    p->upp_shift = 2;
    p->lds_bytes = p->IH * p->IW * 80;
    p->xcd_mode = 0;
    p->q_ngroups = (int)ceil_div(p->npatch, 8);
    p->q_stride = p->q_ngroups;
    p->grid = 8 * p->q_ngroups;
    if (t.P) {                                         // persistent: two workgroups per CU (64 per XCD), each walks several patches of its XCD's run
      if (!convn_pers_ok(t.NL - 1, a) || p->q_ngroups <= 64) return false;
      p->q_stride = 64;
      p->grid = 8 * 64;
    }
    auto magic = [](int d) { return (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)d); };
    p->qm_npatch = magic(p->npatch);
    p->qm_ntiles = magic(1);
    p->qm_perimg = magic(p->npy * p->npx);
    p->qm_npx = magic(p->npx);
    if (convn_lds_bytes(t.NL - 1, a.Cin, a.Ntot, t.P) > 160 * 1024) return false;
    *cost = 1e289;
    return true;
  }
  if (ipp < 1 || ph < 1 || pw < 1 || (int64_t)ipp * ph * pw > QP) return false;
  if (ipp > 1 && (ph != p->Hg || pw != p->Wg)) return false;
  if (t.Q && (ipp != 1 || ph != 8 * t.WM || pw != 12 || !convq_shape_ok(a))) return false;
  p->IPP = ipp;
  p->PH = ph;
  p->PW = pw;
  p->mPP = (uint32_t)(((1ull << 32) + (uint64_t)(ph * pw) - 1) / (uint64_t)(ph * pw));
  p->mPW = (uint32_t)(((1ull << 32) + (uint64_t)pw - 1) / (uint64_t)pw);
  p->npy = (int)ceil_div(p->Hg, ph);
  p->npx = (int)ceil_div(p->Wg, pw);
  p->npatch = (int)ceil_div(a.N, ipp) * p->npy * p->npx;
  p->IH = (ph - 1) * a.stride + a.KH;
  p->IW = (pw - 1) * a.stride + a.KW;
  const int64_t npix = (int64_t)ipp * p->IH * p->IW;
  if (t.GA) {
    // streamed non-overlapping window (1x1 / stride 1, 2x2 / stride 2): no LDS image, no K slices; scored only by
    // measurement (tools/tune_convp.py)
    if (a.KH != a.KW || a.stride != a.KH || a.KH > 2 || a.pad != 0 || a.in_scale != nullptr || a.in_tot != nullptr || a.x_mode != 0 || want_splits > 1) return false;
    p->CK = 16;
    p->nblk = a.Cin / 16;
    p->splits = 1;
    p->blk_per_slice = p->nblk;
    p->pix_stride = 48;
    p->upp_shift = 1;
    p->lds_bytes = 0;
    p->ntiles = (int)ceil_div(a.Ntot, t.WN * t.TN * 32);
    p->xcd_mode = 0;
    p->grid = p->npatch * p->ntiles;
    *cost = 1e290;
    return true;
  }
  // resident channels: the largest power-of-two multiple of 16 (<= 128, dividing Cin) whose image fits
  int ck = 128;
  while (ck > 16 && (a.Cin % ck != 0 || (!t.Q && npix * (ck * 2 + 16) > (int64_t)kLdsSoft))) ck >>= 1;
  if (a.Cin % ck != 0) return false;
  const int qmode = ((int)(&t - kTiles) - CONVQ_TILE) % CONVQ_LAYOUTS;    // (the unrolled form: its wave layout)
  const int ck_min = t.Q ? 32 : 16;
  if (t.Q) {                                       // the largest resident block the layout is instantiated for and whose image fits
    while (ck >= 32 && (a.Cin % ck != 0 || !convq_ck_ok(qmode, ck) || convq_lds_bytes(ck, qmode, a.Cin / ck > 1 ? 2 : 1, t.P) > 160 * 1024)) ck >>= 1;
  }
  if (ck < ck_min) return false;
  size_t lds = (size_t)npix * (ck * 2 + 16);
  if (lds > kLdsHard) return false;
  const int bn = t.WN * t.TN * 32;
  const int ntiles = (int)ceil_div(a.Ntot, bn);
  const int64_t wgs = (int64_t)p->npatch * ntiles;
  int nblk = a.Cin / ck;
  // K slices: only whole channel blocks; more blocks (smaller ck) when the layer needs the parallelism
  int splits = 1;
  if (want_splits > 1) {
    splits = want_splits;
    while (nblk % splits != 0 && ck > ck_min) {
      ck >>= 1;
      nblk = a.Cin / ck;
    }
    if (t.Q && !convq_ck_ok(qmode, ck)) return false;
    if (nblk % splits != 0) return false;
    lds = (size_t)npix * (ck * 2 + 16);
  }
  p->CK = ck;
  p->nblk = nblk;
  p->splits = splits;
  p->ntiles = ntiles;
  {
    // workgroup -> XCD placement (see the kernel): weight-major when one pass over the weights is the larger stream
    static const int xcd_env = [] {
      const char* e = getenv("DFL_CONVP_XCD");       // 0: never weight-major (A/B measurements)
      return e ? atoi(e) : 1;
    }();
    const int64_t wbytes = (int64_t)a.KH * a.KW * a.Cin * a.Ntot * 2;
    const int64_t abytes = (int64_t)a.N * a.Hin * a.Win * a.Cin * 2;
    const int pairs = ntiles * splits;
    p->xcd_mode = 0;
    p->grid = p->npatch * pairs;
    if (xcd_env != 0 && wbytes > abytes && wbytes > (1 << 20)) {
      if (pairs >= 8 && pairs % 8 == 0) {
        p->xcd_mode = 1;
      } else if (pairs < 8 && 8 % pairs == 0 && p->npatch >= 8 / pairs) {
        p->xcd_mode = 2;
        p->grid = 8 * (int)ceil_div(p->npatch, 8 / pairs);
      }
    }
  }
  if (t.Q) {
    if (p->grid >= 65536 || p->npatch >= 65536) return false;      // (the kernel divides workgroup numbers by multiply-high)
    const size_t qlds = convq_lds_bytes(ck, qmode, nblk / splits, t.P);
    if (qlds > 160 * 1024) return false;
    if (t.P) {
      if (!convq_pers_ok(qmode, ck, a.x_mode)) return false;
      // persistent: as many workgroups per (column tile, K slice) as the CUs hold at once, patches dealt out evenly; pointless
      // (and refused) when that leaves one patch per workgroup
      int occ = convq_threads(qmode) == 512 ? 1 : 2;
      if ((size_t)occ * qlds > 160 * 1024) occ = 1;
      int g0 = 256 * occ / (ntiles * splits);
      if (g0 < 1) g0 = 1;
      if (g0 >= p->npatch) return false;
      const int per = (int)ceil_div(p->npatch, g0);
      p->q_ngroups = (int)ceil_div(p->npatch, per);
      p->xcd_mode = 0;
      p->grid = p->q_ngroups * ntiles * splits;
    }
    auto magic = [](int d) { return (uint32_t)(((1ull << 32) + (uint64_t)d - 1) / (uint64_t)d); };
    p->qm_npatch = magic(t.P ? p->q_ngroups : p->npatch);
    p->qm_ntiles = magic(ntiles);
    p->qm_perimg = magic(p->npy * p->npx);
    p->qm_npx = magic(p->npx);
  }
  p->blk_per_slice = nblk / splits;
  p->pix_stride = ck * 2 + 16;
  int sh = 0;
  while ((1 << sh) < (ck >> 3)) ++sh;
  p->upp_shift = sh;
  p->lds_bytes = (int)lds;
  // Cost model (cycles), fitted to per-layer timings on MI355X (tools/kbench_bf16.py).  A workgroup's k loop is bound by
  // the matrix pipe (32 cycles per instruction and wave, one wave per SIMD), by its LDS fragment reads, or by the B
  // fragments coming from L1/L2 (each wave 1 KiB per k-step and column tile: ~32 B/clk/CU when the four waves read
  // different columns, more when they share them); around it sit latencies nobody inside the workgroup hides: staging
  // the patch (one global round trip + the LDS writes) and one pass per accumulator tile row in the epilogue.  `occ`
  // workgroups share a CU (LDS, registers: 512 / allocation of the tile configuration): while one waits another computes.
  static const int occ_by_tm[10] = {0, 5, 4, 4, 3, 2, 2, 2, 2, 2};
  int occ = (int)((160 * 1024) / (lds + 1024));
  if (occ > occ_by_tm[t.TM]) occ = occ_by_tm[t.TM];
  if (occ < 1) occ = 1;
  const int64_t total = wgs * splits;
  int64_t eff = ceil_div(total, 256);
  if (eff > occ) eff = occ;                                                      // workgroups actually sharing a CU
  const int64_t rounds = ceil_div(total, 256 * eff);
  const int steps = (a.KH * a.KW * (ck / 16) + 3) / 4 * 4 * p->blk_per_slice;
  const double mfma = (double)t.TM * t.TN * steps * 32.0;
  const double ldsread = (double)t.TM * steps * 4.0 * 4.0;                       // ds_read_b128 cycles of the 4 waves
  const double bbw = t.WN == 4 ? 32.0 : (t.WN == 2 ? 48.0 : 64.0);
  const double bload = (double)steps * t.TN * 4096.0 / bbw;
  double loop = mfma > ldsread ? mfma : ldsread;
  if (bload > loop) loop = bload;
  const double lat = (double)p->blk_per_slice * (4000.0 + (double)lds / 40.0) + 2000.0 + (splits > 1 ? 1500.0 : 2500.0 * t.TM);
  const double busy = (double)eff * (loop + (double)p->blk_per_slice * (double)lds / 40.0);   // what the CU's pipes must do per round
  const double per_round = busy > lat + loop ? busy : lat + loop;
  double c = (double)rounds * per_round;
  if (splits > 1) c += (double)p->Mtot * a.Ntot * 4.0 * (splits + 1) / 2000.0 + 6000.0;   // partial sums: bytes / (B per cycle of the chip) + the finish launch
  *cost = (t.KS == 2 || t.Q) ? 1e289 : c;              // (two k-groups, the unrolled form: scored only by measurement, tools/tune_convp.py)
  return true;
}

// ---- geometry candidates, forced geometry, tuning table
struct ConvGeom { int tile, ipp, ph, pw, splits; };
static thread_local ConvGeom t_force = {-1, 0, 0, 0, 0};
struct TuneEntry { int key[10]; ConvGeom g; };
static std::mutex g_tune_mu;
static std::vector<TuneEntry> g_tune;

static void tune_key(const dfl_conv_args& a, int* k) {
  k[0] = a.N; k[1] = a.Hin; k[2] = a.Win; k[3] = a.Cin; k[4] = a.Ntot;
  k[5] = a.KH; k[6] = a.KW; k[7] = a.stride; k[8] = a.pad; k[9] = a.scatter2x2 ? 1 : 0;
}

static bool tune_lookup(const dfl_conv_args& a, ConvGeom* g) {
  int k[10];
  tune_key(a, k);
  std::lock_guard<std::mutex> lock(g_tune_mu);
  for (const TuneEntry& e : g_tune)
    if (memcmp(e.key, k, sizeof(k)) == 0) {
      *g = e.g;
      return true;
    }
  return false;
}

// Tile configurations x patch shapes x K slices that are valid for the layer.  The cost model looks at the tiles whose
// column count matches the layer (`wide` false); tuners also get the narrower column tiles (more workgroups).
template <typename F>
static void for_each_candidate(const dfl_conv_args& a, const ConvP& base, int force_splits, bool wide, F&& f) {
  static const unsigned tile_off = [] {      // DFL_CONVP_TILES_OFF: bit mask of tile configurations to leave out (A/B measurements)
    const char* e = getenv("DFL_CONVP_TILES_OFF");
    return e ? (unsigned)strtoul(e, nullptr, 0) : 0u;
  }();
  const int n32 = (int)ceil_div(a.Ntot, 32) * 32;
  for (int ti = 0; ti < kNumTiles; ++ti) {
    if ((tile_off >> ti) & 1u) continue;
    const TileCfg& t = kTiles[ti];
    const int bn = t.WN * t.TN * 32;
    if (t.NL) {
      // (takes all the layer's columns itself)
    } else if (wide) {
      if (bn > n32 || (bn < 64 && n32 >= 64) || (t.TN > 1 && bn > a.Ntot)) continue;
    } else {
      if (t.TN != 1 || t.GA || t.KS != 1) continue;   // the model was fitted on the one-column-tile, one-k-group configurations
      if (a.Ntot <= 32 && t.WN != 1) continue;
      if (a.Ntot > 32 && a.Ntot <= 64 && t.WN != 2) continue;
      if (a.Ntot > 64 && t.WN != 4) continue;
    }
    const int QP = t.WM * t.TM * 32;
    int shapes[24][3];
    int ns = 0;
    const int HW = base.Hg * base.Wg;
    if (t.NL) {                                      // the layout's patch
      if (!wide || !convn_shape_ok(a)) continue;
      int nph, npw;
      convn_patch(t.NL - 1, &nph, &npw);
      shapes[ns][0] = 1; shapes[ns][1] = nph; shapes[ns][2] = npw; ++ns;
    } else
    if (t.Q) {                                       // one patch shape
      if (!wide) continue;
      shapes[ns][0] = 1; shapes[ns][1] = 8 * t.WM; shapes[ns][2] = 12; ++ns;
    } else
    if (!t.NL && HW <= QP) {                         // whole images
      shapes[ns][0] = QP / HW; shapes[ns][1] = base.Hg; shapes[ns][2] = base.Wg; ++ns;
    }
    if (!t.Q && !t.NL && base.Wg <= QP) {            // whole rows
      int ph = QP / base.Wg;
      if (ph > base.Hg) ph = base.Hg;
      shapes[ns][0] = 1; shapes[ns][1] = ph; shapes[ns][2] = base.Wg; ++ns;
    }
    static const int kPh[] = {1, 2, 4, 8, 16, 3, 6, 12};        // row pieces (the model looks at the powers of two)
    for (int k = 0; k < (wide ? 8 : 5) && ns < 22 && !t.Q && !t.NL; ++k) {
      const int ph = kPh[k];
      int pw = QP / ph;
      if (pw >= base.Wg || ph > base.Hg) continue;
      shapes[ns][0] = 1; shapes[ns][1] = ph; shapes[ns][2] = pw; ++ns;
    }
    for (int si = 0; si < ns; ++si) {
      for (int sp = 1; sp <= 32; sp *= 2) {
        if (force_splits > 0 && sp != force_splits) continue;
        if (!wide && a.scatter2x2 == 0 && a.accumulate == 0 && sp > 1 && base.T * a.Cin < 1024) break;   // short K: never sliced
        ConvP q = base;
        double c;
        if (!try_geometry(a, &q, t, shapes[si][0], shapes[si][1], shapes[si][2], sp, &c)) continue;
        if (q.splits != sp) continue;
        f(ti, q, c);
      }
    }
  }
}

static int convp_plan_search(const dfl_conv_args* a, ConvP* p, int force_splits) {
  DFL_REQUIRE(a->x && a->w && a->y, "dfl_conv2d (bf16): x, w and y are required");
  DFL_REQUIRE(a->N > 0 && a->Hin > 0 && a->Win > 0 && a->Cin > 0 && a->Ntot > 0, "dfl_conv2d (bf16): bad sizes");
  DFL_REQUIRE(a->Ntot % 8 == 0 && a->ldy % 8 == 0 && aligned16(a->y) && (a->add == nullptr || (a->ldadd % 8 == 0 && aligned16(a->add))) &&
                  (a->stat_other == nullptr || (a->ldso % 8 == 0 && aligned16(a->stat_other))),
              "dfl_conv2d (bf16): output columns and the pixel strides of y / add / stat_other must be multiples of 8, tensors 16-byte aligned");
  {
    const int64_t opx = (int64_t)a->N * a->Hout * a->Wout, lim = 1ll << 32;
    DFL_REQUIRE(opx * a->ldy < lim && (a->add == nullptr || opx * a->ldadd < lim) && (a->stat_other == nullptr || opx * a->ldso < lim),
                "dfl_conv2d (bf16): y / add / stat_other must stay below 2^32 elements");
  }
  DFL_REQUIRE(a->Cin % 16 == 0 && a->ldx % 8 == 0 && aligned16(a->x) && aligned16(a->w),
              "dfl_conv2d (bf16): needs Cin %% 16 == 0, ldx %% 8 == 0 and 16-byte aligned x / w (Cin = %d, ldx = %d)", a->Cin, a->ldx);
  DFL_REQUIRE(a->w_split == 2, "dfl_conv2d (bf16): weights must be packed with dfl_pack_job.split = 2");
  DFL_REQUIRE(a->x_mode != 0 || (a->in_scale == nullptr) == (a->in_shift == nullptr), "dfl_conv2d: in_scale/in_shift go together");
  DFL_REQUIRE(a->in_tot == nullptr || (a->in_scale == nullptr && a->in_gamma != nullptr && a->in_count > 0 &&
                                       (a->x_mode == 0 ? a->in_beta != nullptr : (a->in_mean != nullptr && a->in_invstd != nullptr))),
              "dfl_conv2d (bf16): in_tot replaces in_scale / in_shift and needs in_gamma, in_count and in_beta (x_mode 0) or in_mean, in_invstd (x_mode 1)");
  DFL_REQUIRE(a->add_tot == nullptr || (a->add != nullptr && a->add_scale == nullptr && a->add_gamma != nullptr && a->add_beta != nullptr && a->add_count > 0),
              "dfl_conv2d (bf16): add_tot replaces add_scale / add_shift and needs add, add_gamma, add_beta, add_count");

  DFL_REQUIRE((a->add_scale == nullptr) == (a->add_shift == nullptr), "dfl_conv2d: add_scale/add_shift go together");
  DFL_REQUIRE(a->KH * a->KW <= 16 && a->KH > 0 && a->KW > 0 && a->stride > 0 && a->pad >= 0, "dfl_conv2d (bf16): bad window");
  memset(p, 0, sizeof(*p));
  p->a = *a;
  if (a->scatter2x2) {
    DFL_REQUIRE(a->KH == 1 && a->KW == 1 && a->stride == 1 && a->pad == 0, "dfl_conv2d: scatter2x2 needs a 1x1 gather");
    DFL_REQUIRE(a->Ntot % 4 == 0 && a->Hout >= 2 * a->Hin && a->Wout >= 2 * a->Win, "dfl_conv2d: scatter2x2 geometry");
    DFL_REQUIRE(a->add == nullptr, "dfl_conv2d: scatter2x2 has no add epilogue");
    p->Hg = a->Hin;
    p->Wg = a->Win;
    p->Cout = a->Ntot / 4;
  } else {
    const int ho = (a->Hin + 2 * a->pad - a->KH) / a->stride + 1;
    const int wo = (a->Win + 2 * a->pad - a->KW) / a->stride + 1;
    DFL_REQUIRE(ho == a->Hout && wo == a->Wout, "dfl_conv2d: Hout/Wout (%d,%d) do not match the window (%d,%d)", a->Hout,
                a->Wout, ho, wo);
    p->Hg = a->Hout;
    p->Wg = a->Wout;
    p->Cout = a->Ntot;
  }
  DFL_REQUIRE(a->ldy >= p->Cout, "dfl_conv2d: ldy < Cout");
  const int64_t M = (int64_t)a->N * p->Hg * p->Wg;
  DFL_REQUIRE(M < (1ll << 31), "dfl_conv2d: too many pixels");
  p->Mtot = (int)M;
  p->T = a->KH * a->KW;
  const int64_t xb = (((int64_t)a->N * a->Hin * a->Win - 1) * a->ldx + a->Cin) * 2;
  const int64_t wb = (int64_t)p->T * a->Cin * a->Ntot * 2;
  const int64_t lim = (1ll << 31) - 4096;
  DFL_REQUIRE(xb < lim && wb < lim, "dfl_conv2d (bf16): tensors must stay below 2 GiB");
  p->x_bytes = (uint32_t)xb;
  p->w_bytes = (uint32_t)wb;
  if (a->x_mode != 0) {
    DFL_REQUIRE(a->x_mode == 1 && a->x2 != nullptr && a->in_shift == nullptr && a->ldx2 % 8 == 0 && aligned16(a->x2) &&
                    (a->in_scale == nullptr || aligned16(a->in_scale)),
                "dfl_conv2d (bf16): x_mode 1 needs x2 (16-byte aligned, ldx2 %% 8 == 0), coefficients in in_scale and no in_shift");
    const int64_t x2b = (((int64_t)a->N * a->Hin * a->Win - 1) * a->ldx2 + a->Cin) * 2;
    DFL_REQUIRE(x2b < lim, "dfl_conv2d (bf16): tensors must stay below 2 GiB");
    p->x2_bytes = (uint32_t)x2b;
  }
  p->xo_bytes = 0;
  if (a->x_out != nullptr) {
    DFL_REQUIRE(a->x_mode == 1 && a->KH == 3 && a->KW == 3 && a->stride == 1 && a->pad == 1 && a->Hout == a->Hin && a->Wout == a->Win,
                "dfl_conv2d (bf16): x_out goes with x_mode 1 of a 3x3 stride-1 pad-1 layer");
    DFL_REQUIRE(a->ldxo % 8 == 0 && a->ldxo >= a->Cin && aligned16(a->x_out) && a->x_out != a->x && a->x_out != a->x2,
                "dfl_conv2d (bf16): x_out must be 16-byte aligned with ldxo %% 8 == 0, ldxo >= Cin, and a tensor of its own");
    const int64_t xob = (((int64_t)a->N * a->Hin * a->Win - 1) * a->ldxo + a->Cin) * 2;
    DFL_REQUIRE(xob < lim, "dfl_conv2d (bf16): tensors must stay below 2 GiB");
    p->xo_bytes = (uint32_t)xob;
  }

  // the latency form (dfl_conv_args.latency_form: small problems of a batch-1 inference forward, convs_bf16.hip) plans itself
  if (t_force.tile < 0 && convs_eligible(*a, *p)) {
    convs_plan(*a, p, force_splits);
    return DFL_OK;
  }

  DFL_REQUIRE(a->out_scale == nullptr, "dfl_conv2d (bf16): out_scale / out_shift are implemented by the latency form only (dfl_conv_config tells)");
  // a forced geometry (dfl_conv_force_geometry: tuners, tests) or an entry of the tuning table (dfl_conv_tune_add) wins
  // over the cost model, as long as it is valid for this layer and agrees with the caller's K slices
  double best = 1e300;
  ConvP bestp = *p;
  int best_tile = -1;
  {
    ConvGeom g;
    const bool forced = t_force.tile >= 0;
    if (forced) g = t_force;
    if (forced || tune_lookup(*a, &g)) {
      if (g.tile >= 0 && g.tile < kNumTiles && (force_splits <= 0 || force_splits == g.splits)) {
        ConvP q = *p;
        double c;
        if (try_geometry(*a, &q, kTiles[g.tile], g.ipp, g.ph, g.pw, g.splits, &c) && q.splits == g.splits) {
          best = c;
          bestp = q;
          best_tile = g.tile;
        }
      }
      DFL_REQUIRE(!forced || best_tile >= 0, "dfl_conv2d (bf16): the forced geometry (tile %d, patch %dx%dx%d, %d slices) does not fit this layer",
                  g.tile, g.ipp, g.ph, g.pw, g.splits);
    }
  }
  if (best_tile < 0) {
    for_each_candidate(*a, *p, force_splits, false, [&](int ti, const ConvP& q, double c) {
      if (c < best) {
        best = c;
        bestp = q;
        best_tile = ti;
      }
    });
  }
  DFL_REQUIRE(best_tile >= 0, "dfl_conv2d (bf16): no patch geometry fits this layer (%dx%d, Cin %d, Ntot %d)", a->Hin, a->Win,
              a->Cin, a->Ntot);
  *p = bestp;
  p->tile = best_tile;
  static const bool dbg = getenv("DFL_CONVP_DEBUG") != nullptr;
  if (dbg && force_splits == 0)
    fprintf(stderr, "convp %dx%d Cin%d->%d k%d s%d: tile %d,%d,%d patch %dx%dx%d CK %d blocks %d splits %d lds %d cost %.0f\n", a->Hin, a->Win,
            a->Cin, a->Ntot, a->KH, a->stride, kTiles[best_tile].WM, kTiles[best_tile].WN, kTiles[best_tile].TM, p->IPP, p->PH, p->PW, p->CK,
            p->nblk, p->splits, p->lds_bytes, best);
  return DFL_OK;
}

// The geometry of an argument block is resolved ONCE: a recorded program replays the same blocks every step (97 convolutions
// per training step of the paper network), and the search above -- validation, the tuning-table scan under a mutex, for
// layers the table does not list up to ~1000 try_geometry calls -- is host time the GPU waits for on a slow host
// (8.4 ms per step observed where the kernels take 5.3).  Keyed by the whole argument block (addresses
// included: they select alignment-dependent paths and are copied into ConvP) and the requested K slices; a forced
// geometry (tuners, tests) bypasses it, a changed tuning table clears it.
struct PlanMemo { dfl_conv_args a; int force_splits; ConvP p; };
static std::mutex g_memo_mu;
static std::vector<PlanMemo> g_memo[64];
static inline unsigned memo_bucket(const dfl_conv_args* a, int force_splits) {
  uint64_t h = 1469598103934665603ull;
  const uint64_t* w = reinterpret_cast<const uint64_t*>(a);
  for (size_t i = 0; i < sizeof(dfl_conv_args) / 8; ++i) h = (h ^ w[i]) * 1099511628211ull;
  h ^= (uint64_t)force_splits;
  return (unsigned)((h ^ (h >> 29)) & 63u);
}
static void memo_clear() {
  std::lock_guard<std::mutex> lock(g_memo_mu);
  for (auto& b : g_memo) b.clear();
}

int convp_plan(const dfl_conv_args* a, ConvP* p, int force_splits) {
  static_assert(sizeof(dfl_conv_args) % 8 == 0, "hashed as 64-bit words");
  DFL_REQUIRE(a != nullptr && p != nullptr, "dfl_conv2d (bf16): null arguments");
  if (t_force.tile >= 0) return convp_plan_search(a, p, force_splits);
  const unsigned b = memo_bucket(a, force_splits);
  {
    std::lock_guard<std::mutex> lock(g_memo_mu);
    for (const PlanMemo& m : g_memo[b])
      if (m.force_splits == force_splits && memcmp(&m.a, a, sizeof(*a)) == 0) {
        *p = m.p;
        return DFL_OK;
      }
  }
  const int rc = convp_plan_search(a, p, force_splits);
  if (rc != DFL_OK) return rc;
  std::lock_guard<std::mutex> lock(g_memo_mu);
  if (g_memo[b].size() >= 256) g_memo[b].clear();          // bounded: 16 K argument blocks, then start over
  g_memo[b].push_back(PlanMemo{*a, force_splits, *p});
  return DFL_OK;
}

template <int WM, int WN, int TM, int TN, bool GA = false, int KS = 1>
static int convp_launch_t(const ConvP& p, hipStream_t s) {
  const bool aff = p.a.in_scale != nullptr || p.a.in_tot != nullptr;
  dim3 grid((unsigned)p.grid);
  size_t lds = (size_t)p.lds_bytes;
  constexpr int BN_ = WN * TN * 32;
  constexpr int NT_ = 256 * KS;
  constexpr int RPS_ = (NT_ / (BN_ / 8)) < WM * 32 ? (NT_ / (BN_ / 8)) : WM * 32;
  const size_t epi = (size_t)WM * 32 * (BN_ + 4) * sizeof(float);              // epilogue image (and the k-group exchange)
  const size_t red = (size_t)RPS_ * 2 * BN_ * sizeof(float);                   // statistics scratch
  if (lds < epi) lds = epi;
  if (lds < red) lds = red;
  ConvP pl = p;                                       // the "live" BatchNorm tables sit behind everything else in LDS
  pl.tab_off = (int)((lds + 15) / 16 * 16);
  lds = (size_t)pl.tab_off + (384 + 3 * BN_) * sizeof(float);
  const ConvP& p_ = pl;
  if constexpr (GA) {
    auto k = convp_kernel<WM, WN, TM, TN, 0, true>;
    DFL_LDS_OPT_IN(k, (int)kLdsHard, "dfl_conv2d (bf16)")
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, p_);
  } else if (p.a.x_mode != 0) {
    auto k = convp_kernel<WM, WN, TM, TN, 2, false, KS>;
    DFL_LDS_OPT_IN(k, (int)kLdsHard, "dfl_conv2d (bf16)")
    hipLaunchKernelGGL(k, grid, dim3(NT_), lds, s, p_);
  } else if (aff) {
    auto k = convp_kernel<WM, WN, TM, TN, 1, false, KS>;
    DFL_LDS_OPT_IN(k, (int)kLdsHard, "dfl_conv2d (bf16)")
    hipLaunchKernelGGL(k, grid, dim3(NT_), lds, s, p_);
  } else {
    auto k = convp_kernel<WM, WN, TM, TN, 0, false, KS>;
    DFL_LDS_OPT_IN(k, (int)kLdsHard, "dfl_conv2d (bf16)")
    hipLaunchKernelGGL(k, grid, dim3(NT_), lds, s, p_);
  }
  return check_launch("dfl_conv2d (bf16)");
}

int convp_launch(const ConvP& p, hipStream_t s) {
  int rc;
  switch (p.tile) {
    case CONVS_TILE: rc = convs_launch(p, s); break;
    case 0: rc = convp_launch_t<4, 1, 2, 1>(p, s); break;
    case 1: rc = convp_launch_t<4, 1, 1, 1>(p, s); break;
    case 2: rc = convp_launch_t<2, 2, 4, 1>(p, s); break;
    case 3: rc = convp_launch_t<2, 2, 3, 1>(p, s); break;
    case 4: rc = convp_launch_t<2, 2, 2, 1>(p, s); break;
    case 5: rc = convp_launch_t<1, 4, 2, 1>(p, s); break;
    case 6: rc = convp_launch_t<1, 4, 3, 1>(p, s); break;
    case 7: rc = convp_launch_t<1, 4, 4, 1>(p, s); break;
    case 8: rc = convp_launch_t<1, 4, 6, 1>(p, s); break;
    case 9: rc = convp_launch_t<1, 4, 9, 1>(p, s); break;
    case 10: rc = convp_launch_t<2, 2, 1, 1>(p, s); break;
    case 11: rc = convp_launch_t<1, 4, 1, 1>(p, s); break;
    case 12: rc = convp_launch_t<4, 1, 3, 1>(p, s); break;
    case 13: rc = convp_launch_t<4, 1, 4, 1>(p, s); break;
    case 14: rc = convp_launch_t<2, 2, 6, 1>(p, s); break;
    case 15: rc = convp_launch_t<2, 2, 2, 2>(p, s); break;
    case 16: rc = convp_launch_t<2, 2, 3, 2>(p, s); break;
    case 17: rc = convp_launch_t<2, 2, 4, 2>(p, s); break;
    case 18: rc = convp_launch_t<4, 1, 2, 2>(p, s); break;
    case 19: rc = convp_launch_t<4, 1, 3, 2>(p, s); break;
    case 20: rc = convp_launch_t<1, 4, 2, 2>(p, s); break;
    case 21: rc = convp_launch_t<1, 4, 3, 2>(p, s); break;
    case 22: rc = convp_launch_t<4, 1, 1, 1, true>(p, s); break;
    case 23: rc = convp_launch_t<4, 1, 2, 1, true>(p, s); break;
    case 24: rc = convp_launch_t<2, 2, 1, 1, true>(p, s); break;
    case 25: rc = convp_launch_t<2, 2, 2, 1, true>(p, s); break;
    case 26: rc = convp_launch_t<1, 4, 1, 1, true>(p, s); break;
    case 27: rc = convp_launch_t<1, 4, 2, 1, true>(p, s); break;
    case 28: rc = convp_launch_t<4, 1, 1, 2, true>(p, s); break;
    case 29: rc = convp_launch_t<2, 2, 1, 2, true>(p, s); break;
    case 30: rc = convp_launch_t<1, 4, 3, 1, false, 2>(p, s); break;
    case 31: rc = convp_launch_t<1, 4, 2, 1, false, 2>(p, s); break;
    case 32: rc = convp_launch_t<1, 4, 4, 1, false, 2>(p, s); break;
    case 33: rc = convp_launch_t<2, 2, 3, 1, false, 2>(p, s); break;
    case 34: rc = convp_launch_t<2, 2, 2, 1, false, 2>(p, s); break;
    case 35: rc = convp_launch_t<1, 4, 2, 2, false, 2>(p, s); break;
    case 36: rc = convp_launch_t<2, 2, 2, 2, false, 2>(p, s); break;
    case 37: rc = convp_launch_t<4, 1, 3, 1, false, 2>(p, s); break;
    case 38: rc = convp_launch_t<4, 1, 2, 1, false, 2>(p, s); break;
    default:
      DFL_REQUIRE(p.tile >= CONVQ_TILE && p.tile < kNumTiles, "dfl_conv2d (bf16): tile configuration %d", p.tile);
      if (p.tile >= CONVN_PERS_TILE) rc = convn_launch(p, p.tile - CONVN_PERS_TILE + 4, 1, s);
      else if (p.tile >= CONVN_TILE) rc = convn_launch(p, p.tile - CONVN_TILE, 0, s);
      else rc = convq_launch(p, (p.tile - CONVQ_TILE) % CONVQ_LAYOUTS, (p.tile - CONVQ_TILE) / CONVQ_LAYOUTS, s);
      break;
  }
  if (rc != DFL_OK || p.splits <= 1) return rc;
  int tx = 1;
  while (tx * 2 <= p.a.Ntot && tx * 2 <= FIN_TX) tx *= 2;
  const int nb = finish_rows_p(p.Mtot, p.a.Ntot);
  const int rpb = (int)ceil_div(p.Mtot, nb);
  dim3 grid((unsigned)nb, (unsigned)ceil_div(p.a.Ntot, tx));
  hipLaunchKernelGGL(convp_finish_kernel, grid, dim3(256), 0, s, p, tx, rpb);
  return check_launch("dfl_conv2d (bf16, split-K finish)");
}


// Candidates of the geometry search for one layer (tuners: tools/tune_convp.py): up to `max` rows of 5 integers
// (tile configuration, images per patch, patch height, patch width, K slices); returns the number of candidates.
int convp_candidates(const dfl_conv_args* a, int32_t* out, int max) {
  ConvP p;
  const int saved_tile = t_force.tile;
  t_force.tile = -1;
  const int rc = convp_plan(a, &p, 0);               // validates the arguments and fills the layer constants
  t_force.tile = saved_tile;
  if (rc != DFL_OK) return -1;
  ConvP base = p;
  int n = 0;
  for_each_candidate(*a, base, 0, true, [&](int ti, const ConvP& q, double) {
    if (n < max) {
      int32_t* o = out + 5 * n;
      o[0] = ti; o[1] = q.IPP; o[2] = q.PH; o[3] = q.PW; o[4] = q.splits;
    }
    ++n;
  });
  return n;
}

int convp_force(const int32_t* g) {
  if (g == nullptr) t_force.tile = -1;
  else t_force = ConvGeom{g[0], g[1], g[2], g[3], g[4]};
  return DFL_OK;
}

int convp_tune_add(const int32_t* key, const int32_t* g) {
  memo_clear();
  std::lock_guard<std::mutex> lock(g_tune_mu);
  if (key == nullptr) {
    g_tune.clear();
    return DFL_OK;
  }
  TuneEntry e;
  for (int i = 0; i < 10; ++i) e.key[i] = key[i];
  e.g = ConvGeom{g[0], g[1], g[2], g[3], g[4]};
  for (TuneEntry& o : g_tune)
    if (memcmp(o.key, e.key, sizeof(e.key)) == 0) {
      o = e;
      return DFL_OK;
    }
  g_tune.push_back(e);
  return DFL_OK;
}

}  // namespace dfl

// ---- C ABI of the geometry search (include/dfl_hip.h)
extern "C" int dfl_conv_candidates(const dfl_conv_args* a, int32_t* out, int32_t max_rows) {
  DFL_REQUIRE(a != nullptr && (out != nullptr || max_rows <= 0) && a->x_bf16, "dfl_conv_candidates: bf16 convolution arguments and an output table are required");
  const int n = dfl::convp_candidates(a, out, max_rows);
  return n < 0 ? (int)DFL_ERR_INVALID_ARG : n;
}

extern "C" int dfl_conv_force_geometry(const int32_t* geom) { return dfl::convp_force(geom); }

extern "C" int dfl_conv_tune_add(const int32_t* key, const int32_t* geom) {
  DFL_REQUIRE(key == nullptr || geom != nullptr, "dfl_conv_tune_add: a key needs a geometry");
  return dfl::convp_tune_add(key, geom);
}
