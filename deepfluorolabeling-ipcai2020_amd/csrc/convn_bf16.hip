// 3x3 convolution for bf16 tensors with FEW output columns (32 or 64): the two shallow levels of the network (round 6).
//
// Serves the 3x3 / stride 1 / pad 1 convolutions and data gradients whose output has 32 or 64 channels (reference:
// train_test_code/unet.py:211-222 and their autograd; levels 0 and 1 of the paper network: 192 x 192 x 32 and 96 x 96 x 64 at the
// 192 x 192 input, 19 launches of a training step).  Those layers move 75-150 MB for 11-22 GFLOP: the HBM stream is their floor, and
// convp / convq ran them at a third of it.  The counters of round 6 say why: 1500-2000 vector instructions per wave and patch of
// 96 pixels x 32 columns (17.7 us of vector issue per SIMD in a 41 us launch, 4.8 us of matrix instructions) -- index arithmetic of
// the staging loop, the accumulators' trip through LDS, a row loop with its run-time cases.  This form is built to issue few
// instructions per pixel:
//
//   * the matrix instruction is turned round: A = weights (32 output channels x 16 k), B = pixels (16 k x 32 pixels of ONE image
//     row), so a lane's 16 accumulator registers are 4 x 4 consecutive channels of one pixel.  The epilogue works on the registers
//     (bias, ReLU, round to bf16, statistics of the stored values), a v_permlane32_swap pairs the two half-waves' groups into 16
//     contiguous bytes, and the wave stores: no LDS image of the output, no barrier, no row loop;
//   * a wave owns R tiles stacked vertically (R image rows x 32 pixels): the fragment of image row y serves tile y at kernel row 1,
//     tile y - 1 at kernel row 2 and tile y + 1 at kernel row 0, so R + 2 fragment reads feed 3 R matrix instructions (convq: 3 per 3);
//   * a thread stages one 16-byte unit per image row at a fixed column: the global offset is `row base (scalar) + lane constant`,
//     the LDS offset `row x pitch (immediate) + lane constant`, the BatchNorm coefficients of its 8 channels sit in registers;
//   * weights come from L1 / L2 in the packed [k/16][n][16] layout through a register ring two groups ahead, as in convq;
//   * all input channels pass through ONE LDS image in blocks of 32 (53 KB for a 64 x 8 patch: three workgroups per CU hide each
//     other's staging); workgroup b works on patch (b mod 8) x npatch / 8 + b / 8: an XCD's L2 sees neighbouring patches' halos.
//
// Epilogue cases it does not take (K slices, `+ BN(add)`, accumulate, out_scale) stay with convp / convq.  Summation order: channel
// blocks, 16-channel chunks, kernel columns, kernel rows -- results agree with the other forms to fp32 rounding.
#include "common.h"
#include "convp.h"

namespace dfl {
namespace {

constexpr uint32_t NOOB = 0x80000000u;
typedef unsigned int nu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float n_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float n_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t n_pack(float a, float b) {
  const bf16x2_t h = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);   // round to nearest even (v_cvt_pk_bf16_f32)
  return __builtin_bit_cast(uint32_t, h);
}

struct NCfg { int WX, R; };
// layouts: waves side by side (32 pixels each) x rows per wave; the patch is 32 WX pixels wide and R (4 / WX) rows high
constexpr NCfg kN[CONVN_LAYOUTS] = {{2, 4}, {2, 6}, {1, 4}, {1, 6}, {2, 3}, {1, 3}};

#ifndef DFL_CONVN_WL
#define DFL_CONVN_WL 0       // several channel blocks: a block's weights through LDS and the next block requested ahead (0: the register ring; A/B builds)
#endif
#ifndef DFL_CONVN_OCC4
#define DFL_CONVN_OCC4 0
#endif
#ifndef DFL_CONVN_UGM
#define DFL_CONVN_UGM 10
#endif
// waves per SIMD the registers are budgeted for: three (168 registers) for layers of ONE channel block with one column tile and up
// to four rows per wave; two (256) otherwise -- several blocks keep the next block's units in registers through the k loop
constexpr int n_occ(int nct, int r, bool mb) { return (!mb || !DFL_CONVN_WL) && nct == 1 ? (r <= 3 && !mb && DFL_CONVN_OCC4 ? 4 : r <= 4 ? 3 : 2) : 2; }
// (two column tiles x six rows are 192 accumulator registers: not built; nor two column tiles x four rows for several channel blocks,
// where the next block's units would have to sit beside 128 accumulator registers)
constexpr bool n_inst(int nct, int r, bool mb) { return !(nct == 2 && (r > 4 || (mb && DFL_CONVN_WL != 0 && r >= 4))); }

// PERS (layers of ONE channel block and 32 columns; rows per wave <= 3): the workgroup is PERSISTENT -- it walks patches pr, pr + q_stride,
// ... of its XCD's run.  The layer's 18 KB of weights sit in LDS, so the k loop waits on LDS only and the NEXT patch's units, requested
// before the loop, are in flight through it (vector memory returns in order: behind a register ring they would be waited for at the
// first weight fragment); they are written into the image once every wave is through the current patch's fragments, and the epilogue
// of the current patch runs behind that.  The statistics stay in registers over all patches of the workgroup and cross lanes and waves
// once, at the end (rows of stat_partials: the workgroup's first patch carries its sum, its other patches zeros).
template <int CK, int NCT, int WX, int R, int AFF, bool MB, bool PERS>
__global__ void __launch_bounds__(256, PERS ? 2 : n_occ(NCT, R, MB)) convn_kernel(const ConvP p) {
  static_assert(!PERS || (!MB && NCT == 1 && R <= 3), "the persistent form: one channel block, 32 columns, three rows per wave");
  constexpr int NT = 256, WY = 4 / WX, PW = 32 * WX, PH = R * WY, IW = PW + 2, IH = PH + 2;
  constexpr int S = 2 * CK + 16;               // bytes per staged pixel: an odd multiple of 16, so the 32 pixels of a fragment read hit 16 bank quads twice
  constexpr int RPB = IW * S;                  // row pitch of the image
  constexpr int UPX = CK / 8;                  // 16-byte units per staged pixel
  constexpr int UR = PW * UPX;                 // units of the interior of one image row
  static_assert(UR <= NT && NT % UR == 0, "a pass of the workgroup's threads covers whole rows");
  constexpr int RPP = NT / UR;                 // image rows per pass
  static_assert(IH % RPP == 0, "passes cover the image exactly");
  constexpr int UI = IH / RPP;                 // interior units per thread
  constexpr int NUH = IH * 2 * UPX;            // units of the two halo columns
  static_assert(NUH <= NT, "one halo unit per thread");
  constexpr int CKC = CK / 16;                 // 16-channel chunks per block
  constexpr int NG = CKC * 3;                  // groups (chunk, kernel column) per block
  constexpr int RING = NCT == 1 ? 3 : 2;       // weight ring: groups in registers (RING - 1 ahead)
  static_assert(NG % RING == 0 && NG % 2 == 0, "ring and fragment slots are static");
  constexpr int NB = 32 * NCT;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const dfl_conv_args& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wx = wave % WX, wy = wave / WX;

  // ---- patch of this workgroup (an XCD = workgroups b = x mod 8 works through one eighth of the patches, in order)
  const int xcd_run = ((int)blockIdx.x & 7) * p.q_ngroups;
  int prun = (int)blockIdx.x >> 3;                  // patch of the XCD's run (PERS: advances by q_stride)
  if (xcd_run + prun >= p.npatch || prun >= p.q_ngroups) return;
  auto qdiv = [](int q, uint32_t m, int d) { return d == 1 ? q : (int)__umulhi((uint32_t)q, m); };
  const int per_img = p.npy * p.npx;
  // the patch being STAGED (PERS: one ahead of the patch the epilogue writes, o_* below)
  int pidx, img, gy0, gx0;
  auto patch_pos = [&](int pi) {
    pidx = pi;
    img = qdiv(pi, p.qm_perimg, per_img);
    const int prem = pi - img * per_img;
    const int ppy = qdiv(prem, p.qm_npx, p.npx), ppx = prem - ppy * p.npx;
    gy0 = ppy * PH;
    gx0 = ppx * PW;
  };
  patch_pos(xcd_run + prun);
  const int nblk = MB ? p.nblk : 1;                 // (MB false: the layer has 32 input channels)

  __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)p.w_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)((uint32_t)p.Mtot * (uint32_t)a.ldy * 2u), 0x00020000);
  __amdgpu_buffer_rsrc_t rsR = rsX, rsO = rsX;
  bool store_on = false;
  if constexpr (AFF == 2) {
    rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x2), 0, (int)p.x2_bytes, 0x00020000);
    store_on = a.x_out != nullptr;
    if (store_on) rsO = __builtin_amdgcn_make_buffer_rsrc(a.x_out, 0, (int)p.xo_bytes, 0x00020000);
  }

  // ---- weight ring.  Group gi of a block = (chunk gi / 3, kernel column gi % 3): its 3 kernel rows x NCT column tiles
  const uint32_t w_voff = (uint32_t)(li * 32 + lh * 16);
  const uint32_t nt32 = (uint32_t)a.Ntot * 32u;
  const uint32_t tapS = (uint32_t)(a.Cin >> 4) * nt32;
  // MB (several channel blocks): a block's weights -- 9 CKC pieces of NCT KB, piece (tap, chunk) -- pass through LDS beside the
  // image instead: the k loop then waits on LDS only, and the NEXT block's image and weights, requested before the loop, are in
  // flight through it (vector memory returns in order: a wait for a weight fragment would be a wait for the image).
  constexpr bool WL = (MB && DFL_CONVN_WL != 0) || PERS;
  constexpr int WRS = WL ? 2 : RING;
  nu32x4 wreg[WRS][3 * NCT];
  auto load_w = [&](const int gi, const uint32_t wb, const bool live) __attribute__((always_inline)) {
    const int c = gi / 3, dx = gi % 3;
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct) {
        if constexpr (WL) {
          wreg[gi % WRS][dy * NCT + ct] = *reinterpret_cast<const nu32x4*>(smem + p.tab_off - 9 * CKC * 1024 * NCT + ((dy * 3 + dx) * CKC + c) * 1024 * NCT + ct * 1024 + w_voff);
        } else {
          const uint32_t soff = wb + (uint32_t)(dy * 3 + dx) * tapS + (uint32_t)c * nt32 + (uint32_t)ct * 1024u;
          wreg[gi % WRS][dy * NCT + ct] = __builtin_amdgcn_raw_buffer_load_b128(rsW, live ? w_voff : NOOB, live ? soff : 0u, 0);
        }
      }
  };
  auto wbase = [&](int blk) { return (uint32_t)(blk * CKC) * nt32; };
  if constexpr (!WL) {
#pragma unroll
    for (int g = 0; g < RING - 1; ++g) load_w(g, 0u, true);
  }
  // (MB) weight unit j of this thread: 16 bytes at LDS offset (tid + 256 j) 16 of the block's 9 CKC NCT KB; piece (tid + 256 j) / (64 NCT)
  constexpr int WU = WL ? (9 * CKC * 64 * NCT + NT - 1) / NT : 1;
  constexpr bool WPRE = true;
  nu32x4 wu[WU];
  auto wstage_load = [&](int blk) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < WU; ++j) {
      const int u = tid + j * NT;
      const int pc = __builtin_amdgcn_readfirstlane(u / (64 * NCT));        // (wave-uniform: 64 NCT units per piece)
      const int t = pc / CKC, c = pc - t * CKC;
      const bool live = pc < 9 * CKC;
      wu[j] = __builtin_amdgcn_raw_buffer_load_b128(rsW, live ? (uint32_t)((u & (64 * NCT - 1)) * 16) : NOOB, live ? (uint32_t)t * tapS + (uint32_t)(blk * CKC + c) * nt32 : 0u, 0);
    }
  };
  auto wstage_store = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < WU; ++j) {
      const int u = tid + j * NT;
      if (u < 9 * CKC * 64 * NCT) *reinterpret_cast<nu32x4*>(smem + p.tab_off - 9 * CKC * 1024 * NCT + u * 16) = wu[j];
    }
  };

  // ---- staging.  Interior unit j of this thread: image row prow0 + j RPP, pixel px of the patch's columns, channel group cg.
  // Halo unit (threads below NUH): image row tid / (2 UPX), column gx0 - 1 or gx0 + PW.
  const int cg = tid & (UPX - 1);
  const int px = (tid & (UR - 1)) / UPX;
  const int prow0 = __builtin_amdgcn_readfirstlane(tid / UR);          // (a wave stays inside one row: UR >= 64)
  static_assert(UR >= 64, "wave-uniform image rows");
  const uint32_t lds_i = (uint32_t)((px + 1) * S + cg * 16);
  const int hrow = tid / (2 * UPX), hside = (tid / UPX) & 1;
  const uint32_t lds_h = (uint32_t)(hrow * RPB + (hside ? (PW + 1) * S : 0) + cg * 16);
  bool col_ok, h_ok;
  uint32_t x_voff, x2_voff, xo_voff, hpix;
  auto patch_lanes = [&]() {                        // this thread's offsets in the patch being staged
    col_ok = gx0 + px < a.Win;
    x_voff = col_ok ? (uint32_t)((gx0 + px) * a.ldx * 2 + cg * 16) : NOOB;
    x2_voff = col_ok ? (uint32_t)((gx0 + px) * a.ldx2 * 2 + cg * 16) : NOOB;
    xo_voff = col_ok ? (uint32_t)((gx0 + px) * a.ldxo * 2 + cg * 16) : NOOB;
    const int hgx = hside ? gx0 + PW : gx0 - 1, hgy = gy0 - 1 + hrow;
    h_ok = tid < NUH && (unsigned)hgx < (unsigned)a.Win && (unsigned)hgy < (unsigned)a.Hin;
    hpix = ((uint32_t)img * (uint32_t)a.Hin + (uint32_t)hgy) * (uint32_t)a.Win + (uint32_t)hgx;
  };
  patch_lanes();

  float* in_tab = reinterpret_cast<float*>(smem + p.tab_off);        // [3][Cin]: scale, shift (AFF 1) / A, B, C (AFF 2)
  float* col_tab = in_tab + 3 * a.Cin;                               // [NB]: bias
  float* red2 = col_tab + NB;                                        // [2][NB][4]: the statistics' partial sums

  struct Unit { nu32x4 v, v2; };
  float tA[8], tB[8], tC[8];                                         // coefficients of this thread's 8 channels of the current block
  auto transform = [&](const Unit& un, const bool ok) __attribute__((always_inline)) {
    nu32x4 w = un.v;
    if constexpr (AFF == 1) {                      // zero padding applies AFTER the BatchNorm affine: outside pixels stay 0
      if (ok) {
        w.x = n_pack(fmaf(n_lo(w.x), tA[0], tB[0]), fmaf(n_hi(w.x), tA[1], tB[1]));
        w.y = n_pack(fmaf(n_lo(w.y), tA[2], tB[2]), fmaf(n_hi(w.y), tA[3], tB[3]));
        w.z = n_pack(fmaf(n_lo(w.z), tA[4], tB[4]), fmaf(n_hi(w.z), tA[5], tB[5]));
        w.w = n_pack(fmaf(n_lo(w.w), tA[6], tB[6]), fmaf(n_hi(w.w), tA[7], tB[7]));
      }
    }
    if constexpr (AFF == 2) {                      // outside pixels were loaded as zeros: r = 0 there, the value stays 0
      const nu32x4 r = un.v2;
      auto brb = [](float dy, float rv, float A, float B, float Cc) { return rv > 0.f ? fmaf(A, dy, fmaf(B, rv, Cc)) : 0.f; };
      w.x = n_pack(brb(n_lo(w.x), n_lo(r.x), tA[0], tB[0], tC[0]), brb(n_hi(w.x), n_hi(r.x), tA[1], tB[1], tC[1]));
      w.y = n_pack(brb(n_lo(w.y), n_lo(r.y), tA[2], tB[2], tC[2]), brb(n_hi(w.y), n_hi(r.y), tA[3], tB[3], tC[3]));
      w.z = n_pack(brb(n_lo(w.z), n_lo(r.z), tA[4], tB[4], tC[4]), brb(n_hi(w.z), n_hi(r.z), tA[5], tB[5], tC[5]));
      w.w = n_pack(brb(n_lo(w.w), n_lo(r.w), tA[6], tB[6], tC[6]), brb(n_hi(w.w), n_hi(r.w), tA[7], tB[7], tC[7]));
    }
    return w;
  };
  auto load_tab = [&](int c0) __attribute__((always_inline)) {
    if constexpr (AFF != 0) {
      const float4 s0 = *reinterpret_cast<const float4*>(in_tab + c0 + cg * 8), s1 = *reinterpret_cast<const float4*>(in_tab + c0 + cg * 8 + 4);
      const float4 h0 = *reinterpret_cast<const float4*>(in_tab + a.Cin + c0 + cg * 8), h1 = *reinterpret_cast<const float4*>(in_tab + a.Cin + c0 + cg * 8 + 4);
      tA[0] = s0.x; tA[1] = s0.y; tA[2] = s0.z; tA[3] = s0.w; tA[4] = s1.x; tA[5] = s1.y; tA[6] = s1.z; tA[7] = s1.w;
      tB[0] = h0.x; tB[1] = h0.y; tB[2] = h0.z; tB[3] = h0.w; tB[4] = h1.x; tB[5] = h1.y; tB[6] = h1.z; tB[7] = h1.w;
      if constexpr (AFF == 2) {
        const float4 c0_ = *reinterpret_cast<const float4*>(in_tab + 2 * a.Cin + c0 + cg * 8), c1_ = *reinterpret_cast<const float4*>(in_tab + 2 * a.Cin + c0 + cg * 8 + 4);
        tC[0] = c0_.x; tC[1] = c0_.y; tC[2] = c0_.z; tC[3] = c0_.w; tC[4] = c1_.x; tC[5] = c1_.y; tC[6] = c1_.z; tC[7] = c1_.w;
      }
    }
  };
  // requests of a block's image (at most UG interior units per thread in flight at once behind the first round), then their transform
  // and LDS writes.  The first round of the FIRST block takes UG0 units; of a later block UGP units, requested before the previous
  // block's k loop and held in registers through it (as many as the register budget of two waves per SIMD leaves).
  constexpr int UPR_ = AFF == 2 ? 8 : 4;                            // registers per unit
  constexpr int UGM = AFF == 2 ? 5 : DFL_CONVN_UGM;
  constexpr int UG = (UI + (UI + UGM - 1) / UGM - 1) / ((UI + UGM - 1) / UGM);
  constexpr int P_AVAIL = PERS ? (256 - (16 * R + 8 * (R + 2) + 24 + 44 + 32)) / UPR_ - 1      // (the statistics' 32 registers; weights and tables are read again)
                               : (256 - (16 * R * NCT + 8 * (R + 2) + 24 * NCT + (WPRE ? 4 * WU : 0) + 44 + (AFF == 2 ? 24 : AFF == 1 ? 16 : 0))) / UPR_ - 1;   // (- 1: the halo unit)
  constexpr int UGP = !WL ? 0 : (P_AVAIL >= UI ? UI : (P_AVAIL > 0 ? P_AVAIL : 0));
  constexpr int UGX = UG > UGP ? UG : UGP;
  Unit ui[UGX], uh;
  auto stage_load = [&](int c0, const int j0, const int cnt) __attribute__((always_inline)) {
    const uint32_t cb = (uint32_t)(c0 * 2);
#pragma unroll
    for (int jj = 0; jj < cnt; ++jj) {
      const int j = j0 + jj;
      if (j >= UI) break;
      const int gy = gy0 - 1 + prow0 + j * RPP;
      const bool rok = (unsigned)gy < (unsigned)a.Hin;
      const uint32_t rowpix = ((uint32_t)img * (uint32_t)a.Hin + (uint32_t)gy) * (uint32_t)a.Win;
      ui[jj].v = __builtin_amdgcn_raw_buffer_load_b128(rsX, rok ? x_voff : NOOB, rok ? rowpix * (uint32_t)a.ldx * 2u + cb : 0u, 0);
      if constexpr (AFF == 2) ui[jj].v2 = __builtin_amdgcn_raw_buffer_load_b128(rsR, rok ? x2_voff : NOOB, rok ? rowpix * (uint32_t)a.ldx2 * 2u + cb : 0u, 0);
    }
    if (j0 == 0) {
      uh.v = __builtin_amdgcn_raw_buffer_load_b128(rsX, h_ok ? hpix * (uint32_t)a.ldx * 2u + cb + (uint32_t)(cg * 16) : NOOB, 0, 0);
      if constexpr (AFF == 2) uh.v2 = __builtin_amdgcn_raw_buffer_load_b128(rsR, h_ok ? hpix * (uint32_t)a.ldx2 * 2u + cb + (uint32_t)(cg * 16) : NOOB, 0, 0);
    }
  };
  auto stage_store = [&](int c0, const int j0, const int cnt) __attribute__((always_inline)) {
#pragma unroll
    for (int jj = 0; jj < cnt; ++jj) {
      const int j = j0 + jj;
      if (j >= UI) break;
      const int row = prow0 + j * RPP, gy = gy0 - 1 + row;
      const bool rok = (unsigned)gy < (unsigned)a.Hin;
      const nu32x4 w = transform(ui[jj], rok && col_ok);
      if constexpr (AFF == 2) {
        if (store_on && row >= 1 && row <= PH && rok) {            // x_out: the patch's own pixels (wave-uniform condition)
          const uint32_t rowpix = ((uint32_t)img * (uint32_t)a.Hin + (uint32_t)gy) * (uint32_t)a.Win;
          __builtin_amdgcn_raw_buffer_store_b128(w, rsO, xo_voff, rowpix * (uint32_t)a.ldxo * 2u + (uint32_t)(c0 * 2), 0);
        }
      }
      *reinterpret_cast<nu32x4*>(smem + lds_i + (uint32_t)(row * RPB)) = w;
    }
    if (j0 == 0 && tid < NUH) *reinterpret_cast<nu32x4*>(smem + lds_h) = transform(uh, h_ok);
  };
  auto stage_rest = [&](int c0, const int from) __attribute__((always_inline)) {      // the rounds behind the first
#pragma unroll
    for (int j0 = from; j0 < UI; j0 += UG) {
      stage_load(c0, j0, UG);
      stage_store(c0, j0, UG);
    }
  };

  // ---- first block: its loads go out first, the tables are derived while they fly
  stage_load(0, 0, UG);
  if constexpr (WL) wstage_load(0);
  if (tid < NB) {
    const int col = tid;
    col_tab[tid] = (a.bias != nullptr && col < a.Ntot) ? a.bias[col] : 0.f;
  }
  if constexpr (AFF == 1) {
    for (int ch = tid; ch < a.Cin; ch += NT) {
      float sc_, sh_;
      if (a.in_tot != nullptr) bn_live_affine(a.in_tot, a.in_gamma, a.in_beta, a.in_count, a.bn_eps, a.Cin, ch, &sc_, &sh_);
      else sc_ = a.in_scale[ch], sh_ = a.in_shift[ch];
      in_tab[ch] = sc_;
      in_tab[a.Cin + ch] = sh_;
    }
  }
  if constexpr (AFF == 2) {
    for (int ch = tid; ch < a.Cin; ch += NT) {
      float A = 1.f, B = 0.f, Cc = 0.f;
      if (a.in_tot != nullptr) bn_live_coef(a.in_tot, a.in_gamma, a.in_mean, a.in_invstd, a.in_count, a.Cin, ch, &A, &B, &Cc);
      else if (a.in_scale != nullptr) A = a.in_scale[ch], B = a.in_scale[a.Cin + ch], Cc = a.in_scale[2 * a.Cin + ch];
      in_tab[ch] = A;
      in_tab[a.Cin + ch] = B;
      in_tab[2 * a.Cin + ch] = Cc;
    }
  }
  __syncthreads();                                  // tables complete
  load_tab(0);
  stage_store(0, 0, UG);
  if constexpr (WL) wstage_store();
  stage_rest(0, UG);

  // ---- k loop
  // fragment of image row rr (0 .. R + 1 of this wave's rows), kernel column dx, chunk c: pixel wx 32 + li + dx, this lane's k half
  const uint32_t f_addr = (uint32_t)((wy * R) * RPB + (wx * 32 + li) * S + lh * 16);
  f32x16 acc[NCT][R];
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[ct][r][e] = 0.f;
  float s1[NCT][16], s2[NCT][16];                   // statistics of this lane's 16 channels (PERS: over all patches of the workgroup)
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      s1[ct][e] = 0.f;
      s2[ct][e] = 0.f;
    }
  const int first_pidx = pidx;
  bf16x8_t fr[2][R + 2];
  auto fetch_f = [&](const int gi) __attribute__((always_inline)) {
    const int c = gi / 3, dx = gi % 3;
#pragma unroll
    for (int rr = 0; rr < R + 2; ++rr)
      fr[gi % 2][rr] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const nu32x4*>(smem + f_addr + (uint32_t)(rr * RPB + dx * S + c * 32)));
  };
  for (;;) {                                        // (PERS: the patches of this workgroup; else one pass)
  const int o_pidx = pidx, o_img = img, o_gy0 = gy0, o_gx0 = gx0;     // the patch whose image is in LDS: the one the epilogue writes
  bool has_next = false;
  if constexpr (PERS) {
    prun += p.q_stride;
    has_next = prun < p.q_ngroups && xcd_run + prun < p.npatch;
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[0][r][e] = 0.f;
  }
  for (int blk = 0; blk < nblk; ++blk) {
    __syncthreads();                                // the image is complete
    const bool more = blk + 1 < nblk;
    if constexpr (PERS) {
      if (has_next) {                               // the next patch: in flight through this patch's k loop
        patch_pos(xcd_run + prun);
        patch_lanes();
        if constexpr (UGP > 0) stage_load(0, 0, UGP);
      }
      load_w(0, 0u, true);
    } else if constexpr (WL) {
      if (more) {                                   // the next block: in flight through this block's k loop
        if constexpr (WPRE) wstage_load(blk + 1);
        if constexpr (UGP > 0) stage_load((blk + 1) * CK, 0, UGP);
      }
      load_w(0, 0u, true);
    }
    const uint32_t wb = wbase(blk), wbn = wbase(blk + 1);
    fetch_f(0);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
      if constexpr (WL) {
        if (gi + 1 < NG) load_w(gi + 1, 0u, true);
      } else {
        const int g = gi + RING - 1;
        if (g < NG) load_w(g, wb, true);
        else load_w(g - NG, wbn, more);
      }
      if (gi + 1 < NG) fetch_f(gi + 1);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int ct = 0; ct < NCT; ++ct) {
          const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, wreg[gi % WRS][dy * NCT + ct]);
#pragma unroll
          for (int r = 0; r < R; ++r) acc[ct][r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, fr[gi % 2][r + dy], acc[ct][r], 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);            // keep the prefetch distance as written
    }
    if (MB && more) {
      if constexpr (!WPRE) wstage_load(blk + 1);
      if constexpr (UGP == 0) stage_load((blk + 1) * CK, 0, UG);
      load_tab((blk + 1) * CK);
      __syncthreads();                              // every wave is through this block's fragments
      stage_store((blk + 1) * CK, 0, UGP > 0 ? UGP : UG);
      if constexpr (WL) wstage_store();
      stage_rest((blk + 1) * CK, UGP > 0 ? UGP : UG);
    }
  }
  if constexpr (PERS) {
    __syncthreads();                                // every wave is through this patch's fragments
    if (has_next) {
      if constexpr (UGP == 0) stage_load(0, 0, UG);
      load_tab(0);
      stage_store(0, 0, UGP > 0 ? UGP : UG);
      stage_rest(0, UGP > 0 ? UGP : UG);
    }
  }

  // ================================================================== epilogue on the registers
  // acc[ct][r][e]: pixel (gy0 + wy R + r, gx0 + wx 32 + li), channel ct 32 + (e / 4) 8 + lh 4 + e % 4
  const bool do_stats = a.stat_partials != nullptr || a.stat_totals != nullptr;
  const int ogx = o_gx0 + wx * 32 + li;
  const bool ocol_ok = ogx < p.Wg;
  const uint32_t y_voff = ocol_ok ? (uint32_t)(ogx * a.ldy * 2 + lh * 16) : NOOB;
  const uint32_t so_voff = (uint32_t)(ogx * a.ldso * 2 + lh * 8);
  const unsigned short* sop = reinterpret_cast<const unsigned short*>(a.stat_other);
  const float relu_floor = a.relu ? 0.f : -__builtin_inff();
  const bool ragged = o_gx0 + PW > p.Wg;            // (workgroup-uniform)
#pragma unroll
  for (int ct = 0; ct < NCT; ++ct) {
    float bias[16];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float4 b = *reinterpret_cast<const float4*>(col_tab + ct * 32 + g4 * 8 + lh * 4);
      bias[g4 * 4 + 0] = b.x; bias[g4 * 4 + 1] = b.y; bias[g4 * 4 + 2] = b.z; bias[g4 * 4 + 3] = b.w;
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int gy = o_gy0 + wy * R + r;
      if (gy >= p.Hg) continue;                     // (wave-uniform)
      const uint32_t rowpix = ((uint32_t)o_img * (uint32_t)p.Hg + (uint32_t)gy) * (uint32_t)p.Wg;
      uint32_t pk[8];                               // pk[g4 * 2 + h]: channels g4 8 + lh 4 + 2 h, + 1
#pragma unroll
      for (int d = 0; d < 8; ++d) {
        const float v0 = fmaxf(acc[ct][r][2 * d] + bias[2 * d], relu_floor), v1 = fmaxf(acc[ct][r][2 * d + 1] + bias[2 * d + 1], relu_floor);
        pk[d] = n_pack(v0, v1);
      }
      if (do_stats) {                               // statistics of the values as stored (columns beyond a ragged edge: none)
        float vs[16];
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          vs[2 * d] = n_lo(pk[d]);
          vs[2 * d + 1] = n_hi(pk[d]);
        }
        if (ragged) {
#pragma unroll
          for (int e = 0; e < 16; ++e) vs[e] = ocol_ok ? vs[e] : 0.f;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) s1[ct][e] += vs[e];
        if (sop != nullptr) {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            uint2 o = make_uint2(0u, 0u);
            if (ocol_ok) o = *reinterpret_cast<const uint2*>(sop + ((size_t)rowpix * (size_t)a.ldso + (size_t)(ct * 32 + g4 * 8)) + (so_voff >> 1));
            s2[ct][g4 * 4 + 0] = fmaf(vs[g4 * 4 + 0], n_lo(o.x), s2[ct][g4 * 4 + 0]);
            s2[ct][g4 * 4 + 1] = fmaf(vs[g4 * 4 + 1], n_hi(o.x), s2[ct][g4 * 4 + 1]);
            s2[ct][g4 * 4 + 2] = fmaf(vs[g4 * 4 + 2], n_lo(o.y), s2[ct][g4 * 4 + 2]);
            s2[ct][g4 * 4 + 3] = fmaf(vs[g4 * 4 + 3], n_hi(o.y), s2[ct][g4 * 4 + 3]);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 16; ++e) s2[ct][e] = fmaf(vs[e], vs[e], s2[ct][e]);
        }
      }
      // pair the half-waves' channel groups: afterwards lanes 0-31 hold channels 0-7 (16-23) of their pixel, lanes 32-63 channels 8-15 (24-31)
#pragma unroll
      for (int gp = 0; gp < 2; ++gp) {
        const auto sw0 = __builtin_amdgcn_permlane32_swap(pk[gp * 4 + 0], pk[gp * 4 + 2], false, false);
        const auto sw1 = __builtin_amdgcn_permlane32_swap(pk[gp * 4 + 1], pk[gp * 4 + 3], false, false);
        nu32x4 w;
        w.x = sw0[0]; w.y = sw1[0]; w.z = sw0[1]; w.w = sw1[1];
        __builtin_amdgcn_raw_buffer_store_b128(w, rsY, y_voff, rowpix * (uint32_t)a.ldy * 2u + (uint32_t)((ct * 32 + gp * 16) * 2), 0);
      }
    }
  }
  if constexpr (PERS) {
    if (has_next) {                                 // (a later patch of this workgroup: its row of stat_partials carries nothing)
      if (a.stat_partials != nullptr && a.stat_totals == nullptr && o_pidx != first_pidx && tid < 2 * NB) a.stat_partials[(int64_t)o_pidx * 2 * a.Ntot + tid] = 0.f;
      continue;
    }
    if (a.stat_partials != nullptr && a.stat_totals == nullptr && o_pidx != first_pidx && tid < 2 * NB) a.stat_partials[(int64_t)o_pidx * 2 * a.Ntot + tid] = 0.f;
  }
  if (do_stats) {
    // per-column sums of this patch (PERS: of the workgroup's patches): lanes and waves through LDS in a fixed order -> one row of
    // stat_partials (rows = patches) or the layer's live totals
    float* red = reinterpret_cast<float*>(smem);   // [wave][ct][e][lane]
#pragma unroll
    for (int which = 0; which < 2; ++which) {
      __syncthreads();                              // the image (which = 0) / the previous pass's sums are no longer read
#pragma unroll
      for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
        for (int e = 0; e < 16; ++e) red[((wave * NCT + ct) * 16 + e) * 64 + lane] = which == 0 ? s1[ct][e] : s2[ct][e];
      __syncthreads();
      for (int idx = tid; idx < NB * 4; idx += NT) {
        const int part = idx & 3, col = idx >> 2;                  // part = wave
        const int ct = col >> 5, ch = col & 31;
        const int e = (ch >> 3) * 4 + (ch & 3), h = (ch >> 2) & 1;
        const float* src = red + ((part * NCT + ct) * 16 + e) * 64 + h * 32;
        float sum = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const float4 t = *reinterpret_cast<const float4*>(src + q * 4);
          sum += t.x; sum += t.y; sum += t.z; sum += t.w;
        }
        red2[(which * NB + col) * 4 + part] = sum;
      }
    }
    __syncthreads();
    for (int idx = tid; idx < 2 * NB; idx += NT) {
      const int which = idx / NB, n = idx - which * NB;
      if (n < a.Ntot) {
        const float4 t = *reinterpret_cast<const float4*>(red2 + idx * 4);
        const float sum = ((t.x + t.y) + t.z) + t.w;
        if (a.stat_totals != nullptr) bn_live_add(a.stat_totals, first_pidx, which, a.Ntot, n, sum);
        else a.stat_partials[((int64_t)first_pidx * 2 + which) * a.Ntot + n] = sum;
      }
    }
  }
  break;
  }
}

// LDS: the image, (several channel blocks) a block's weights, the tables
size_t n_tab_off(int layout, int cin, int ntot, int pers) {
  const NCfg c = kN[layout];
  const size_t img = (size_t)(c.R * (4 / c.WX) + 2) * (32 * c.WX + 2) * (2 * 32 + 16);
  return (img + 15) / 16 * 16 + ((cin > 32 && DFL_CONVN_WL != 0) || pers != 0 ? (size_t)18 * 1024 * (ntot / 32) : 0);
}

template <int NCT, int WX, int R, bool MB, bool PERS>
int convn_launch_t(const ConvP& p, int layout, hipStream_t s) {
  ConvP pl = p;
  pl.tab_off = (int)n_tab_off(layout, p.a.Cin, p.a.Ntot, PERS ? 1 : 0);
  const size_t lds = convn_lds_bytes(layout, p.a.Cin, p.a.Ntot, PERS ? 1 : 0);
  DFL_REQUIRE(lds <= 160 * 1024, "dfl_conv2d (bf16, narrow 3x3): %zu bytes of LDS", lds);
  const bool aff = p.a.in_scale != nullptr || p.a.in_tot != nullptr;
  dim3 grid((unsigned)p.grid);
#define DFL_CN_LAUNCH(AFF_)                                                                                                   \
  {                                                                                                                             \
    auto k = convn_kernel<32, NCT, WX, R, AFF_, MB, PERS>;                                                                      \
    DFL_LDS_OPT_IN(k, 160 * 1024, "dfl_conv2d (bf16, narrow 3x3)") \
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, pl);                                                                         \
  }
  if (p.a.x_mode != 0) DFL_CN_LAUNCH(2)
  else if (aff) DFL_CN_LAUNCH(1)
  else DFL_CN_LAUNCH(0)
#undef DFL_CN_LAUNCH
  return check_launch("dfl_conv2d (bf16, narrow 3x3)");
}

template <int WX, int R>
int convn_launch_n(const ConvP& p, int layout, int pers, hipStream_t s) {
  if (pers != 0) {
    if constexpr (R <= 3) {
      if (p.a.Ntot == 32 && p.nblk == 1) return convn_launch_t<1, WX, R, false, true>(p, layout, s);
    }
    set_error("dfl_conv2d (bf16, narrow 3x3): the persistent form of layout %d takes 32 -> 32 layers", layout);
    return DFL_ERR_INVALID_ARG;
  }
  if (p.a.Ntot == 32) return p.nblk > 1 ? convn_launch_t<1, WX, R, true, false>(p, layout, s) : convn_launch_t<1, WX, R, false, false>(p, layout, s);
  if (p.nblk > 1) {
    if constexpr (n_inst(2, R, true)) return convn_launch_t<2, WX, R, true, false>(p, layout, s);
  } else {
    if constexpr (n_inst(2, R, false)) return convn_launch_t<2, WX, R, false, false>(p, layout, s);
  }
  set_error("dfl_conv2d (bf16, narrow 3x3): layout %d is not built for 64 columns and %d input channels", layout, p.a.Cin);
  return DFL_ERR_INVALID_ARG;
}

}  // namespace

// The layers this form takes (the caller has validated the argument block as convp_plan_search does)
bool convn_shape_ok(const dfl_conv_args& a) {
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.scatter2x2 == 0 && a.Cin % 32 == 0 && a.Cin <= 256 &&
         (a.Ntot == 32 || a.Ntot == 64) && a.Hout == a.Hin && a.Wout == a.Win && a.out_scale == nullptr && a.add == nullptr &&
         a.accumulate == 0 && (a.x_mode == 0 || (a.x2 != nullptr && !(a.Cin > 32 && a.Ntot == 64)));
  // (the two-tensor operand of a 64-column layer with several channel blocks -- 16 registers per staged unit beside 96 accumulator
  // registers -- measured slower than convp: 29 against 27 us for the 96 x 96 64 -> 64 data gradient)
}

bool convn_layout_ok(int layout, int ntot, int cin) { return layout >= 0 && layout < CONVN_LAYOUTS && n_inst(ntot / 32, kN[layout].R, cin > 32); }

void convn_patch(int layout, int* ph, int* pw) {
  *ph = kN[layout].R * (4 / kN[layout].WX);
  *pw = 32 * kN[layout].WX;
}

bool convn_pers_ok(int layout, const dfl_conv_args& a) { return layout >= 0 && layout < CONVN_LAYOUTS && kN[layout].R <= 3 && a.Cin == 32 && a.Ntot == 32; }

size_t convn_lds_bytes(int layout, int cin, int ntot, int pers) {
  size_t img = n_tab_off(layout, cin, ntot, pers);
  const size_t red = (size_t)4 * (ntot / 32) * 16 * 64 * sizeof(float);      // the statistics' pass through LDS
  if (img < red) img = red;
  return img + (size_t)(3 * cin + ntot + 2 * ntot * 4) * sizeof(float);
}

int convn_launch(const ConvP& p, int layout, int pers, hipStream_t s) {
  DFL_REQUIRE(layout >= 0 && layout < CONVN_LAYOUTS && p.CK == 32 && p.splits == 1, "dfl_conv2d (bf16, narrow 3x3): layout %d, %d resident channels, %d K slices", layout, p.CK, p.splits);
  switch (layout) {
    case 0: return convn_launch_n<2, 4>(p, layout, pers, s);
    case 1: return convn_launch_n<2, 6>(p, layout, pers, s);
    case 2: return convn_launch_n<1, 4>(p, layout, pers, s);
    case 3: return convn_launch_n<1, 6>(p, layout, pers, s);
    case 4: return convn_launch_n<2, 3>(p, layout, pers, s);
    default: return convn_launch_n<1, 3>(p, layout, pers, s);
  }
}

}  // namespace dfl
