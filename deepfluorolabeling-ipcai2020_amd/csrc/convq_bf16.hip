// Deep-level 3x3 convolution for bf16 tensors with a fully unrolled, compile-time k loop (round 6).
//
// Serves the 3x3 / stride 1 / pad 1 convolutions and data gradients of the levels with >= 64 input channels (reference:
// train_test_code/unet.py:211-222 and their autograd) -- the layers convp_kernel<1,4,3,1> ran at 12-30 % of the matrix rate.
// Same decomposition as convp_bf16.hip (a workgroup owns a patch of output pixels x 128 output columns, the patch with its
// halo lives in LDS as [pixel][CK] bf16, weights come straight from global memory in the packed [k/16][n][16] layout), same
// argument block, same epilogue semantics, same rounding places.  What differs is everything around the matrix instruction:
//
//   * the patch is ALWAYS 8 x 12 output pixels (10 x 14 staged pixels) and the resident channel block CK is a template
//     parameter, so every LDS fragment address of the k loop is `lane base + immediate` and the loop over (tap, 16-channel chunk)
//     is straight-line code: per k-step 3 matrix instructions, 3 ds_read_b128, 1 buffer load and one scalar add -- convp's loop
//     spent 12-25 scalar instructions per k-step on its run-time cursors, more than one wave can issue beside the matrix pipe;
//   * two LDS images: the next channel block is staged (global -> registers -> BatchNorm affine / BatchNorm + ReLU backward ->
//     LDS) BETWEEN the k-steps of the current one, in three chunks, so that only the first block's staging is exposed;
//   * A fragments run two k-steps ahead in three register sets, the weight ring (three groups of 3-4 k-steps) two groups ahead
//     and across block boundaries;
//   * KS = 2: 512 threads, the two k-groups take the two halves of a block's 16-channel chunks (two waves per SIMD cover each
//     other's latencies when a layer has one workgroup per CU); both groups drop their accumulators into LDS and ALL threads
//     run the row epilogue on the sum -- one pass, one barrier, instead of a reduction pass plus three epilogue passes;
//   * the live-BatchNorm tables of ALL channels of the workgroup's K slice are derived once, while the first image's loads fly.
//
// Summation order differs from convp's (chunks inside taps inside channel blocks, k-groups by chunk halves): results agree
// to fp32 rounding, as between two convp geometries.
#include "common.h"
#include "convp.h"

namespace dfl {
namespace {

constexpr uint32_t QOOB = 0x80000000u;
typedef unsigned int qu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float q_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float q_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ uint32_t q_pack(float a, float b) {
  const bf16x2_t h = __builtin_convertvector((f32x2_t){a, b}, bf16x2_t);   // round to nearest even (v_cvt_pk_bf16_f32)
  return __builtin_bit_cast(uint32_t, h);
}
__device__ __forceinline__ void q_unpack(const qu32x4 w, float* f) {
  f[0] = q_lo(w.x); f[1] = q_hi(w.x); f[2] = q_lo(w.y); f[3] = q_hi(w.y);
  f[4] = q_lo(w.z); f[5] = q_hi(w.z); f[6] = q_lo(w.w); f[7] = q_hi(w.w);
}

constexpr int QPW = 12, QIW = 14, QTM = 3;   // patch width, staged width, 32-row tiles per wave
constexpr int q_row_pitch(int ck) {
  int r = QIW * (2 * ck + 16) / 16;
  while (r % 16 != 12) ++r;
  return r * 16;
}
// Distance of the two image buffers.  One patch per workgroup: the image (the epilogue starts at 0 and may run over both).  Persistent
// forms: a buffer also has to hold the epilogue's pass image and the statistics scratch, because the other one holds the next patch.
constexpr int q_buf_bytes(int ck, int wm, int wn, int ks, int pers) {
  const int img = (8 * wm + 2) * q_row_pitch(ck);
  if (pers == 0) return img;
  const int nt = 64 * wm * wn * ks, bn = 32 * wn;
  const int epi = ks * wm * 32 * (bn + 4) * 4, red = (nt / (bn / 8)) * 2 * bn * 4;
  int b = img > epi ? img : epi;
  b = b > red ? b : red;
  return (b + 15) / 16 * 16;
}

// WM x WN x KS waves (4 or 8): WN waves side by side take 32 output columns each (the workgroup's column tile is 32 WN wide), WM
// waves take 8 patch rows = 96 pixels each (the patch is 8 WM x 12 pixels; layers whose K is short want many pixels per workgroup:
// a workgroup's life is mostly fill and drain there), KS k-groups split a block's 16-channel chunks.
// PERS: 0 = one patch per workgroup.  1, 2 = PERSISTENT: a workgroup walks patches pidx, pidx + q_ngroups, ... of its (column tile, K
// slice); the next patch's first image is staged while the current patch computes -- 1: chunk by chunk between the k-steps of the
// current patch's last block; 2 (patches of ONE channel block: K is short, the k loop shorter than a memory round trip): its loads are
// requested before the PREVIOUS patch's epilogue and fly through that epilogue and the whole k loop of the current patch, so a CU
// has a patch's worth of requests in flight all the time -- and the epilogue runs in the image the k loop has just finished with.
template <int CK, int WM, int WN, int KS, int AFF, int PERS>
__global__ void __launch_bounds__(64 * WM * WN * KS, 2) convq_kernel(const ConvP p) {
  constexpr int NT = 64 * WM * WN * KS;
  static_assert(NT == 256 || NT == 512, "four or eight waves");
  constexpr int QBN = 32 * WN, QEP = QBN + 4;          // columns per workgroup, row pitch (floats) of the epilogue image
  constexpr int QPH = 8 * WM, QIH = QPH + 2, QNPIX = QIH * QIW;
  constexpr int S = 2 * CK + 16;             // bytes per staged pixel (odd multiple of 16: consecutive pixels on different banks)
  // Row pitch of the image: with pixel (y, x) at y * RPB + x * S a ds_read_b128 lane group -- 16 lanes = pixels q, q + 1, ... of the
  // row-major patch, which wrap into the next patch row -- touches bank quad (y * RPB / 16 + x) mod 16; RPB / 16 = 12 (mod 16) is the
  // residue for which the four hardware lane groups of all three tile rows hit 16 different quads (searched exhaustively)
  constexpr int RPB = q_row_pitch(CK);
  static_assert(RPB >= QIW * S && (RPB / 16) % 16 == 12, "conflict-free row pitch");
  constexpr int CKC = CK / 16, CCL = CKC / KS, STEPS = 9 * CCL;   // 16-channel chunks per block / per k-group; k-steps per block and k-group
  constexpr int GS = (STEPS % 12 == 0) ? 4 : 3;  // k-steps per weight-ring group
  constexpr int NG = STEPS / GS;
  static_assert(CCL >= 1 && STEPS % GS == 0 && NG % 3 == 0 && STEPS % 3 == 0, "ring and fragment slots are static");
  constexpr int UPX = CK / 8;                // 16-byte units per staged pixel
  constexpr int NU = QNPIX * UPX;
  constexpr int U = (NU + NT - 1) / NT;      // units per thread and image
  constexpr int CHU = (U + 2) / 3;           // units per staging chunk (at most three chunks per block)
  constexpr int NCH = (U + CHU - 1) / CHU;
  static_assert(NT % UPX == 0, "a thread keeps its channel group");
  // epilogue: EPP passes over the wave's three tile rows.  One pass where the image may spread over all of the workgroup's LDS
  // (128-column forms, one patch per workgroup); three where it must stay small (narrow forms: three workgroups per CU) or inside
  // ONE image buffer (persistent forms: the other buffer already holds the next patch)
  constexpr int EPP = (WN == 4 && PERS == 0) ? 1 : 3;
  constexpr int TPP = QTM / EPP;                    // tile rows per pass
  constexpr int RW = 32 * TPP;                      // image rows per wave and pass
  constexpr int RI = WM * RW;                       // image rows per k-group and pass
  constexpr int UPR = QBN / 8;                      // 8-column units per row
  constexpr int RPS = NT / UPR;                     // rows per step of the workgroup's threads
  constexpr int BUF = q_buf_bytes(CK, WM, WN, KS, PERS);    // distance of the two images
  constexpr int ROWS_UNROLL = PERS == 2 ? 1 : 8;            // (PERS 2 holds a whole image in registers across the epilogue's row loop)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const dfl_conv_args& a = p.a;
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // wave-uniform (scalar registers): the k loop's addresses depend on them
  const int wn = wave % WN, wm = (wave / WN) % WM, kg = wave / (WN * WM);

  // ---- which patch(es), column tile and K slice.  One patch per workgroup: convp_kernel's map (weight-heavy layers keep a (tile,
  // slice) pair on one XCD).  Divisions by multiply-high with the host's magic numbers (the grid stays below 65536 workgroups).
  auto qdiv = [](int q, uint32_t m, int d) { return d == 1 ? q : (int)__umulhi((uint32_t)q, m); };
  int pidx, btile, bslice;
  {
    const int b = blockIdx.x;
    if (PERS != 0) {
      const int r = qdiv(b, p.qm_npatch, p.q_ngroups);            // (qm_npatch: magic of q_ngroups here)
      pidx = b - r * p.q_ngroups;
      bslice = qdiv(r, p.qm_ntiles, p.ntiles);
      btile = r - bslice * p.ntiles;
    } else if (p.xcd_mode == 0) {
      const int r = qdiv(b, p.qm_npatch, p.npatch);
      pidx = b - r * p.npatch;
      bslice = qdiv(r, p.qm_ntiles, p.ntiles);
      btile = r - bslice * p.ntiles;
    } else {
      const int x = b & 7, r = b >> 3;
      int s;
      if (p.xcd_mode == 1) {
        const int rq = qdiv(r, p.qm_npatch, p.npatch);
        s = x + 8 * rq;
        pidx = r - rq * p.npatch;
      } else {
        const int pairs = p.ntiles * p.splits;
        s = x % pairs;
        pidx = x / pairs + (8 / pairs) * r;
      }
      if (s >= p.ntiles * p.splits || pidx >= p.npatch) return;
      bslice = qdiv(s, p.qm_ntiles, p.ntiles);
      btile = s - bslice * p.ntiles;
    }
  }
  const int pstride = PERS != 0 ? p.q_ngroups : 0x40000000;
  const int per_img = p.npy * p.npx;
  struct Pos { int gy0, gx0, img; };
  auto pos_of = [&](int pi) {
    const int img = qdiv(pi, p.qm_perimg, per_img), pr = pi - img * per_img;
    const int ppy = qdiv(pr, p.qm_npx, p.npx), ppx = pr - ppy * p.npx;
    return Pos{ppy * QPH, ppx * QPW, img};
  };
  const int n0 = btile * QBN;
  const int blk_begin = bslice * p.blk_per_slice;
  const int blk_end = min(blk_begin + p.blk_per_slice, p.nblk);
  const int CKS = p.blk_per_slice * CK;            // channels of this K slice

  __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (int)p.x_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.w), 0, (int)p.w_bytes, 0x00020000);
  __amdgpu_buffer_rsrc_t rsR = rsX, rsO = rsX;
  bool store_on = false;
  if constexpr (AFF == 2) {
    rsR = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x2), 0, (int)p.x2_bytes, 0x00020000);
    store_on = a.x_out != nullptr && btile == 0;
    if (store_on) rsO = __builtin_amdgcn_make_buffer_rsrc(a.x_out, 0, (int)p.xo_bytes, 0x00020000);
  }

  // ---- weight ring: group g of a block = k-steps g*GS .. g*GS + GS - 1 of this k-group; step s = (tap s / CCL, chunk s % CCL)
  const int ncol = n0 + wn * 32 + li;
  const uint32_t b_voff = ncol < a.Ntot ? (uint32_t)(ncol * 32 + lh * 16) : QOOB;
  const uint32_t nt32 = (uint32_t)a.Ntot * 32u;
  const uint32_t tapS = (uint32_t)(a.Cin >> 4) * nt32;
  qu32x4 breg[3][GS];
  auto load_group = [&](const int g, const uint32_t wb, const bool live) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < GS; ++e) {
      const int s = g * GS + e, t = s / CCL, c = s % CCL;
      const uint32_t soff = wb + (uint32_t)t * tapS + (uint32_t)c * nt32;
      breg[g % 3][e] = __builtin_amdgcn_raw_buffer_load_b128(rsW, live ? b_voff : QOOB, live ? soff : 0u, 0);
    }
  };
  auto wbase = [&](int blk) { return (uint32_t)(blk * CKC + kg * CCL) * nt32; };
  load_group(0, wbase(blk_begin), true);
  load_group(1, wbase(blk_begin), true);

  // ---- staging: unit j of this thread = 16 bytes (8 channels, group cg) of staged pixel tid / UPX + j * (NT / UPX)
  const int cg = tid & (UPX - 1);
  const int pix0 = tid / UPX;
  float* col_tab = reinterpret_cast<float*>(smem + p.tab_off);   // [3][QBN]: bias, scale and shift of "+ BN(add)" of the workgroup's columns
  float* in_tab = col_tab + 3 * QBN;                              // [3][CKS]: scale, shift (AFF 1) / A, B, C (AFF 2) of the slice's channels
  struct Unit { qu32x4 v, v2; };
  auto unit_pix = [&](int j) { return pix0 + j * (NT / UPX); };
  auto unit_load = [&](int j, const Pos& ps, int c0, bool live, Unit* un) __attribute__((always_inline)) {
    const int pix = unit_pix(j);
    const int iy = pix / QIW, ix = pix - iy * QIW;
    const int gy = ps.gy0 - 1 + iy, gx = ps.gx0 - 1 + ix;
    const bool ok = live && pix < QNPIX && (unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win;
    const uint32_t pixel = ((uint32_t)ps.img * (uint32_t)a.Hin + (uint32_t)gy) * (uint32_t)a.Win + (uint32_t)gx;
    const uint32_t cb = (uint32_t)((c0 + cg * 8) * 2);
    un->v = __builtin_amdgcn_raw_buffer_load_b128(rsX, ok ? pixel * (uint32_t)a.ldx * 2u + cb : QOOB, 0, 0);
    if constexpr (AFF == 2) un->v2 = __builtin_amdgcn_raw_buffer_load_b128(rsR, ok ? pixel * (uint32_t)a.ldx2 * 2u + cb : QOOB, 0, 0);
  };
  // (the decode is repeated here instead of carried in registers: a dozen integer instructions per unit against 6-10 live registers)
  auto unit_store = [&](int j, const Pos& ps, int c0, int crel, uint32_t buf, const Unit& un) __attribute__((always_inline)) {
    const int pix = unit_pix(j);
    if (pix >= QNPIX) return;
    const int iy = pix / QIW, ix = pix - iy * QIW;
    const int gy = ps.gy0 - 1 + iy, gx = ps.gx0 - 1 + ix;
    const bool ok = (unsigned)gy < (unsigned)a.Hin && (unsigned)gx < (unsigned)a.Win;
    qu32x4 w = un.v;
    if constexpr (AFF == 1) {                      // zero padding applies AFTER the BatchNorm affine: outside pixels stay 0
      if (ok) {
        const float4 s0 = *reinterpret_cast<const float4*>(in_tab + crel + cg * 8), s1 = *reinterpret_cast<const float4*>(in_tab + crel + cg * 8 + 4);
        const float4 h0 = *reinterpret_cast<const float4*>(in_tab + CKS + crel + cg * 8), h1 = *reinterpret_cast<const float4*>(in_tab + CKS + crel + cg * 8 + 4);
        w.x = q_pack(fmaf(q_lo(w.x), s0.x, h0.x), fmaf(q_hi(w.x), s0.y, h0.y));
        w.y = q_pack(fmaf(q_lo(w.y), s0.z, h0.z), fmaf(q_hi(w.y), s0.w, h0.w));
        w.z = q_pack(fmaf(q_lo(w.z), s1.x, h1.x), fmaf(q_hi(w.z), s1.y, h1.y));
        w.w = q_pack(fmaf(q_lo(w.w), s1.z, h1.z), fmaf(q_hi(w.w), s1.w, h1.w));
      }
    }
    if constexpr (AFF == 2) {                      // outside pixels were loaded as zeros: r = 0 there, the value stays 0
      const qu32x4 r = un.v2;
      const float4 A0 = *reinterpret_cast<const float4*>(in_tab + crel + cg * 8), A1 = *reinterpret_cast<const float4*>(in_tab + crel + cg * 8 + 4);
      const float4 B0 = *reinterpret_cast<const float4*>(in_tab + CKS + crel + cg * 8), B1 = *reinterpret_cast<const float4*>(in_tab + CKS + crel + cg * 8 + 4);
      const float4 C0 = *reinterpret_cast<const float4*>(in_tab + 2 * CKS + crel + cg * 8), C1 = *reinterpret_cast<const float4*>(in_tab + 2 * CKS + crel + cg * 8 + 4);
      auto brb = [](float dy, float rv, float A, float B, float Cc) { return rv > 0.f ? fmaf(A, dy, fmaf(B, rv, Cc)) : 0.f; };
      w.x = q_pack(brb(q_lo(w.x), q_lo(r.x), A0.x, B0.x, C0.x), brb(q_hi(w.x), q_hi(r.x), A0.y, B0.y, C0.y));
      w.y = q_pack(brb(q_lo(w.y), q_lo(r.y), A0.z, B0.z, C0.z), brb(q_hi(w.y), q_hi(r.y), A0.w, B0.w, C0.w));
      w.z = q_pack(brb(q_lo(w.z), q_lo(r.z), A1.x, B1.x, C1.x), brb(q_hi(w.z), q_hi(r.z), A1.y, B1.y, C1.y));
      w.w = q_pack(brb(q_lo(w.w), q_lo(r.w), A1.z, B1.z, C1.z), brb(q_hi(w.w), q_hi(r.w), A1.w, B1.w, C1.w));
      if (store_on) {                              // x_out: the interior of the patch, this slice's channels
        const bool own = ok && (unsigned)(iy - 1) < (unsigned)QPH && (unsigned)(ix - 1) < (unsigned)QPW;
        const uint32_t pixel = ((uint32_t)ps.img * (uint32_t)a.Hin + (uint32_t)gy) * (uint32_t)a.Win + (uint32_t)gx;
        __builtin_amdgcn_raw_buffer_store_b128(w, rsO, own ? pixel * (uint32_t)a.ldxo * 2u + (uint32_t)((c0 + cg * 8) * 2) : QOOB, 0, 0);
      }
    }
    *reinterpret_cast<qu32x4*>(smem + buf + (uint32_t)(iy * RPB + ix * S) + (uint32_t)cg * 16u) = w;
  };

  // ---- first image: its loads go out first, the tables are derived while they fly
  Pos pos = pos_of(pidx);
  {
    constexpr int UG = AFF == 2 ? 6 : 10;         // units in flight per thread (registers: 4 or 8 per unit); larger images take several rounds
    constexpr int UG0 = U < UG ? U : UG;
    Unit un[UG0];
#pragma unroll
    for (int j = 0; j < UG0; ++j) unit_load(j, pos, blk_begin * CK, true, &un[j]);
    // the epilogue's per-column constants: fetched here, behind the image loads, instead of in front of the row loop (where the
    // whole workgroup waited 1.4-2 us for them: phase clocks of round 6)
    if (tid < QBN) {                                 // (QBN <= 128 < NT)
      const int col = n0 + tid;
      const bool ok = col < a.Ntot;
      float b_ = 0.f, sc_ = 1.f, sh_ = 0.f;
      if (a.bias != nullptr && ok && p.splits <= 1) b_ = a.bias[col];
      if (a.add != nullptr && ok) {
        if (a.add_scale != nullptr) sc_ = a.add_scale[col], sh_ = a.add_shift[col];
        else if (a.add_tot != nullptr) bn_live_affine(a.add_tot, a.add_gamma, a.add_beta, a.add_count, a.bn_eps, a.Ntot, col, &sc_, &sh_);
      }
      col_tab[tid] = b_;
      col_tab[QBN + tid] = sc_;
      col_tab[2 * QBN + tid] = sh_;
    }
    if constexpr (AFF == 1) {
      for (int c = tid; c < CKS; c += NT) {
        const int ch = blk_begin * CK + c;
        float sc_, sh_;
        if (a.in_tot != nullptr) bn_live_affine(a.in_tot, a.in_gamma, a.in_beta, a.in_count, a.bn_eps, a.Cin, ch, &sc_, &sh_);
        else sc_ = a.in_scale[ch], sh_ = a.in_shift[ch];
        in_tab[c] = sc_;
        in_tab[CKS + c] = sh_;
      }
    }
    if constexpr (AFF == 2) {
      for (int c = tid; c < CKS; c += NT) {
        const int ch = blk_begin * CK + c;
        float A = 1.f, B = 0.f, Cc = 0.f;
        if (a.in_tot != nullptr) bn_live_coef(a.in_tot, a.in_gamma, a.in_mean, a.in_invstd, a.in_count, a.Cin, ch, &A, &B, &Cc);
        else if (a.in_scale != nullptr) A = a.in_scale[ch], B = a.in_scale[a.Cin + ch], Cc = a.in_scale[2 * a.Cin + ch];
        in_tab[c] = A;
        in_tab[CKS + c] = B;
        in_tab[2 * CKS + c] = Cc;
      }
    }
    if (AFF != 0) __syncthreads();
#pragma unroll
    for (int j = 0; j < UG0; ++j) unit_store(j, pos, blk_begin * CK, 0, 0u, un[j]);
#pragma unroll
    for (int j0 = UG0; j0 < U; j0 += UG0) {
#pragma unroll
      for (int j = 0; j < UG0; ++j)
        if (j0 + j < U) unit_load(j0 + j, pos, blk_begin * CK, true, &un[j]);
#pragma unroll
      for (int j = 0; j < UG0; ++j)
        if (j0 + j < U) unit_store(j0 + j, pos, blk_begin * CK, 0, 0u, un[j]);
    }
  }
  // PERS 2: the registers the next patch waits in (requested one patch ahead: here for the second patch)
  Unit deep[PERS == 2 ? U : 1];
  if constexpr (PERS == 2) {
    const bool live = pidx + pstride < p.npatch;
    const Pos pn = pos_of(live ? pidx + pstride : pidx);
#pragma unroll
    for (int j = 0; j < U; ++j) unit_load(j, pn, blk_begin * CK, live, &deep[j]);
  }

  // ---- LDS base of each tile row of this lane: pixel (py, px) of the patch at tap (0, 0), this lane's k half, this k-group's chunks
  uint32_t a_addr[QTM];
#pragma unroll
  for (int i = 0; i < QTM; ++i) {
    const int q = i * 32 + li, py = q / QPW, px = q - py * QPW;
    a_addr[i] = (uint32_t)((py + wm * 8) * RPB + px * S + lh * 16 + kg * CCL * 32);
  }
  f32x16 acc[QTM];
  bf16x8_t afr[3][QTM];
  auto fetch_a = [&](const int s) __attribute__((always_inline)) {
    const int t = s / CCL, c = s % CCL;
    const int off = (t / 3) * RPB + (t % 3) * S + c * 32;
#pragma unroll
    for (int i = 0; i < QTM; ++i) afr[s % 3][i] = __builtin_bit_cast(bf16x8_t, *reinterpret_cast<const qu32x4*>(smem + a_addr[i] + off));
  };
  // epilogue constants of this thread
  const int ucol = (tid % UPR) * 8, urow = tid / UPR;
  const int ecol = n0 + ucol;
  const bool cok = ecol < a.Ntot;                   // (Ntot % 8 == 0: a unit is inside or outside as a whole)
  const bool sliced = p.splits > 1;
  const bool do_stats = !sliced && (a.stat_partials != nullptr || a.stat_totals != nullptr);
  const unsigned short* addp = reinterpret_cast<const unsigned short*>(a.add);
  const unsigned short* sop = reinterpret_cast<const unsigned short*>(a.stat_other);
  unsigned short* yp = reinterpret_cast<unsigned short*>(a.y);
  __syncthreads();

  uint32_t cur = 0u;                                // byte offset of the image the k-steps read
  for (;;) {
    const int pnext = pidx + pstride;
    const bool next_ok = PERS != 0 && pnext < p.npatch;
    const Pos posn = pos_of(next_ok ? pnext : pidx);
#pragma unroll
    for (int i = 0; i < QTM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    fetch_a(0);
    fetch_a(1);
    for (int blk = blk_begin; blk < blk_end; ++blk) {
      const bool last = PERS == 2 ? true : blk + 1 == blk_end;        // (PERS 2: patches of one block, known at compile time)
      const bool ring_next = !last || next_ok;                        // the ring's last two groups: the next block's / next patch's first ones
      const bool has_next = PERS == 2 ? false : (!last || (PERS == 1 && next_ok));   // an image is staged between this block's k-steps
      const uint32_t wb = wbase(blk), wbn = wbase(last ? blk_begin : blk + 1);
      const uint32_t nxt = BUF - cur;
      const Pos& psn = last ? posn : pos;
      const int c0n = last ? blk_begin * CK : (blk + 1) * CK, creln = last ? 0 : (blk + 1 - blk_begin) * CK;
      Unit un[CHU];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
#pragma unroll
        for (int c = 0; c < CCL; ++c) {
          const int s = t * CCL + c;
          if (s % GS == 0) {
            const int g = s / GS + 2;
            if (g < NG) load_group(g, wb, true);
            else load_group(g - NG, wbn, ring_next);
          }
          if (s + 2 < STEPS) fetch_a(s + 2);
          const bf16x8_t bf = __builtin_bit_cast(bf16x8_t, breg[(s / GS) % 3][s % GS]);
#pragma unroll
          for (int i = 0; i < QTM; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[s % 3][i], bf, acc[i], 0, 0, 0);
          // keep the software pipeline as written: left alone the scheduler sinks every fragment read to just in front of its matrix
          // instruction (one register set, lgkmcnt(0) before each instruction)
          __builtin_amdgcn_sched_barrier(0);
        }
        // the next image, chunk by chunk between the taps: chunk ch is requested behind tap 2 ch and written behind tap 2 ch + 2
        if ((t & 1) == 0 && has_next) {
          const int ch = t / 2;
          if (ch >= 1 && ch - 1 < NCH) {
#pragma unroll
            for (int e = 0; e < CHU; ++e)
              if ((ch - 1) * CHU + e < U) unit_store((ch - 1) * CHU + e, psn, c0n, creln, nxt, un[e]);
          }
          if (ch < NCH) {
#pragma unroll
            for (int e = 0; e < CHU; ++e)
              if (ch * CHU + e < U) unit_load(ch * CHU + e, psn, c0n, true, &un[e]);
          }
        }
      }
      if (!last) {
        __syncthreads();                            // the next image is complete, this one is no longer read
#pragma unroll
        for (int i = 0; i < QTM; ++i) a_addr[i] += nxt - cur;     // (flips between the two images)
        cur = nxt;
        fetch_a(0);
        fetch_a(1);
      }
    }
    if constexpr (PERS == 2) {                      // the next patch's image: its loads have been flying since before the last epilogue
      if (next_ok) {
#pragma unroll
        for (int j = 0; j < U; ++j) unit_store(j, posn, blk_begin * CK, 0, BUF - cur, deep[j]);
      }
      const bool live = pnext + pstride < p.npatch;               // ... and the one after it goes out now
      const Pos pn2 = pos_of(live ? pnext + pstride : pidx);
#pragma unroll
      for (int j = 0; j < U; ++j) unit_load(j, pn2, blk_begin * CK, live, &deep[j]);
    }

    // ================================================================== epilogue: the k-groups' tiles -> LDS, all threads run the rows
    __syncthreads();                                // every wave is done with this patch's image (and has written its share of the next)
    float* ep = reinterpret_cast<float*>(smem + (PERS != 0 ? cur : 0u));
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s1[e] = 0.f;
      s2[e] = 0.f;
    }
#pragma unroll
    for (int pass = 0; pass < EPP; ++pass) {
      if (pass > 0) __syncthreads();                // the previous pass's rows have been read
#pragma unroll
      for (int i = 0; i < TPP; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) ep[(kg * RI + wm * RW + i * 32 + mfma32_row(r, lane)) * QEP + wn * 32 + li] = acc[pass * TPP + i][r];
      __syncthreads();
#pragma unroll ROWS_UNROLL
      for (int rl = urow; rl < RI; rl += RPS) {
        const int rw = rl / RW, q = rw * 96 + pass * RW + (rl - rw * RW);      // image row -> patch row
        const int py = q / QPW, px = q - py * QPW;
        const int gy = pos.gy0 + py, gx = pos.gx0 + px;
        if (!(cok && gy < p.Hg && gx < p.Wg)) continue;
        const uint32_t m = (uint32_t)((pos.img * p.Hg + gy) * p.Wg + gx);
        float v[8];
        {
          const float4 v0 = *reinterpret_cast<const float4*>(ep + rl * QEP + ucol);
          const float4 v1 = *reinterpret_cast<const float4*>(ep + rl * QEP + ucol + 4);
          v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
        }
        if constexpr (KS == 2) {
          const float4 v0 = *reinterpret_cast<const float4*>(ep + (RI + rl) * QEP + ucol);
          const float4 v1 = *reinterpret_cast<const float4*>(ep + (RI + rl) * QEP + ucol + 4);
          v[0] += v0.x; v[1] += v0.y; v[2] += v0.z; v[3] += v0.w; v[4] += v1.x; v[5] += v1.y; v[6] += v1.z; v[7] += v1.w;
        }
        if (sliced) {                               // K slices: raw fp32 sums, convp_finish_kernel does the rest
          float* part = a.partial + ((int64_t)bslice * p.Mtot + m) * a.Ntot + ecol;
          *reinterpret_cast<float4*>(part) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(part + 4) = make_float4(v[4], v[5], v[6], v[7]);
          continue;
        }
        {                                           // (the per-column constants come from the LDS table row by row: 24 registers less to hold)
          const float4 b0 = *reinterpret_cast<const float4*>(col_tab + ucol), b1 = *reinterpret_cast<const float4*>(col_tab + ucol + 4);
          const float cb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v[e] += cb[e];
            if (a.relu) v[e] = fmaxf(v[e], 0.f);
          }
        }
        if (addp != nullptr) {
          float o[8];
          q_unpack(*reinterpret_cast<const qu32x4*>(addp + (m * (uint32_t)a.ldadd + (uint32_t)ecol)), o);
          const float4 c0 = *reinterpret_cast<const float4*>(col_tab + QBN + ucol), c1 = *reinterpret_cast<const float4*>(col_tab + QBN + ucol + 4);
          const float4 h0 = *reinterpret_cast<const float4*>(col_tab + 2 * QBN + ucol), h1 = *reinterpret_cast<const float4*>(col_tab + 2 * QBN + ucol + 4);
          const float sc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += fmaf(o[e], sc[e], sh[e]);
        }
        const uint32_t yo = m * (uint32_t)a.ldy + (uint32_t)ecol;
        if (a.accumulate) {
          float o[8];
          q_unpack(*reinterpret_cast<const qu32x4*>(yp + yo), o);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += o[e];
        }
        qu32x4 w;
        w.x = q_pack(v[0], v[1]);
        w.y = q_pack(v[2], v[3]);
        w.z = q_pack(v[4], v[5]);
        w.w = q_pack(v[6], v[7]);
        *reinterpret_cast<qu32x4*>(yp + yo) = w;
        if (do_stats) {
          float vr[8], u[8];
          q_unpack(w, vr);                          // statistics of the values as stored
          if (sop != nullptr) {
            q_unpack(*reinterpret_cast<const qu32x4*>(sop + (m * (uint32_t)a.ldso + (uint32_t)ecol)), u);
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
    }
    if (do_stats) {
      // per-column sums of this patch -> one row of stat_partials (rows = patches) or the layer's live totals; fixed order
      __syncthreads();
      float* red = ep;                              // [RPS][2][QBN]
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        red[(urow * 2 + 0) * QBN + ucol + e] = s1[e];
        red[(urow * 2 + 1) * QBN + ucol + e] = s2[e];
      }
      __syncthreads();
      for (int idx = tid; idx < 2 * QBN; idx += NT) {
        const int which = idx / QBN, col = idx - which * QBN;
        const int n = n0 + col;
        if (n < a.Ntot) {
          float sum = 0.f;
          for (int w = 0; w < RPS; ++w) sum += red[(w * 2 + which) * QBN + col];
          if (a.stat_totals != nullptr) bn_live_add(a.stat_totals, pidx, which, a.Ntot, n, sum);
          else a.stat_partials[((int64_t)pidx * 2 + which) * a.Ntot + n] = sum;
        }
      }
    }
    if (!next_ok) break;
    __syncthreads();                                // the epilogue's image region goes back to the staging of the patch after the next
#pragma unroll
    for (int i = 0; i < QTM; ++i) a_addr[i] += (BUF - cur) - cur;
    cur = BUF - cur;
    pidx = pnext;
    pos = posn;
  }
}

struct QCfg { int WM, WN, KS; };
// tile configuration CONVQ_TILE + index (csrc/convp_bf16.hip kTiles lists the same triples)
constexpr QCfg kQ[] = {{1, 4, 1}, {1, 4, 2}, {2, 4, 1}, {2, 2, 1}, {4, 2, 1}, {2, 2, 2}, {4, 1, 1}, {8, 1, 1}, {4, 1, 2}};
constexpr int kNumQ = (int)(sizeof(kQ) / sizeof(kQ[0]));

size_t q_tab_off(int ck, int mode, int blk_per_slice, int pers) {
  const QCfg c = kQ[mode];
  if (pers != 0) return 2 * (size_t)q_buf_bytes(ck, c.WM, c.WN, c.KS, 1);
  const int NT = 64 * c.WM * c.WN * c.KS, BN = 32 * c.WN, EPP = c.WN == 4 ? 1 : 3;
  size_t lds = (size_t)(blk_per_slice > 1 ? 2 : 1) * (8 * c.WM + 2) * q_row_pitch(ck);
  const size_t epi = (size_t)c.KS * c.WM * (96 / EPP) * (BN + 4) * sizeof(float);
  const size_t red = (size_t)(NT / (BN / 8)) * 2 * BN * sizeof(float);
  if (lds < epi) lds = epi;
  if (lds < red) lds = red;
  return (lds + 15) / 16 * 16;
}

// Persistent forms that would spill are not built: four waves staging 128-channel images of the two-tensor operand (x_mode)
constexpr bool q_pers_ok(int ck, int nt, int x_mode, int pers) { return pers == 0 || !(ck == 128 && nt == 256 && x_mode != 0); }

template <int CK, int WM, int WN, int KS, int PERS>
int convq_launch_t(const ConvP& p, int mode, hipStream_t s) {
  constexpr int NT = 64 * WM * WN * KS;
  ConvP pl = p;
  pl.tab_off = (int)q_tab_off(CK, mode, p.blk_per_slice, PERS);
  const size_t lds = convq_lds_bytes(CK, mode, p.blk_per_slice, PERS);
  DFL_REQUIRE(lds <= 160 * 1024, "dfl_conv2d (bf16, unrolled 3x3): %zu bytes of LDS", lds);
  const bool aff = p.a.in_scale != nullptr || p.a.in_tot != nullptr;
  dim3 grid((unsigned)p.grid);
#define DFL_CQ_LAUNCH(AFF_)                                                                                                   \
  {                                                                                                                             \
    auto k = convq_kernel<CK, WM, WN, KS, AFF_, PERS>;                                                                          \
    DFL_LDS_OPT_IN(k, 160 * 1024, "dfl_conv2d (bf16, unrolled 3x3)") \
    hipLaunchKernelGGL(k, grid, dim3(NT), lds, s, pl);                                                                          \
  }
  if (p.a.x_mode != 0) {
    if constexpr (q_pers_ok(CK, NT, 1, PERS) && PERS != 2) DFL_CQ_LAUNCH(2)
    else DFL_REQUIRE(false, "dfl_conv2d (bf16, unrolled 3x3): the persistent form of this layout does not take the two-tensor operand");
  } else if (aff) DFL_CQ_LAUNCH(1)
  else DFL_CQ_LAUNCH(0)
#undef DFL_CQ_LAUNCH
  return check_launch("dfl_conv2d (bf16, unrolled 3x3)");
}

// A configuration is instantiated for the resident blocks whose image is at most 10 staging units (16 bytes) per thread: beyond
// that the staging registers spill (and such an image leaves no room for a second one anyway)
constexpr bool q_inst(int ck, int wm, int wn, int ks) {
  return ((8 * wm + 2) * QIW * (ck / 8) + 64 * wm * wn * ks - 1) / (64 * wm * wn * ks) <= 10 && ck / 16 >= ks;
}

template <int CK, int WM, int WN, int KS>
int convq_launch_p(const ConvP& p, int mode, int pers, hipStream_t s) {
  if (pers == 0) return convq_launch_t<CK, WM, WN, KS, 0>(p, mode, s);
  // persistent: patches of one channel block keep the next patch's loads in flight across the epilogue, longer ones stage it in their last block
  // (128 resident channels are 72 k-steps: long enough for the chunked staging, and the whole image in registers would spill)
  // ... as would the two-tensor operand's)
  if constexpr (CK <= 64) {
    if (p.blk_per_slice == 1 && p.a.x_mode == 0) return convq_launch_t<CK, WM, WN, KS, 2>(p, mode, s);
  }
  return convq_launch_t<CK, WM, WN, KS, 1>(p, mode, s);
}

template <int WM, int WN, int KS>
int convq_launch_ck(const ConvP& p, int mode, int pers, hipStream_t s) {
  if (p.CK == 128) {
    if constexpr (q_inst(128, WM, WN, KS)) return convq_launch_p<128, WM, WN, KS>(p, mode, pers, s);
  } else if (p.CK == 64) {
    if constexpr (q_inst(64, WM, WN, KS)) return convq_launch_p<64, WM, WN, KS>(p, mode, pers, s);
  } else if (p.CK == 32) {
    if constexpr (q_inst(32, WM, WN, KS)) return convq_launch_p<32, WM, WN, KS>(p, mode, pers, s);
  }
  set_error("dfl_conv2d (bf16, unrolled 3x3): no instantiation for %d resident channels in configuration %d", p.CK, mode);
  return DFL_ERR_INVALID_ARG;
}

}  // namespace

// The layer shapes this form takes (the caller has validated the argument block as convp_plan_search does)
bool convq_shape_ok(const dfl_conv_args& a) {
  return a.KH == 3 && a.KW == 3 && a.stride == 1 && a.pad == 1 && a.scatter2x2 == 0 && a.Cin % 32 == 0 && a.Ntot % 8 == 0 &&
         a.Hout == a.Hin && a.Wout == a.Win && a.out_scale == nullptr;
}

// Resident channels a configuration is instantiated for (q_inst)
bool convq_ck_ok(int mode, int ck) {
  if (mode < 0 || mode >= kNumQ || (ck != 32 && ck != 64 && ck != 128)) return false;
  const QCfg c = kQ[mode];
  return q_inst(ck, c.WM, c.WN, c.KS);
}

size_t convq_lds_bytes(int ck, int mode, int blk_per_slice, int pers) {
  return q_tab_off(ck, mode, blk_per_slice, pers) + (size_t)(3 * 32 * kQ[mode].WN + 3 * blk_per_slice * ck) * sizeof(float);
}

bool convq_pers_ok(int mode, int ck, int x_mode) { return q_pers_ok(ck, convq_threads(mode), x_mode, 1); }

int convq_threads(int mode) { return 64 * kQ[mode].WM * kQ[mode].WN * kQ[mode].KS; }

int convq_launch(const ConvP& p, int mode, int pers, hipStream_t s) {
  DFL_REQUIRE(convq_ck_ok(mode, p.CK), "dfl_conv2d (bf16, unrolled 3x3): resident channel block %d, configuration %d", p.CK, mode);
  switch (mode) {
    case 0: return convq_launch_ck<1, 4, 1>(p, mode, pers, s);
    case 1: return convq_launch_ck<1, 4, 2>(p, mode, pers, s);
    case 2: return convq_launch_ck<2, 4, 1>(p, mode, pers, s);
    case 3: return convq_launch_ck<2, 2, 1>(p, mode, pers, s);
    case 4: return convq_launch_ck<4, 2, 1>(p, mode, pers, s);
    case 5: return convq_launch_ck<2, 2, 2>(p, mode, pers, s);
    case 6: return convq_launch_ck<4, 1, 1>(p, mode, pers, s);
    case 7: return convq_launch_ck<8, 1, 1>(p, mode, pers, s);
    default: return convq_launch_ck<4, 1, 2>(p, mode, pers, s);
  }
}

}  // namespace dfl
