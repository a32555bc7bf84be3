// Output heads of the U-Net (reference: train_test_code/unet.py:176-191):
//   logits = seg_conv(x)                  1x1, no bias, F -> NC
//   seg    = Softmax2d(logits)            (or logits when do_soft_max=False)
//   heat   = lands_1x1(cat(x, logits))    1x1 (F+NC -> NM) [-> 1x1 (NM -> L)], no bias, no non-linearity
// One thread per pixel, 256 consecutive pixels per workgroup.  A thread-private walk over its 128-byte feature row would
// touch 64 cache lines per wave-instruction (measured: 5x the compulsory HBM reads), so the [256][F] tile is loaded
// cooperatively (consecutive lanes = consecutive 16 bytes) into LDS (pitch F + 4: conflict-free b128 row reads) and the
// same tile carries dx back out in the backward kernel; the head weights (5 KB) are read with
// wave-uniform addresses through `const __restrict__` kernel parameters, i.e. as scalar loads into SGPR operands of the
// FMAs (no LDS traffic: the LDS-broadcast version spent 224 ds_read_b128 per pixel), logits / mid / heat stay in registers, seg and heat are stored NCHW (lane = pixel
// => unit-stride stores per channel plane).  HBM-bound: 4*F bytes in, 4*(NC+L) bytes out per pixel.
// The backward kernel recomputes logits and mid from x (cheaper than saving them), applies the softmax Jacobian
//  and leaves dx plus a per-pixel scratch row from which the three small weight gradients are
// taken by dfl_conv2d_wgrad.  With bf16 features (F = 32) the weight gradients are taken inside the kernel instead (the
// scratch is 448 bytes per pixel to write and read back): a tile's rows [dlogits | dmid | dheat] and [x | logits | mid] go
// to LDS as bf16, four waves multiply them on the matrix cores through transposing LDS reads (contraction over pixels,
// as csrc/wgradp_bf16.hip), workgroups leave 64 x 64 fp32 partials and head_wgrad_finish_kernel adds them up.
#include "common.h"


namespace dfl {

constexpr int MAXNC = DFL_HEAD_MAX_NC, MAXL = DFL_HEAD_MAX_L, MAXNM = DFL_HEAD_MAX_NM;

static inline int head_fc(int F) { return ((F + MAXNC) + 3) / 4 * 4; }

constexpr int HT = 256;   // pixels per workgroup tile

__device__ __forceinline__ unsigned hpack2(float x, float y) {   // two floats -> two bf16 (round to nearest even)
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){x, y}, bf16x2_t));
}
// 8 consecutive pixels of one column of a [pixel][column] bf16 image: two transposing reads (csrc/wgradp_bf16.hip)
__device__ __forceinline__ bf16x8_t htr_read8(const unsigned char* base, uint32_t r0, uint32_t r1) {
  typedef short hs16x4_t __attribute__((ext_vector_type(4)));
  typedef __attribute__((address_space(3))) hs16x4_t* lds_p;
  const hs16x4_t x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + r0));
  const hs16x4_t y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(base + r1));
  return __builtin_bit_cast(bf16x8_t, __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7));
}

// tile[p][0..F) = x[m0 + p][0..F) (zeros past M); optionally the same values go to cat[(m0 + p) * cat_ld + ..] (scratch)
__device__ __forceinline__ void head_load_tile(float* tile, int pitch, const float* __restrict__ x, int ldx, int64_t m0,
                                               int64_t M, int F, float* __restrict__ cat, int cat_ld, int x_bf16) {
  if (x_bf16) {   // bf16 features (math mode 4): 8 channels per 16-byte load, widened to fp32 in the tile
    const int f8 = F / 8;
    const unsigned short* xb = reinterpret_cast<const unsigned short*>(x);
    for (int e = threadIdx.x; e < HT * f8; e += HT) {
      const int p = e / f8, q = e - p * f8;
      float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
      if (m0 + p < M) {
        const uint4 w = *reinterpret_cast<const uint4*>(xb + (m0 + p) * ldx + 8 * q);
        lo = make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16),
                         __uint_as_float(w.y & 0xffff0000u));
        hi = make_float4(__uint_as_float(w.z << 16), __uint_as_float(w.z & 0xffff0000u), __uint_as_float(w.w << 16),
                         __uint_as_float(w.w & 0xffff0000u));
        if (cat != nullptr) {
          *reinterpret_cast<float4*>(cat + (m0 + p) * cat_ld + 8 * q) = lo;
          *reinterpret_cast<float4*>(cat + (m0 + p) * cat_ld + 8 * q + 4) = hi;
        }
      }
      *reinterpret_cast<float4*>(tile + p * pitch + 8 * q) = lo;
      *reinterpret_cast<float4*>(tile + p * pitch + 8 * q + 4) = hi;
    }
    return;
  }
  const int fq = F / 4;
  for (int e = threadIdx.x; e < HT * fq; e += HT) {
    const int p = e / fq, q = e - p * fq;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (m0 + p < M) {
      v = *reinterpret_cast<const float4*>(x + (m0 + p) * ldx + 4 * q);
      if (cat != nullptr) *reinterpret_cast<float4*>(cat + (m0 + p) * cat_ld + 4 * q) = v;
    }
    *reinterpret_cast<float4*>(tile + p * pitch + 4 * q) = v;
  }
}

// The channel counts are template parameters for the reference's two head shapes (7 classes alone, 7 + 14 landmarks
// through 21): with run-time bounds every 4-FMA group became its own basic block (scalar load, wait, branch -- ~950
// branches per pixel, the kernels ran at 1 TB/s).  GEN = 1 keeps the run-time bounds for any other shape.
#define HEAD_TPL template <int NCc, int NMc, int Lc, bool GEN>
#define HEAD_BOUNDS                                                              \
  constexpr int BNC = GEN ? MAXNC : NCc, BNM = GEN ? MAXNM : NMc, BL = GEN ? MAXL : Lc; \
  (void)BNC; (void)BNM; (void)BL;

// logits (lg) and mid for pixel row xr (in LDS).  K1 = F + NC = row length of w_l1.
HEAD_TPL __device__ __forceinline__ void head_features(const float* xr, const float* __restrict__ w_seg,
                                              const float* __restrict__ w_l1, int F, int NC, int NM, bool lands,
                                              float* lg, float* mid) {
  HEAD_BOUNDS
  const int K1 = F + NC;
#pragma unroll
  for (int c = 0; c < MAXNC; ++c) lg[c] = 0.f;
#pragma unroll
  for (int j = 0; j < MAXNM; ++j) mid[j] = 0.f;
  for (int k = 0; k < F; k += 4) {
    const float4 xv = *reinterpret_cast<const float4*>(xr + k);
#pragma unroll
    for (int c = 0; c < BNC; ++c) {
      if (!GEN || c < NC) {
        const float* w = w_seg + c * F + k;
        lg[c] = fmaf(w[0], xv.x, fmaf(w[1], xv.y, fmaf(w[2], xv.z, fmaf(w[3], xv.w, lg[c]))));
      }
    }
    if (lands) {
#pragma unroll
      for (int j = 0; j < BNM; ++j) {
        if (!GEN || j < NM) {
          const float* w = w_l1 + j * K1 + k;
          mid[j] = fmaf(w[0], xv.x, fmaf(w[1], xv.y, fmaf(w[2], xv.z, fmaf(w[3], xv.w, mid[j]))));
        }
      }
    }
  }
  if (lands) {
#pragma unroll
    for (int j = 0; j < BNM; ++j) {
      if (!GEN || j < NM) {
#pragma unroll
        for (int c = 0; c < BNC; ++c)
          if (!GEN || c < NC) mid[j] = fmaf(w_l1[j * K1 + F + c], lg[c], mid[j]);
      }
    }
  }
}

HEAD_TPL __device__ __forceinline__ void softmax_inplace(float* lg, int NC) {
  HEAD_BOUNDS
  float mx = lg[0];
#pragma unroll
  for (int c = 1; c < BNC; ++c)
    if (!GEN || c < NC) mx = fmaxf(mx, lg[c]);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < BNC; ++c) {
    if (!GEN || c < NC) {
      lg[c] = expf(lg[c] - mx);
      sum += lg[c];
    }
  }
  const float inv = 1.0f / sum;
#pragma unroll
  for (int c = 0; c < BNC; ++c)
    if (!GEN || c < NC) lg[c] *= inv;
}

HEAD_TPL __global__ void __launch_bounds__(256) head_fwd_kernel(const dfl_head_fwd_args a, const float* __restrict__ w_seg,
                                                      const float* __restrict__ w_l1, const float* __restrict__ w_l2) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  HEAD_BOUNDS
  const int F = a.F, NC = GEN ? a.NC : NCc, NM = GEN ? a.NM : NMc, L = GEN ? a.L : Lc;
  const int pitch = F + 4;
  const int64_t HW = (int64_t)a.H * a.W;
  const int64_t M = (int64_t)a.N * HW;
  const bool lands = L > 0;
  for (int64_t m0 = (int64_t)blockIdx.x * HT; m0 < M; m0 += (int64_t)gridDim.x * HT) {
    __syncthreads();
    head_load_tile(tile, pitch, a.x, a.ldx, m0, M, F, nullptr, 0, a.x_bf16);
    __syncthreads();
    const int64_t m = m0 + threadIdx.x;
    if (m >= M) continue;
    float lg[MAXNC], mid[MAXNM];
    head_features<NCc, NMc, Lc, GEN>(tile + threadIdx.x * pitch, w_seg, w_l1, F, NC, NM, lands, lg, mid);
    const int64_t n = m / HW, pp = m - n * HW;
    if (lands) {
      float* hp = a.heat + n * L * HW + pp;
      if (w_l2 != nullptr) {
#pragma unroll
        for (int l = 0; l < BL; ++l) {
          if (!GEN || l < L) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < BNM; ++j)
              if (!GEN || j < NM) acc = fmaf(w_l2[l * NM + j], mid[j], acc);
            hp[(int64_t)l * HW] = acc;
          }
        }
      } else {
#pragma unroll
        for (int l = 0; l < BL; ++l)
          if (!GEN || l < L) hp[(int64_t)l * HW] = mid[l];
      }
    }
    if (a.softmax) softmax_inplace<NCc, NMc, Lc, GEN>(lg, NC);
    float* sp = a.seg + n * NC * HW + pp;
#pragma unroll
    for (int c = 0; c < BNC; ++c)
      if (!GEN || c < NC) sp[(int64_t)c * HW] = lg[c];
  }
}

template <int NCc, int NMc, int Lc, bool GEN, bool FUSED>
__global__ void __launch_bounds__(256) head_bwd_kernel(const dfl_head_bwd_args a, int Fc, const float* __restrict__ w_seg,
                                                      const float* __restrict__ w_l1, const float* __restrict__ w_l2) {
  extern __shared__ __attribute__((aligned(16))) float tile[];
  HEAD_BOUNDS
  const int F = a.F, NC = GEN ? a.NC : NCc, NM = GEN ? a.NM : NMc, L = GEN ? a.L : Lc;
  const int K1 = F + NC;
  const int pitch = F + 4;
  const int64_t HW = (int64_t)a.H * a.W;
  const int64_t M = (int64_t)a.N * HW;
  const bool lands = L > 0 && a.dheat != nullptr;
  const int o_dlg = Fc, o_dmid = Fc + MAXNC, o_mid = o_dmid + MAXNM, o_dh = o_mid + MAXNM;   // all multiples of 4
  // fused weight gradients: rows of 320 bytes behind the feature tile, 128 pixels at a time --
  //   bytes [0,128): dlogits (8) | dmid (24) | dheat (16) | 0 (16)      bytes [128,256): x (32) | logits (8) | mid (24)   (bf16)
  constexpr bool fused = FUSED;
  unsigned char* ab = reinterpret_cast<unsigned char*>(tile) + (size_t)HT * pitch * sizeof(float);
  constexpr int ABP = 320, ABH = 128;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int trow = 8 * (lane >> 5) + ((lane & 15) >> 2);
  const uint32_t tcb = (uint32_t)((16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2);
  const uint32_t a_col = (uint32_t)((wave >> 1) * 64) + tcb, b_col = 128u + (uint32_t)((wave & 1) * 64) + tcb;
  f32x16 wacc;
#pragma unroll
  for (int r = 0; r < 16; ++r) wacc[r] = 0.f;
  for (int64_t m0 = (int64_t)blockIdx.x * HT; m0 < M; m0 += (int64_t)gridDim.x * HT) {
    __syncthreads();
    head_load_tile(tile, pitch, a.x, a.ldx, m0, M, F, fused ? nullptr : a.scratch, a.scratch_ld, a.x_bf16);   // also copies x into the cat columns
    __syncthreads();
    const int64_t m = m0 + threadIdx.x;
    uint4 rowA[8], rowB[8];                 // this pixel's two bf16 rows (fused form): 8 x 16 bytes each
    if (fused) {
#pragma unroll
      for (int q = 0; q < 8; ++q) rowA[q] = rowB[q] = make_uint4(0u, 0u, 0u, 0u);
    }
    if (m < M) {
      float* myrow = tile + threadIdx.x * pitch;
      float* sr = fused ? nullptr : a.scratch + m * a.scratch_ld;
      float lg[MAXNC], mid[MAXNM];
      head_features<NCc, NMc, Lc, GEN>(myrow, w_seg, w_l1, F, NC, NM, L > 0, lg, mid);
      const int64_t n = m / HW, pp = m - n * HW;
      if (fused) {                          // x (exact: it was bf16), logits, mid
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v0 = *reinterpret_cast<const float4*>(myrow + 8 * q), v1 = *reinterpret_cast<const float4*>(myrow + 8 * q + 4);
          rowB[q] = make_uint4(hpack2(v0.x, v0.y), hpack2(v0.z, v0.w), hpack2(v1.x, v1.y), hpack2(v1.z, v1.w));
        }
        rowB[4] = make_uint4(hpack2(lg[0], lg[1]), hpack2(lg[2], lg[3]), hpack2(lg[4], lg[5]), hpack2(lg[6], lg[7]));
        const bool has = L > 0;
#pragma unroll
        for (int q = 0; q < 3; ++q)
          rowB[5 + q] = make_uint4(hpack2(has ? mid[8 * q] : 0.f, has ? mid[8 * q + 1] : 0.f), hpack2(has ? mid[8 * q + 2] : 0.f, has ? mid[8 * q + 3] : 0.f),
                                   hpack2(has ? mid[8 * q + 4] : 0.f, has ? mid[8 * q + 5] : 0.f), hpack2(has ? mid[8 * q + 6] : 0.f, has ? mid[8 * q + 7] : 0.f));
      } else {
      // cat tail: logits then zero pad (16-byte stores: the scratch row is 16-byte aligned, every block a multiple of 4)
#pragma unroll
      for (int c = 0; c < MAXNC; c += 4)
        *reinterpret_cast<float4*>(sr + F + c) = make_float4(c + 0 < NC ? lg[c + 0] : 0.f, c + 1 < NC ? lg[c + 1] : 0.f,
                                                             c + 2 < NC ? lg[c + 2] : 0.f, c + 3 < NC ? lg[c + 3] : 0.f);
      for (int c = F + MAXNC; c < Fc; c += 4) *reinterpret_cast<float4*>(sr + c) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      // landmark branch
      float dmid[MAXNM], dh[MAXL];
#pragma unroll
      for (int l = 0; l < MAXL; ++l) dh[l] = (lands && l < L) ? a.dheat[(n * L + l) * HW + pp] : 0.f;
#pragma unroll
      for (int j = 0; j < MAXNM; ++j) {
        float acc = 0.f;
        if (lands && j < NM) {
          if (w_l2 != nullptr) {
#pragma unroll
            for (int l = 0; l < BL; ++l)
              if (!GEN || l < L) acc = fmaf(w_l2[l * NM + j], dh[l], acc);
          } else {
            acc = (j < MAXL) ? dh[j < MAXL ? j : 0] : 0.f;
          }
        }
        dmid[j] = acc;
      }
      // logits gradient: through cat (landmark branch) + through softmax (seg branch)
      float dlg[MAXNC];
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < MAXNC; ++c) {
        float g = 0.f, sv = 0.f;
        if (c < NC) {
          g = a.dseg[(n * NC + c) * HW + pp];
          sv = a.seg[(n * NC + c) * HW + pp];
        }
        dlg[c] = g;      // provisional: upstream gradient
        lg[c] = sv;      // reuse lg for the softmax output
        dot = fmaf(g, sv, dot);
      }
#pragma unroll
      for (int c = 0; c < MAXNC; ++c) {
        float v = 0.f;
        if (c < NC) {
          v = a.softmax ? lg[c] * (dlg[c] - dot) : dlg[c];
          if (lands) {
#pragma unroll
            for (int j = 0; j < BNM; ++j)
              if (!GEN || j < NM) v = fmaf(w_l1[j * K1 + F + c], dmid[j], v);
          }
        }
        dlg[c] = v;
      }
      if (fused) {
        rowA[0] = make_uint4(hpack2(dlg[0], dlg[1]), hpack2(dlg[2], dlg[3]), hpack2(dlg[4], dlg[5]), hpack2(dlg[6], dlg[7]));
#pragma unroll
        for (int q = 0; q < 3; ++q)
          rowA[1 + q] = make_uint4(hpack2(dmid[8 * q], dmid[8 * q + 1]), hpack2(dmid[8 * q + 2], dmid[8 * q + 3]),
                                   hpack2(dmid[8 * q + 4], dmid[8 * q + 5]), hpack2(dmid[8 * q + 6], dmid[8 * q + 7]));
#pragma unroll
        for (int q = 0; q < 2; ++q)
          rowA[4 + q] = make_uint4(hpack2(dh[8 * q], dh[8 * q + 1]), hpack2(dh[8 * q + 2], dh[8 * q + 3]),
                                   hpack2(dh[8 * q + 4], dh[8 * q + 5]), hpack2(dh[8 * q + 6], dh[8 * q + 7]));
      } else {
#pragma unroll
      for (int c = 0; c < MAXNC; c += 4)
        *reinterpret_cast<float4*>(sr + o_dlg + c) = make_float4(dlg[c], dlg[c + 1], dlg[c + 2], dlg[c + 3]);
#pragma unroll
      for (int j = 0; j < MAXNM; j += 4) {
        *reinterpret_cast<float4*>(sr + o_dmid + j) = make_float4(dmid[j], dmid[j + 1], dmid[j + 2], dmid[j + 3]);
        const bool has = L > 0;
        *reinterpret_cast<float4*>(sr + o_mid + j) = make_float4((has && j + 0 < NM) ? mid[j + 0] : 0.f, (has && j + 1 < NM) ? mid[j + 1] : 0.f,
                                                                (has && j + 2 < NM) ? mid[j + 2] : 0.f, (has && j + 3 < NM) ? mid[j + 3] : 0.f);
      }
#pragma unroll
      for (int l = 0; l < MAXL; l += 4) *reinterpret_cast<float4*>(sr + o_dh + l) = make_float4(dh[l], dh[l + 1], dh[l + 2], dh[l + 3]);
      }
      // dx = Wseg^T dlogits + W1[:, :F]^T dmid, into this thread's own tile row (x is no longer needed)
      for (int k = 0; k < F; k += 4) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < BNC; ++c) {
          if (!GEN || c < NC) {
            const float* w = w_seg + c * F + k;
            acc.x = fmaf(w[0], dlg[c], acc.x); acc.y = fmaf(w[1], dlg[c], acc.y);
            acc.z = fmaf(w[2], dlg[c], acc.z); acc.w = fmaf(w[3], dlg[c], acc.w);
          }
        }
        if (lands) {
#pragma unroll
          for (int j = 0; j < BNM; ++j) {
            if (!GEN || j < NM) {
              const float* w = w_l1 + j * K1 + k;
              acc.x = fmaf(w[0], dmid[j], acc.x); acc.y = fmaf(w[1], dmid[j], acc.y);
              acc.z = fmaf(w[2], dmid[j], acc.z); acc.w = fmaf(w[3], dmid[j], acc.w);
            }
          }
        }
        *reinterpret_cast<float4*>(myrow + k) = acc;
      }
    }
    if (fused) {            // weight gradients of this tile: two halves of 128 pixels through the row image
      for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();          // the waves are done reading the first half
        if ((threadIdx.x >> 7) == half) {
          uint4* dst = reinterpret_cast<uint4*>(ab + (size_t)(threadIdx.x & (ABH - 1)) * ABP);
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            dst[q] = rowA[q];
            dst[8 + q] = rowB[q];
          }
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < ABH / 16; ++ks) {
          const uint32_t r0 = (uint32_t)(ks * 16 + trow) * (uint32_t)ABP, r1 = r0 + 4u * (uint32_t)ABP;
          const bf16x8_t af = htr_read8(ab, r0 + a_col, r1 + a_col);
          const bf16x8_t bf = htr_read8(ab, r0 + b_col, r1 + b_col);
          wacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, wacc, 0, 0, 0);
        }
      }
    }
    __syncthreads();
    if (a.x_bf16) {         // dx as bf16: 8 channels per 16-byte store
      const int f8 = F / 8;
      unsigned short* dxb = reinterpret_cast<unsigned short*>(a.dx);
      for (int e = threadIdx.x; e < HT * f8; e += HT) {
        const int p = e / f8, q = e - p * f8;
        if (m0 + p < M) {
          const float* t = tile + p * pitch + 8 * q;
          uint4 w;
          w.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){t[0], t[1]}, bf16x2_t));
          w.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){t[2], t[3]}, bf16x2_t));
          w.z = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){t[4], t[5]}, bf16x2_t));
          w.w = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){t[6], t[7]}, bf16x2_t));
          *reinterpret_cast<uint4*>(dxb + (m0 + p) * a.lddx + 8 * q) = w;
        }
      }
      continue;
    }
    const int fq = F / 4;   // cooperative, coalesced store of the dx tile
    for (int e = threadIdx.x; e < HT * fq; e += HT) {
      const int p = e / fq, q = e - p * fq;
      if (m0 + p < M) *reinterpret_cast<float4*>(a.dx + (m0 + p) * a.lddx + 4 * q) = *reinterpret_cast<const float4*>(tile + p * pitch + 4 * q);
    }
  }
  if (fused) {              // partial[block][row of A (64)][column of B (64)]
    float* part = a.wg_partial + (int64_t)blockIdx.x * 4096;
    const int col = (wave & 1) * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) part[((wave >> 1) * 32 + mfma32_row(r, lane)) * 64 + col] = wacc[r];
  }
}

// Sums the workgroup partials of the fused head weight gradients (fixed order, fp64) and files the three blocks of the
// 64 x 64 product: rows 0-7 x columns 0-31 -> dw_seg, rows 8-31 x columns 0-39 -> dw_l1, rows 32-47 x columns 40-63 -> dw_l2.
__global__ void __launch_bounds__(256) head_wgrad_finish_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ dw_seg,
                                                                float* __restrict__ dw_l1, float* __restrict__ dw_l2, int F, int NC, int NM, int L) {
  // 16 elements of the 64 x 64 product per workgroup x 16 lanes over the workgroup partials (eight loads in flight per
  // lane), lanes added up through LDS in a fixed order
  __shared__ double red[16][17];
  const int el = threadIdx.x & 15, ln = threadIdx.x >> 4;
  const int e = blockIdx.x * 16 + el;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int b = ln;
  for (; b + 7 * 16 < nblocks; b += 8 * 16) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = partial[(int64_t)(b + u * 16) * 4096 + e];
    s0 += (double)v[0] + (double)v[4];
    s1 += (double)v[1] + (double)v[5];
    s2 += (double)v[2] + (double)v[6];
    s3 += (double)v[3] + (double)v[7];
  }
  for (; b < nblocks; b += 16) s0 += (double)partial[(int64_t)b * 4096 + e];
  red[ln][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ln != 0) return;
  double tot = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) tot += red[k][el];
  const int row = e >> 6, col = e & 63;
  float* dst = nullptr;
  if (row < 8) {
    if (row < NC && col < F) dst = dw_seg + row * F + col;
  } else if (row < 32) {
    if (dw_l1 != nullptr && row - 8 < NM && col < F + NC) dst = dw_l1 + (row - 8) * (F + NC) + col;
  } else if (row < 48) {
    if (dw_l2 != nullptr && row - 32 < L && col >= 40 && col - 40 < NM) dst = dw_l2 + (row - 32) * NM + (col - 40);
  }
  if (dst != nullptr) *dst = (float)tot;
}

static int head_check(int F, int NC, int NM, int L, const void* w_l1, const void* w_l2, int ldx, const void* x, int x_bf16 = 0) {
  DFL_REQUIRE(F >= 4 && F % 4 == 0, "dfl_head: F must be a multiple of 4 (got %d)", F);
  DFL_REQUIRE(!x_bf16 || (F % 8 == 0 && ldx % 8 == 0), "dfl_head (bf16): F and ldx must be multiples of 8");
  DFL_REQUIRE(NC >= 1 && NC <= MAXNC, "dfl_head: n_classes %d exceeds the supported maximum %d", NC, MAXNC);
  DFL_REQUIRE(L >= 0 && L <= MAXL, "dfl_head: num_lands %d exceeds the supported maximum %d", L, MAXL);
  DFL_REQUIRE(ldx % 4 == 0 && aligned16(x), "dfl_head: x must be 16-byte aligned with ld %% 4 == 0");
  if (L > 0) {
    DFL_REQUIRE(w_l1 != nullptr, "dfl_head: w_l1 required when L > 0");
    DFL_REQUIRE(NM >= 1 && NM <= MAXNM, "dfl_head: mid width %d exceeds the supported maximum %d", NM, MAXNM);
    DFL_REQUIRE(w_l2 != nullptr || NM == L, "dfl_head: single 1x1 needs NM == L");
  }
  return DFL_OK;
}

}  // namespace dfl

using namespace dfl;

extern "C" int dfl_head_scratch_ld(int32_t F) { return head_fc(F) + MAXNC + 2 * MAXNM + MAXL; }

extern "C" int dfl_head_scratch_off(int32_t F, int32_t which) {
  const int fc = head_fc(F);
  switch (which) {
    case 0: return 0;
    case 1: return fc;
    case 2: return fc + MAXNC;
    case 3: return fc + MAXNC + MAXNM;
    case 4: return fc + MAXNC + 2 * MAXNM;
    default: return -1;
  }
}

static unsigned head_grid(int64_t M) {
  int64_t b = ceil_div(M, 256);
  if (b > 4096) b = 4096;
  return (unsigned)(b < 1 ? 1 : b);
}

extern "C" int dfl_head_fwd(const dfl_head_fwd_args* a, dfl_stream_t stream) {
  DFL_REQUIRE(a && a->x && a->w_seg && a->seg, "dfl_head_fwd: missing pointer");
  DFL_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->ldx >= a->F, "dfl_head_fwd: bad sizes");
  int rc = head_check(a->F, a->NC, a->NM, a->L, a->w_l1, a->w_l2, a->ldx, a->x, a->x_bf16);
  if (rc != DFL_OK) return rc;
  DFL_REQUIRE(a->L == 0 || a->heat != nullptr, "dfl_head_fwd: heat output required when L > 0");
  const int64_t M = (int64_t)a->N * a->H * a->W;
#define DFL_HF(NC_, NM_, L_, G_) hipLaunchKernelGGL((head_fwd_kernel<NC_, NM_, L_, G_>), dim3(head_grid(M)), dim3(HT), \
    (size_t)HT * (a->F + 4) * sizeof(float), static_cast<hipStream_t>(stream), *a, a->w_seg, a->w_l1, a->w_l2)
  if (a->NC == 7 && a->L == 14 && a->NM == 21 && a->w_l2 != nullptr) DFL_HF(7, 21, 14, false);
  else if (a->NC == 7 && a->L == 0) DFL_HF(7, 0, 0, false);
  else DFL_HF(0, 0, 0, true);
#undef DFL_HF
  return check_launch("dfl_head_fwd");
}

static unsigned head_wgrad_grid(int64_t M) {
  int64_t b = ceil_div(M, HT);
  if (b > 512) b = 512;                     // two workgroups per CU walk the tiles; one partial each
  return (unsigned)(b < 1 ? 1 : b);
}

extern "C" int dfl_head_wgrad_blocks(int64_t M) {
  DFL_REQUIRE(M > 0, "dfl_head_wgrad_blocks: bad size");
  return (int)head_wgrad_grid(M);
}

extern "C" int dfl_head_bwd(const dfl_head_bwd_args* a, dfl_stream_t stream) {
  DFL_REQUIRE(a && a->x && a->seg && a->dseg && a->w_seg && a->dx && (a->scratch || a->dw_seg), "dfl_head_bwd: missing pointer");
  DFL_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->ldx >= a->F && a->lddx >= a->F, "dfl_head_bwd: bad sizes");
  int rc = head_check(a->F, a->NC, a->NM, a->L, a->w_l1, a->w_l2, a->ldx, a->x, a->x_bf16);
  if (rc != DFL_OK) return rc;
  DFL_REQUIRE(a->lddx % (a->x_bf16 ? 8 : 4) == 0 && aligned16(a->dx), "dfl_head_bwd: dx alignment");
  const bool fused = a->dw_seg != nullptr;
  const int64_t M = (int64_t)a->N * a->H * a->W;
  size_t lds = (size_t)HT * (a->F + 4) * sizeof(float);
  unsigned grid = head_grid(M);
  if (fused) {
    DFL_REQUIRE(a->x_bf16 && a->F == 32, "dfl_head_bwd: the fused weight gradients need bf16 features with F == 32 (F = %d)", a->F);
    DFL_REQUIRE(a->wg_partial != nullptr && aligned16(a->wg_partial), "dfl_head_bwd: wg_partial (dfl_head_wgrad_blocks(M) * 4096 floats) is required");
    DFL_REQUIRE((a->L == 0 || a->dw_l1 != nullptr) && ((a->L > 0 && a->w_l2 != nullptr) == (a->dw_l2 != nullptr)),
                "dfl_head_bwd: dw_l1 / dw_l2 must be given exactly for the landmark layers the head has");
    lds += 128 * 320;
    grid = head_wgrad_grid(M);
  } else {
    DFL_REQUIRE(a->scratch_ld >= dfl_head_scratch_ld(a->F) && a->scratch_ld % 4 == 0 && aligned16(a->scratch),
                "dfl_head_bwd: scratch_ld too small or misaligned");
  }
  hipStream_t hs = static_cast<hipStream_t>(stream);
#define DFL_HB(NC_, NM_, L_, G_)                                                                                              \
  do {                                                                                                                        \
    if (fused) {                                                                                                              \
      auto k = head_bwd_kernel<NC_, NM_, L_, G_, true>;                                                                       \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);     \
      hipLaunchKernelGGL(k, dim3(grid), dim3(HT), lds, hs, *a, head_fc(a->F), a->w_seg, a->w_l1, a->w_l2);                    \
    } else {                                                                                                                  \
      hipLaunchKernelGGL((head_bwd_kernel<NC_, NM_, L_, G_, false>), dim3(grid), dim3(HT), lds, hs, *a, head_fc(a->F), a->w_seg, a->w_l1, a->w_l2); \
    }                                                                                                                         \
  } while (0)
  if (a->NC == 7 && a->L == 14 && a->NM == 21 && a->w_l2 != nullptr) DFL_HB(7, 21, 14, false);
  else if (a->NC == 7 && a->L == 0) DFL_HB(7, 0, 0, false);
  else DFL_HB(0, 0, 0, true);
#undef DFL_HB
  int rc2 = check_launch("dfl_head_bwd");
  if (rc2 != DFL_OK || !fused) return rc2;
  hipLaunchKernelGGL(head_wgrad_finish_kernel, dim3(256), dim3(256), 0, hs, a->wg_partial, (int)grid, a->dw_seg, a->dw_l1, a->dw_l2, a->F,
                     a->NC, a->NM, a->L);
  return check_launch("dfl_head_bwd (weight-gradient sums)");
}
