// Output heads of the U-Net (reference: train_test_code/unet.py:176-191):
//   logits = seg_conv(x)                  1x1, no bias, F -> NC
//   seg    = Softmax2d(logits)            (or logits when do_soft_max=False)
//   heat   = lands_1x1(cat(x, logits))    1x1 (F+NC -> NM) [-> 1x1 (NM -> L)], no bias, no non-linearity
// One thread per pixel: the F features are streamed once as float4 (NHWC row); the head weights (5 KB) are read with
// wave-uniform addresses through `const __restrict__` kernel parameters, i.e. as scalar loads into SGPR operands of the
// FMAs (no LDS traffic: the LDS-broadcast version spent 224 ds_read_b128 per pixel), logits / mid / heat stay in registers, seg and heat are stored NCHW (lane = pixel
// => unit-stride stores per channel plane).  HBM-bound: 4*F bytes in, 4*(NC+L) bytes out per pixel.
// The backward kernel recomputes logits and mid from x (cheaper than saving them), applies the softmax Jacobian
//  and leaves dx plus a per-pixel scratch row from which the three small weight gradients are
// taken by dfl_conv2d_wgrad.
#include "common.h"

namespace dfl {

constexpr int MAXNC = DFL_HEAD_MAX_NC, MAXL = DFL_HEAD_MAX_L, MAXNM = DFL_HEAD_MAX_NM;

static inline int head_fc(int F) { return ((F + MAXNC) + 3) / 4 * 4; }

// logits (lg) and mid for pixel row xr; optionally copies x into `cat` (scratch row).  K1 = F + NC = row length of w_l1.
__device__ __forceinline__ void head_features(const float* __restrict__ xr, const float* __restrict__ w_seg,
                                              const float* __restrict__ w_l1, int F, int NC, int NM, bool lands,
                                              float* lg, float* mid, float* __restrict__ cat) {
  const int K1 = F + NC;
#pragma unroll
  for (int c = 0; c < MAXNC; ++c) lg[c] = 0.f;
#pragma unroll
  for (int j = 0; j < MAXNM; ++j) mid[j] = 0.f;
  for (int k = 0; k < F; k += 4) {
    const float4 xv = *reinterpret_cast<const float4*>(xr + k);
    if (cat != nullptr) *reinterpret_cast<float4*>(cat + k) = xv;
#pragma unroll
    for (int c = 0; c < MAXNC; ++c) {
      if (c < NC) {
        const float* w = w_seg + c * F + k;
        lg[c] = fmaf(w[0], xv.x, fmaf(w[1], xv.y, fmaf(w[2], xv.z, fmaf(w[3], xv.w, lg[c]))));
      }
    }
    if (lands) {
#pragma unroll
      for (int j = 0; j < MAXNM; ++j) {
        if (j < NM) {
          const float* w = w_l1 + j * K1 + k;
          mid[j] = fmaf(w[0], xv.x, fmaf(w[1], xv.y, fmaf(w[2], xv.z, fmaf(w[3], xv.w, mid[j]))));
        }
      }
    }
  }
  if (lands) {
#pragma unroll
    for (int j = 0; j < MAXNM; ++j) {
      if (j < NM) {
#pragma unroll
        for (int c = 0; c < MAXNC; ++c)
          if (c < NC) mid[j] = fmaf(w_l1[j * K1 + F + c], lg[c], mid[j]);
      }
    }
  }
}

__device__ __forceinline__ void softmax_inplace(float* lg, int NC) {
  float mx = lg[0];
#pragma unroll
  for (int c = 1; c < MAXNC; ++c)
    if (c < NC) mx = fmaxf(mx, lg[c]);
  float sum = 0.f;
#pragma unroll
  for (int c = 0; c < MAXNC; ++c) {
    if (c < NC) {
      lg[c] = expf(lg[c] - mx);
      sum += lg[c];
    }
  }
  const float inv = 1.0f / sum;
#pragma unroll
  for (int c = 0; c < MAXNC; ++c)
    if (c < NC) lg[c] *= inv;
}

__global__ void __launch_bounds__(256) head_fwd_kernel(const dfl_head_fwd_args a, const float* __restrict__ w_seg,
                                                      const float* __restrict__ w_l1, const float* __restrict__ w_l2) {
  const int F = a.F, NC = a.NC, NM = a.NM, L = a.L;
  const int64_t HW = (int64_t)a.H * a.W;
  const int64_t M = (int64_t)a.N * HW;
  const bool lands = L > 0;
  for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
    float lg[MAXNC], mid[MAXNM];
    head_features(a.x + m * a.ldx, w_seg, w_l1, F, NC, NM, lands, lg, mid, nullptr);
    const int64_t n = m / HW, pp = m - n * HW;
    if (lands) {
      float* hp = a.heat + n * L * HW + pp;
      if (w_l2 != nullptr) {
#pragma unroll
        for (int l = 0; l < MAXL; ++l) {
          if (l < L) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < MAXNM; ++j)
              if (j < NM) acc = fmaf(w_l2[l * NM + j], mid[j], acc);
            hp[(int64_t)l * HW] = acc;
          }
        }
      } else {
#pragma unroll
        for (int l = 0; l < MAXL; ++l)
          if (l < L) hp[(int64_t)l * HW] = mid[l];
      }
    }
    if (a.softmax) softmax_inplace(lg, NC);
    float* sp = a.seg + n * NC * HW + pp;
#pragma unroll
    for (int c = 0; c < MAXNC; ++c)
      if (c < NC) sp[(int64_t)c * HW] = lg[c];
  }
}

__global__ void __launch_bounds__(256) head_bwd_kernel(const dfl_head_bwd_args a, int Fc, const float* __restrict__ w_seg,
                                                      const float* __restrict__ w_l1, const float* __restrict__ w_l2) {
  const int F = a.F, NC = a.NC, NM = a.NM, L = a.L;
  const int K1 = F + NC;
  const int64_t HW = (int64_t)a.H * a.W;
  const int64_t M = (int64_t)a.N * HW;
  const bool lands = L > 0 && a.dheat != nullptr;
  const int o_dlg = Fc, o_dmid = Fc + MAXNC, o_mid = o_dmid + MAXNM, o_dh = o_mid + MAXNM;
  for (int64_t m = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; m < M; m += (int64_t)gridDim.x * blockDim.x) {
    float* sr = a.scratch + m * a.scratch_ld;
    float lg[MAXNC], mid[MAXNM];
    head_features(a.x + m * a.ldx, w_seg, w_l1, F, NC, NM, L > 0, lg, mid, sr);
    const int64_t n = m / HW, pp = m - n * HW;
    // cat tail: logits then zero pad
#pragma unroll
    for (int c = 0; c < MAXNC; ++c) sr[F + c] = (c < NC) ? lg[c] : 0.f;
    for (int c = F + MAXNC; c < Fc; ++c) sr[c] = 0.f;
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
          for (int l = 0; l < MAXL; ++l)
            if (l < L) acc = fmaf(w_l2[l * NM + j], dh[l], acc);
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
      float g = 0.f, s = 0.f;
      if (c < NC) {
        g = a.dseg[(n * NC + c) * HW + pp];
        s = a.seg[(n * NC + c) * HW + pp];
      }
      dlg[c] = g;      // provisional: upstream gradient
      lg[c] = s;       // reuse lg for the softmax output
      dot = fmaf(g, s, dot);
    }
#pragma unroll
    for (int c = 0; c < MAXNC; ++c) {
      float v = 0.f;
      if (c < NC) {
        v = a.softmax ? lg[c] * (dlg[c] - dot) : dlg[c];
        if (lands) {
#pragma unroll
          for (int j = 0; j < MAXNM; ++j)
            if (j < NM) v = fmaf(w_l1[j * K1 + F + c], dmid[j], v);
        }
      }
      dlg[c] = v;
      sr[o_dlg + c] = v;
    }
#pragma unroll
    for (int j = 0; j < MAXNM; ++j) {
      sr[o_dmid + j] = dmid[j];
      sr[o_mid + j] = (L > 0 && j < NM) ? mid[j] : 0.f;
    }
#pragma unroll
    for (int l = 0; l < MAXL; ++l) sr[o_dh + l] = dh[l];
    // dx = Wseg^T dlogits + W1[:, :F]^T dmid
    float* dxr = a.dx + m * a.lddx;
    for (int k = 0; k < F; k += 4) {
      float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int c = 0; c < MAXNC; ++c) {
        if (c < NC) {
          const float* w = w_seg + c * F + k;
          acc.x = fmaf(w[0], dlg[c], acc.x); acc.y = fmaf(w[1], dlg[c], acc.y);
          acc.z = fmaf(w[2], dlg[c], acc.z); acc.w = fmaf(w[3], dlg[c], acc.w);
        }
      }
      if (lands) {
#pragma unroll
        for (int j = 0; j < MAXNM; ++j) {
          if (j < NM) {
            const float* w = w_l1 + j * K1 + k;
            acc.x = fmaf(w[0], dmid[j], acc.x); acc.y = fmaf(w[1], dmid[j], acc.y);
            acc.z = fmaf(w[2], dmid[j], acc.z); acc.w = fmaf(w[3], dmid[j], acc.w);
          }
        }
      }
      *reinterpret_cast<float4*>(dxr + k) = acc;
    }
  }
}

static int head_check(int F, int NC, int NM, int L, const void* w_l1, const void* w_l2, int ldx, const void* x) {
  DFL_REQUIRE(F >= 4 && F % 4 == 0, "dfl_head: F must be a multiple of 4 (got %d)", F);
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
  int rc = head_check(a->F, a->NC, a->NM, a->L, a->w_l1, a->w_l2, a->ldx, a->x);
  if (rc != DFL_OK) return rc;
  DFL_REQUIRE(a->L == 0 || a->heat != nullptr, "dfl_head_fwd: heat output required when L > 0");
  const int64_t M = (int64_t)a->N * a->H * a->W;
  hipLaunchKernelGGL(head_fwd_kernel, dim3(head_grid(M)), dim3(256), 0, static_cast<hipStream_t>(stream), *a, a->w_seg,
                     a->w_l1, a->w_l2);
  return check_launch("dfl_head_fwd");
}

extern "C" int dfl_head_bwd(const dfl_head_bwd_args* a, dfl_stream_t stream) {
  DFL_REQUIRE(a && a->x && a->seg && a->dseg && a->w_seg && a->dx && a->scratch, "dfl_head_bwd: missing pointer");
  DFL_REQUIRE(a->N > 0 && a->H > 0 && a->W > 0 && a->ldx >= a->F && a->lddx >= a->F, "dfl_head_bwd: bad sizes");
  int rc = head_check(a->F, a->NC, a->NM, a->L, a->w_l1, a->w_l2, a->ldx, a->x);
  if (rc != DFL_OK) return rc;
  DFL_REQUIRE(a->lddx % 4 == 0 && aligned16(a->dx), "dfl_head_bwd: dx alignment");
  DFL_REQUIRE(a->scratch_ld >= dfl_head_scratch_ld(a->F) && a->scratch_ld % 4 == 0 && aligned16(a->scratch),
              "dfl_head_bwd: scratch_ld too small or misaligned");
  const int64_t M = (int64_t)a->N * a->H * a->W;
  hipLaunchKernelGGL(head_bwd_kernel, dim3(head_grid(M)), dim3(256), 0, static_cast<hipStream_t>(stream), *a,
                     head_fc(a->F), a->w_seg, a->w_l1, a->w_l2);
  return check_launch("dfl_head_bwd");
}
