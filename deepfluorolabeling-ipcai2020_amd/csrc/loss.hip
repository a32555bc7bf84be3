// Soft-Dice + NCC heat-map loss with closed-form gradients, and the ensemble reduction of test_ensemble.py.
// Reference: train_test_code/dice.py:14-86, train_test_code/ncc.py:12-38, train_test_code/util.py:326-373.
// Gradient formulas: SURVEY.md Appendix F (checked there against the reference's autograd in fp64).
// All three loss kernels are HBM-bound streams over the [B,C,h,w] windows; the per-(image,channel) sums are
// accumulated in fp64 (the reference sums in fp32 pairwise; fp64 is the tighter of the two).
#include "common.h"

namespace dfl {

constexpr double DICE_EPS = 1.0e-4;  // dice.py:24
constexpr double NCC_EPS = 1.0e-8;   // ncc.py:38

__device__ __forceinline__ void block_sum5(double* v, int nv, double (*red)[256]) {
  for (int k = 0; k < nv; ++k) red[k][threadIdx.x] = v[k];
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off)
      for (int k = 0; k < nv; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + off];
    __syncthreads();
  }
  for (int k = 0; k < nv; ++k) v[k] = red[k][0];
}

// blockIdx.x in [0, B*C): Dice sums (sum t*s, sum t*t, sum s*s) of one (image, class) plane; [B*C, B*C + B*L): NCC sums
// (x, y, xx, yy, xy) of one (image, landmark) plane; blockIdx.y = one of LOSS_NS row ranges of the plane (336 planes alone
// would leave 1.3 waves per SIMD).  A wave walks whole rows -- lane = 4 consecutive pixels, one float4 per tensor when the
// window is 16-byte aligned (the centre crop of a 192-wide row to 184 is: offset 4), two rows in flight -- so there is no
// index division per pixel.  Partial sums go to part[(plane * LOSS_NS + range)][k] (fp64); the finalize kernel adds the
// ranges in order.
constexpr int LOSS_NS = 4;

template <int NV>
__device__ __forceinline__ void loss_acc(const float* __restrict__ p, const float* __restrict__ q, int x, int w, bool vec, double* v) {
  float pv[4] = {0.f, 0.f, 0.f, 0.f}, qv[4] = {0.f, 0.f, 0.f, 0.f};
  if (vec) {
    const float4 a4 = *reinterpret_cast<const float4*>(p + x), b4 = *reinterpret_cast<const float4*>(q + x);
    pv[0] = a4.x; pv[1] = a4.y; pv[2] = a4.z; pv[3] = a4.w;
    qv[0] = b4.x; qv[1] = b4.y; qv[2] = b4.z; qv[3] = b4.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (x + e < w) {
        pv[e] = p[x + e];
        qv[e] = q[x + e];
      }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const double pd = (double)pv[e], qd = (double)qv[e];
    if (NV == 3) {            // p = seg, q = target
      v[0] += qd * pd; v[1] += qd * qd; v[2] += pd * pd;
    } else {                  // p = x, q = y
      v[0] += pd; v[1] += qd; v[2] += pd * pd; v[3] += qd * qd; v[4] += pd * qd;
    }
  }
}

__global__ void __launch_bounds__(256) loss_sums_kernel(const dfl_loss_args a, double* __restrict__ part) {
  __shared__ double red[5][256];
  const int BC = a.B * a.C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rows = (a.h + LOSS_NS - 1) / LOSS_NS;
  const int y0 = blockIdx.y * rows, y1 = min(y0 + rows, a.h);
  double v[5] = {0.0, 0.0, 0.0, 0.0, 0.0};
  const bool dice = (int)blockIdx.x < BC;
  const float *p, *q;
  int64_t pH, qH;
  if (dice) {
    const int n = blockIdx.x / a.C, c = blockIdx.x % a.C;
    p = a.seg + n * a.seg_sN + c * a.seg_sC;
    q = a.tseg + n * a.tseg_sN + c * a.tseg_sC;
    pH = a.seg_sH; qH = a.tseg_sH;
  } else {
    const int b = blockIdx.x - BC;
    const int n = b / a.L, l = b % a.L;
    p = a.heat + n * a.heat_sN + l * a.heat_sC;
    q = a.theat + n * a.theat_sN + l * a.theat_sC;
    pH = a.heat_sH; qH = a.theat_sH;
  }
  const bool vec = (a.w & 3) == 0 && (pH & 3) == 0 && (qH & 3) == 0 && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(q)) & 15) == 0;
  for (int x = 4 * lane; x < a.w; x += 256) {
    int y = y0 + wave;
    for (; y + 4 < y1; y += 8) {          // two rows of this wave in flight
      if (dice) {
        loss_acc<3>(p + y * pH, q + y * qH, x, a.w, vec, v);
        loss_acc<3>(p + (y + 4) * pH, q + (y + 4) * qH, x, a.w, vec, v);
      } else {
        loss_acc<5>(p + y * pH, q + y * qH, x, a.w, vec, v);
        loss_acc<5>(p + (y + 4) * pH, q + (y + 4) * qH, x, a.w, vec, v);
      }
    }
    for (; y < y1; y += 4) {
      if (dice) loss_acc<3>(p + y * pH, q + y * qH, x, a.w, vec, v);
      else loss_acc<5>(p + y * pH, q + y * qH, x, a.w, vec, v);
    }
  }
  block_sum5(v, dice ? 3 : 5, red);
  if (threadIdx.x == 0) {
    if (dice) {
      double* o = part + ((int64_t)blockIdx.x * LOSS_NS + blockIdx.y) * 3;
      o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
    } else {
      double* o = part + (int64_t)BC * LOSS_NS * 3 + ((int64_t)(blockIdx.x - BC) * LOSS_NS + blockIdx.y) * 5;
      for (int k = 0; k < 5; ++k) o[k] = v[k];
    }
  }
}

// Single workgroup: loss value + per-(image,channel) gradient coefficients.
__global__ void __launch_bounds__(256) loss_finalize_kernel(const dfl_loss_args a, const double* __restrict__ part) {
  __shared__ double red[2][256];
  const int BC = a.B * a.C, BL = a.B * a.L;
  const double* dpart = part;                                   // [BC][LOSS_NS][3]
  const double* npart = part + (int64_t)BC * LOSS_NS * 3;       // [BL][LOSS_NS][5]
  double* dcoef = a.sums + (int64_t)BC * 3 + (int64_t)BL * 5;  // [BC][2]
  double* ncoef = dcoef + (int64_t)BC * 2;                      // [BL][3]
  const int ceff = a.C - (a.skip_bg ? 1 : 0);
  double dice_acc = 0.0, ncc_acc = 0.0;
  if (a.dice_wgt != 0.f && a.C > 0) {
    const double g = (double)a.dice_wgt / ((double)a.B * (double)ceff);
    for (int i = threadIdx.x; i < BC; i += 256) {
      const int c = i % a.C;
      double a1 = 0.0, a2 = 0.0;
      if (!(a.skip_bg && c == 0)) {
        double I = 0.0, T = 0.0, S = 0.0;
        for (int r = 0; r < LOSS_NS; ++r) {
          I += dpart[(i * LOSS_NS + r) * 3 + 0];
          T += dpart[(i * LOSS_NS + r) * 3 + 1];
          S += dpart[(i * LOSS_NS + r) * 3 + 2];
        }
        const double num = -2.0 * I + DICE_EPS, den = T + S + DICE_EPS;
        dice_acc += num / den;
        a1 = -2.0 / den * g;
        a2 = -2.0 * num / (den * den) * g;
      }
      dcoef[i * 2 + 0] = a1;
      dcoef[i * 2 + 1] = a2;
    }
  }
  if (a.L > 0 && (a.heat_wgt != 0.f || a.ncc_vals != nullptr)) {
    const double N = (double)a.h * (double)a.w;
    const double q = -0.5 * (double)a.heat_wgt / (double)BL;
    for (int i = threadIdx.x; i < BL; i += 256) {
      double sx = 0.0, sy = 0.0, sxx = 0.0, syy = 0.0, sxy = 0.0;
      for (int r = 0; r < LOSS_NS; ++r) {
        const double* o = npart + (i * LOSS_NS + r) * 5;
        sx += o[0]; sy += o[1]; sxx += o[2]; syy += o[3]; sxy += o[4];
      }
      const double mx = sx / N, my = sy / N;
      double va = sxx - sx * mx, vb = syy - sy * my;  // sum of squared deviations
      if (va < 0.0) va = 0.0;
      if (vb < 0.0) vb = 0.0;
      const double cxy = sxy - sx * my;
      const double sdx = sqrt(va / (N - 1.0)), sdy = sqrt(vb / (N - 1.0));
      const double D = N * sdx * sdy + NCC_EPS;
      const double ncc = cxy / D;
      if (a.ncc_vals != nullptr) a.ncc_vals[i] = (float)ncc;
      ncc_acc += -(ncc + 1.0) * 0.5;
      const double E = cxy * N * sdy / (D * D * (N - 1.0) * sdx);
      const double k1 = q / D, k2 = -q * E;
      ncoef[i * 3 + 0] = k1;
      ncoef[i * 3 + 1] = k2;
      ncoef[i * 3 + 2] = -k1 * my - k2 * mx;
    }
  }
  red[0][threadIdx.x] = dice_acc;
  red[1][threadIdx.x] = ncc_acc;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
      red[0][threadIdx.x] += red[0][threadIdx.x + off];
      red[1][threadIdx.x] += red[1][threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double loss = 0.0;
    if (a.dice_wgt != 0.f && a.C > 0) loss += (double)a.dice_wgt * red[0][0] / ((double)a.B * (double)ceff);
    if (a.L > 0 && a.heat_wgt != 0.f) loss += (double)a.heat_wgt * red[1][0] / (double)BL;
    *a.loss = (float)loss;
  }
}

// Gradient stage: blockIdx.x = plane ((image, class) for the Dice part, then (image, landmark)), blockIdx.y = one of LOSS_NS
// row ranges, a wave per row, a lane per 4 pixels (float4 when the windows are 16-byte aligned) -- no index division per
// element.  dseg = gs * (a1 * t + a2 * s), dheat = gs * (k1 * y + k2 * x + k0) with the per-plane coefficients of the
// finalize kernel and gs = the incoming gradient of the loss.
__global__ void __launch_bounds__(256) loss_grad_kernel(const dfl_loss_args a) {
  const int BC = a.B * a.C, BL = a.B * a.L;
  const double* dcoef = a.sums + (int64_t)BC * 3 + (int64_t)BL * 5;
  const double* ncoef = dcoef + (int64_t)BC * 2;
  const float gs = a.grad_scale != nullptr ? *a.grad_scale : 1.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rows = (a.h + LOSS_NS - 1) / LOSS_NS;
  const int y0 = blockIdx.y * rows, y1 = min(y0 + rows, a.h);
  const bool dice = (int)blockIdx.x < BC;
  const float *p, *q;
  float* o;
  int64_t pH, qH, oH;
  float c1, c2, c0;
  if (dice) {
    if (a.dseg == nullptr) return;
    const int pc = blockIdx.x, n = pc / a.C, c = pc % a.C;
    p = a.seg + n * a.seg_sN + c * a.seg_sC;
    q = a.tseg + n * a.tseg_sN + c * a.tseg_sC;
    pH = a.seg_sH; qH = a.tseg_sH;
    const bool dense = a.dseg_sN == 0;
    o = a.dseg + (dense ? (int64_t)pc * a.h * a.w : n * a.dseg_sN + c * a.dseg_sC);
    oH = dense ? a.w : a.dseg_sH;
    c1 = gs * (float)dcoef[pc * 2 + 1];       // * s
    c2 = gs * (float)dcoef[pc * 2 + 0];       // * t
    c0 = 0.f;
  } else {
    if (a.dheat == nullptr || a.L == 0) return;
    const int pc = blockIdx.x - BC, n = pc / a.L, l = pc % a.L;
    p = a.heat + n * a.heat_sN + l * a.heat_sC;
    q = a.theat + n * a.theat_sN + l * a.theat_sC;
    pH = a.heat_sH; qH = a.theat_sH;
    const bool dense = a.dheat_sN == 0;
    o = a.dheat + (dense ? (int64_t)pc * a.h * a.w : n * a.dheat_sN + l * a.dheat_sC);
    oH = dense ? a.w : a.dheat_sH;
    c1 = gs * (float)ncoef[pc * 3 + 1];       // * x
    c2 = gs * (float)ncoef[pc * 3 + 0];       // * y
    c0 = gs * (float)ncoef[pc * 3 + 2];
  }
  const bool vec = (a.w & 3) == 0 && (pH & 3) == 0 && (qH & 3) == 0 && (oH & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(q) | reinterpret_cast<uintptr_t>(o)) & 15) == 0;
  for (int x = 4 * lane; x < a.w; x += 256) {
    for (int y = y0 + wave; y < y1; y += 4) {
      const float* pr = p + y * pH + x;
      const float* qr = q + y * qH + x;
      float* orow = o + y * oH + x;
      if (vec) {
        const float4 pv = *reinterpret_cast<const float4*>(pr), qv = *reinterpret_cast<const float4*>(qr);
        *reinterpret_cast<float4*>(orow) = make_float4(fmaf(c2, qv.x, fmaf(c1, pv.x, c0)), fmaf(c2, qv.y, fmaf(c1, pv.y, c0)),
                                                       fmaf(c2, qv.z, fmaf(c1, pv.z, c0)), fmaf(c2, qv.w, fmaf(c1, pv.w, c0)));
      } else {
        for (int e = 0; e < 4 && x + e < a.w; ++e) orow[e] = fmaf(c2, qr[e], fmaf(c1, pr[e], c0));
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ ensemble
// (round 4: one thread walked 1764 elements with two 64-bit divisions and ONE load in flight each -- 1.1 ms for the 577 MB of a
// 1436 x 1436 five-net image.  Now a wave takes whole rows of the cropped window, eight 256-byte loads in flight per wave.)
__global__ void __launch_bounds__(1024) ens_minmax_partial(const dfl_ensemble_args a, float* part, int nblk) {
  __shared__ float rmin[16], rmax[16];
  const int net = blockIdx.y;
  const float* hp = a.heat_ptrs[net];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
  const int rows = a.L * a.h;
  float mn = INFINITY, mx = -INFINITY;
  for (int r = (int)blockIdx.x * nwave + wave; r < rows; r += nblk * nwave) {
    const int l = r / a.h, y = r - l * a.h;
    const float* row = hp + ((int64_t)l * a.Hp + a.oy + y) * a.Wp + a.ox;
    for (int x0 = 0; x0 < a.w; x0 += 8 * 64) {
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int x = x0 + u * 64 + lane;
        v[u] = x < a.w ? row[x] : row[0];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        mn = fminf(mn, v[u]);
        mx = fmaxf(mx, v[u]);
      }
    }
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    mn = fminf(mn, __shfl_xor(mn, off, 64));
    mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  }
  if (lane == 0) {
    rmin[wave] = mn;
    rmax[wave] = mx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < nwave; ++w) {
      mn = fminf(mn, rmin[w]);
      mx = fmaxf(mx, rmax[w]);
    }
    part[((int64_t)net * nblk + blockIdx.x) * 2 + 0] = mn;
    part[((int64_t)net * nblk + blockIdx.x) * 2 + 1] = mx;
  }
}

__global__ void __launch_bounds__(256) ens_minmax_final(const float* part, float* minmax, int nblk) {
  __shared__ float rmin[256], rmax[256];
  const int net = blockIdx.x;
  float mn = INFINITY, mx = -INFINITY;
  for (int i = threadIdx.x; i < nblk; i += 256) {
    mn = fminf(mn, part[((int64_t)net * nblk + i) * 2 + 0]);
    mx = fmaxf(mx, part[((int64_t)net * nblk + i) * 2 + 1]);
  }
  rmin[threadIdx.x] = mn;
  rmax[threadIdx.x] = mx;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
      rmin[threadIdx.x] = fminf(rmin[threadIdx.x], rmin[threadIdx.x + off]);
      rmax[threadIdx.x] = fmaxf(rmax[threadIdx.x], rmax[threadIdx.x + off]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    minmax[net * 2 + 0] = rmin[0];
    minmax[net * 2 + 1] = rmax[0];
  }
}

// One thread per output pixel: mean over nets (summed in net order, then one division, as util.py:340-343,359),
// first-maximum argmax (torch.max(dim=1), util.py:361), min-max normalised heat maps (util.py:348-356,370).
constexpr int ENS_NU = 8;
__global__ void __launch_bounds__(256) ens_final(const dfl_ensemble_args a) {
  const int64_t hw = (int64_t)a.h * a.w;
  const int64_t pHW = (int64_t)a.Hp * a.Wp;
  const float fn = (float)a.nnets;
  const bool fast = a.nnets <= ENS_NU;
  const float* sp[ENS_NU];
  const float* hp[ENS_NU];
  float hmn[ENS_NU], hmx[ENS_NU];
#pragma unroll
  for (int u = 0; u < ENS_NU; ++u) {                  // (nets beyond nnets alias the last one: loaded, never added)
    const int k = u < a.nnets ? u : a.nnets - 1;
    sp[u] = a.seg_ptrs[k];
    hp[u] = a.heat_out != nullptr ? a.heat_ptrs[k] : nullptr;
    hmn[u] = (a.heat_out != nullptr && !a.raw_heat) ? a.minmax[k * 2 + 0] : 0.f;
    hmx[u] = (a.heat_out != nullptr && !a.raw_heat) ? a.minmax[k * 2 + 1] : 1.f;
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < hw; i += (int64_t)gridDim.x * blockDim.x) {
    const int y = (int)(i / a.w), x = (int)(i - (int64_t)y * a.w);
    const int64_t po = (int64_t)(a.oy + y) * a.Wp + a.ox + x;
    // (nnets <= ENS_NU, the scripts' case: the nets' base pointers sit in registers and the values of all nets are requested
    // before the first is used -- with a run-time net count in the inner loop ONE load was in flight per thread.  Sums stay in net order.)
    float best = 0.f;
    int arg = 0;
    for (int c = 0; c < a.C; ++c) {
      float s;
      if (fast) {
        float v[ENS_NU];
#pragma unroll
        for (int u = 0; u < ENS_NU; ++u) v[u] = sp[u][c * pHW + po];
        s = v[0];
#pragma unroll
        for (int u = 1; u < ENS_NU; ++u)
          if (u < a.nnets) s += v[u];
      } else {
        s = a.seg_ptrs[0][c * pHW + po];
        for (int k = 1; k < a.nnets; ++k) s += a.seg_ptrs[k][c * pHW + po];
      }
      s = s / fn;
      if (a.avg_seg != nullptr) a.avg_seg[c * hw + i] = s;
      if (c == 0 || s > best) {
        best = s;
        arg = c;
      }
    }
    a.labels[i] = (uint8_t)arg;
    if (a.heat_out != nullptr) {
      for (int l = 0; l < a.L; ++l) {
        float s = 0.f;
        if (fast) {
          float v[ENS_NU];
#pragma unroll
          for (int u = 0; u < ENS_NU; ++u) v[u] = hp[u][l * pHW + po];
#pragma unroll
          for (int u = 0; u < ENS_NU; ++u) {
            if (u < a.nnets) {
              float t = v[u];
              if (!a.raw_heat) t = (t - hmn[u]) / (hmx[u] - hmn[u]);
              s = (u == 0) ? t : s + t;
            }
          }
        } else {
          for (int k = 0; k < a.nnets; ++k) {
            float v = a.heat_ptrs[k][l * pHW + po];
            if (!a.raw_heat) {
              const float mn = a.minmax[k * 2 + 0], mx = a.minmax[k * 2 + 1];
              v = (v - mn) / (mx - mn);
            }
            s = (k == 0) ? v : s + v;
          }
        }
        a.heat_out[l * hw + i] = s / fn;
      }
    }
  }
}

}  // namespace dfl

using namespace dfl;

extern "C" int64_t dfl_loss_scratch_doubles(int32_t B, int32_t C, int32_t L) {
  return (int64_t)B * C * 5 + (int64_t)B * L * 8 + 8 + (int64_t)LOSS_NS * ((int64_t)B * C * 3 + (int64_t)B * L * 5);
}

extern "C" int dfl_dice_ncc_loss(const dfl_loss_args* a, dfl_stream_t stream) {
  DFL_REQUIRE(a && a->loss && a->sums, "dfl_dice_ncc_loss: loss and sums are required");
  DFL_REQUIRE(a->B > 0 && a->C >= 0 && a->L >= 0 && a->C + a->L > 0 && a->h > 0 && a->w > 0, "dfl_dice_ncc_loss: bad sizes");
  DFL_REQUIRE(a->C == 0 || (a->seg && a->tseg), "dfl_dice_ncc_loss: seg and tseg are required when C > 0");
  DFL_REQUIRE(a->L == 0 || (a->heat && a->theat), "dfl_dice_ncc_loss: heat and theat are required when L > 0");
  DFL_REQUIRE(a->L == 0 || (int64_t)a->h * a->w > 1, "dfl_dice_ncc_loss: NCC needs more than one pixel");
  DFL_REQUIRE(!(a->skip_bg && a->C < 2), "dfl_dice_ncc_loss: skip_bg needs at least 2 classes");
  hipStream_t s = static_cast<hipStream_t>(stream);
  DFL_REQUIRE(a->stage >= 0 && a->stage <= 2, "dfl_dice_ncc_loss: stage must be 0, 1 or 2");
  DFL_REQUIRE((a->dseg_sN == 0) == (a->dseg_sH == 0) && (a->dheat_sN == 0) == (a->dheat_sH == 0), "dfl_dice_ncc_loss: give all gradient strides or none");
  double* part = a->sums + (int64_t)a->B * a->C * 5 + (int64_t)a->B * a->L * 8 + 8;     // row-range partial sums behind the coefficients
  if (a->stage != 2) {
    hipLaunchKernelGGL(loss_sums_kernel, dim3((unsigned)(a->B * (a->C + a->L)), (unsigned)LOSS_NS), dim3(256), 0, s, *a, part);
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, *a, part);
  }
  if (a->stage != 1 && (a->dseg != nullptr || (a->dheat != nullptr && a->L > 0)))
    hipLaunchKernelGGL(loss_grad_kernel, dim3((unsigned)(a->B * (a->C + a->L)), (unsigned)LOSS_NS), dim3(256), 0, s, *a);
  return check_launch("dfl_dice_ncc_loss");
}

extern "C" int dfl_ensemble_reduce(const dfl_ensemble_args* a, dfl_stream_t stream) {
  DFL_REQUIRE(a && a->seg_ptrs && a->labels, "dfl_ensemble_reduce: missing pointer");
  DFL_REQUIRE(a->nnets > 0 && a->C > 0 && a->C <= 255 && a->h > 0 && a->w > 0, "dfl_ensemble_reduce: bad sizes");
  DFL_REQUIRE(a->oy >= 0 && a->ox >= 0 && a->oy + a->h <= a->Hp && a->ox + a->w <= a->Wp, "dfl_ensemble_reduce: crop window");
  DFL_REQUIRE(a->heat_out == nullptr || (a->heat_ptrs && (a->minmax || a->raw_heat) && a->L > 0), "dfl_ensemble_reduce: heat inputs");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a->heat_out != nullptr && !a->raw_heat) {
    // the partial min/max rows live behind the final [nnets][2] table in the same scratch
    const int nblk = 64;
    float* part = a->minmax + 2 * a->nnets;
    hipLaunchKernelGGL(ens_minmax_partial, dim3(nblk, (unsigned)a->nnets), dim3(1024), 0, s, *a, part, nblk);
    hipLaunchKernelGGL(ens_minmax_final, dim3((unsigned)a->nnets), dim3(256), 0, s, part, a->minmax, nblk);
  }
  int64_t blocks = ceil_div((int64_t)a->h * a->w, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(ens_final, dim3((unsigned)blocks), dim3(256), 0, s, *a);
  return check_launch("dfl_ensemble_reduce");
}
