// GPU-side input pipeline: what the reference's loader computes on the host per item
// (train_test_code/dataset.py:287-293 reflect pad + standardisation, :302-325 Gaussian heat maps, :421-429 out-of-view
// landmarks, :448-452 one-hot masks) done on the device from the raw arrays, so a training step ships u8 labels and
// 2*L floats of landmarks per image instead of float targets (45 MB per batch-16 step).  Contract: include/dfl_hip.h.
// HBM-bound: one read of the raw image + labels, one write of x / masks / heats.
#include "common.h"

namespace dfl {

constexpr int PREP_NB = 64;   // statistics slices per image

__device__ __forceinline__ int reflect(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

// partial[b][blk][2] = (sum, sum of squares) of the reflect-padded image, fp64
__global__ void __launch_bounds__(256) prep_stats_kernel(const dfl_prep_args a, double* __restrict__ partial) {
  __shared__ double red[2][256];
  const int b = blockIdx.y;
  const int Hp = a.H + 2 * a.pad, Wp = a.W + 2 * a.pad;
  const int64_t n = (int64_t)Hp * Wp;
  const float* src = a.proj + (int64_t)b * a.H * a.W;
  double s1 = 0.0, s2 = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)PREP_NB * 256) {
    const int y = (int)(i / Wp), x = (int)(i - (int64_t)y * Wp);
    const double v = (double)src[(int64_t)reflect(y - a.pad, a.H) * a.W + reflect(x - a.pad, a.W)];
    s1 += v;
    s2 += v * v;
  }
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) {
      red[0][threadIdx.x] += red[0][threadIdx.x + off];
      red[1][threadIdx.x] += red[1][threadIdx.x + off];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[((int64_t)b * PREP_NB + blockIdx.x) * 2 + 0] = red[0][0];
    partial[((int64_t)b * PREP_NB + blockIdx.x) * 2 + 1] = red[1][0];
  }
}

// blockIdx.z: 0 = standardised padded image, 1 = one-hot masks, 2 = heat maps; blockIdx.y = image
__global__ void __launch_bounds__(256) prep_write_kernel(const dfl_prep_args a, const double* __restrict__ partial) {
  const int b = blockIdx.y;
  const int64_t stride = (int64_t)gridDim.x * 256;
  const int64_t t0 = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (blockIdx.z == 0) {
    if (a.x == nullptr) return;
    const int Hp = a.H + 2 * a.pad, Wp = a.W + 2 * a.pad;
    const int64_t n = (int64_t)Hp * Wp;
    float mean = 0.f, inv = 1.f;
    if (a.standardize) {
      double s1 = 0.0, s2 = 0.0;
      for (int k = 0; k < PREP_NB; ++k) {   // same order in every thread: bit-identical statistics
        s1 += partial[((int64_t)b * PREP_NB + k) * 2 + 0];
        s2 += partial[((int64_t)b * PREP_NB + k) * 2 + 1];
      }
      const double m = s1 / (double)n;
      double var = (s2 - s1 * m) / (double)(n - 1);   // torch.std: unbiased
      if (var < 0.0) var = 0.0;
      mean = (float)m;
      inv = (float)(1.0 / sqrt(var));
    }
    const float* src = a.proj + (int64_t)b * a.H * a.W;
    float* dst = a.x + (int64_t)b * n;
    for (int64_t i = t0; i < n; i += stride) {
      const int y = (int)(i / Wp), x = (int)(i - (int64_t)y * Wp);
      dst[i] = (src[(int64_t)reflect(y - a.pad, a.H) * a.W + reflect(x - a.pad, a.W)] - mean) * inv;
    }
  } else if (blockIdx.z == 1) {
    if (a.masks == nullptr) return;
    const int64_t hw = (int64_t)a.H * a.W;
    const unsigned char* lab = a.labels + (int64_t)b * hw;
    float* dst = a.masks + (int64_t)b * a.C * hw;
    for (int64_t i = t0; i < hw; i += stride) {
      const int l = lab[i];
      for (int c = 0; c < a.C; ++c) dst[(int64_t)c * hw + i] = (l == c) ? 1.f : 0.f;
    }
  } else {
    if (a.heats == nullptr) return;
    const int64_t hw = (int64_t)a.H * a.W;
    const float* ld = a.lands + (int64_t)b * 2 * a.L;
    float* dst = a.heats + (int64_t)b * a.L * hw;
    const float s2 = a.sigma * a.sigma;
    const float kexp = 1.f / (s2 * -2.f), knorm = 1.f / (2.f * 3.14159265358979323846f * s2);
    for (int l = 0; l < a.L; ++l) {
      const float mx = ld[l], my = ld[a.L + l];
      // landmarks outside the view (or already marked inf) give an all-zero map
      const bool ok = !(isinf(mx) || isinf(my) || isnan(mx) || isnan(my)) && mx >= 0.f && mx <= (float)(a.W - 1) &&
                      my >= 0.f && my <= (float)(a.H - 1);
      for (int64_t i = t0; i < hw; i += stride) {
        const int y = (int)(i / a.W), x = (int)(i - (int64_t)y * a.W);
        const float dx = (float)x - mx, dy = (float)y - my;
        dst[(int64_t)l * hw + i] = ok ? expf((dx * dx + dy * dy) * kexp) * knorm : 0.f;
      }
    }
  }
}

}  // namespace dfl

extern "C" int64_t dfl_prep_scratch_doubles(int32_t B) { return (int64_t)B * dfl::PREP_NB * 2; }

extern "C" int dfl_prep_batch(const dfl_prep_args* a, dfl_stream_t stream) {
  DFL_REQUIRE(a != nullptr, "dfl_prep_batch: null args");
  DFL_REQUIRE(a->B > 0 && a->H > 0 && a->W > 0 && a->pad >= 0, "dfl_prep_batch: bad sizes");
  DFL_REQUIRE(a->pad < a->H && a->pad < a->W, "dfl_prep_batch: reflect padding needs pad < H, W");
  DFL_REQUIRE(a->x == nullptr || a->proj != nullptr, "dfl_prep_batch: x needs proj");
  DFL_REQUIRE(a->masks == nullptr || (a->labels != nullptr && a->C > 0), "dfl_prep_batch: masks need labels and C");
  DFL_REQUIRE(a->heats == nullptr || (a->lands != nullptr && a->L > 0 && a->sigma > 0.f), "dfl_prep_batch: heats need lands, L, sigma");
  DFL_REQUIRE(!(a->x != nullptr && a->standardize) || a->scratch != nullptr, "dfl_prep_batch: standardisation needs scratch");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a->x != nullptr && a->standardize) {
    hipLaunchKernelGGL(dfl::prep_stats_kernel, dim3(dfl::PREP_NB, (unsigned)a->B), dim3(256), 0, s, *a, a->scratch);
    int rc = dfl::check_launch("dfl_prep_batch");
    if (rc != DFL_OK) return rc;
  }
  const int64_t n = (int64_t)(a->H + 2 * a->pad) * (a->W + 2 * a->pad);
  int64_t gx = dfl::ceil_div(n, 256 * 4);
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(dfl::prep_write_kernel, dim3((unsigned)gx, (unsigned)a->B, 3), dim3(256), 0, s, *a, a->scratch);
  return dfl::check_launch("dfl_prep_batch");
}

// ------------------------------------------------------------------------------------------------------------------
// Landmark extraction (est_lands_csv.py:96-124): per (image, landmark) the arg-max of the heat map,
// optionally restricted to the pixels of one segmentation label, accepted only if the 25x25 window around it (heat map
// reflect-padded by 12) correlates with the Gaussian template (ncc_2d, ncc.py:12-38) at >= 0.9.  One workgroup per
// (landmark, image); ties resolve to the lowest flat index like torch.argmax.
namespace dfl {

constexpr int LM_T = 25, LM_R = 12, LM_N = LM_T * LM_T;

__device__ __forceinline__ double block_sum(double v, double* red) {
  red[threadIdx.x] = v;
  __syncthreads();
  for (int off = 128; off >= 1; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  const double r = red[0];
  __syncthreads();
  return r;
}

// The arg-max pass (round 4: ONE workgroup per map walked it with one load in flight per thread -- 4.5 ms for the fourteen
// 1436 x 1436 maps of an image on 14 of 256 CUs).  LM_SCAN_NB workgroups per map now, and the map's (row, col) output slot --
// 8 bytes, zeroed by the launcher -- doubles as the meeting point: an unsigned 64-bit atomic max over
// (order-preserving bits of the value) << 32 | (2^32 - 1 - flat index) keeps the largest value and, among equals, the lowest
// index, exactly what the one-workgroup loop kept.  est_lands_kernel decodes the slot before it writes its answer there.
constexpr int LM_SCAN_NB = 64;
__device__ __forceinline__ unsigned lm_key(float v) {          // monotone in v (no NaN); +0 and -0 are the same value
  if (v == 0.f) v = 0.f;
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float lm_unkey(unsigned k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }

__global__ void __launch_bounds__(256) est_lands_scan_kernel(const dfl_est_lands_args a) {
  const int l = blockIdx.y, b = blockIdx.z;
  const int64_t hw = (int64_t)a.H * a.W;
  const float* heat = a.heats + ((int64_t)b * a.L + l) * hw;
  const int want = (a.segs != nullptr && a.label_for_land != nullptr) ? a.label_for_land[l] : -1;
  const unsigned char* seg = (want >= 0) ? a.segs + (int64_t)b * hw : nullptr;
  float best = -INFINITY;
  int besti = 0x7fffffff;
  constexpr int U = 8;
  for (int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x; i0 < hw; i0 += (int64_t)LM_SCAN_NB * 256 * U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t i = i0 + (int64_t)u * LM_SCAN_NB * 256;
      v[u] = -INFINITY;
      if (i < hw && (seg == nullptr || seg[i] == want)) v[u] = heat[i];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = (int)(i0 + (int64_t)u * LM_SCAN_NB * 256);   // (ascending within a thread: '>' keeps the lowest index of equals)
      if (v[u] > best) {
        best = v[u];
        besti = i;
      }
    }
  }
  // (every lane takes part in the butterflies: a lane without a candidate -- all of its values -inf or NaN -- carries key 0)
  unsigned long long m = 0ull;
  if (best > -INFINITY) m = ((unsigned long long)lm_key(best) << 32) | (unsigned long long)(0xffffffffu - (unsigned)besti);
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long o = __shfl_xor(m, off, 64);
    m = o > m ? o : m;
  }
  if ((threadIdx.x & 63) == 0 && m != 0ull) atomicMax(reinterpret_cast<unsigned long long*>(a.rowcol) + (int64_t)b * a.L + l, m);
}

__global__ void __launch_bounds__(256) est_lands_kernel(const dfl_est_lands_args a) {
  __shared__ double red[256];
  const int l = blockIdx.x, b = blockIdx.y;
  const int H = a.H, W = a.W;
  const int64_t hw = (int64_t)H * W;
  const float* heat = a.heats + ((int64_t)b * a.L + l) * hw;
  const unsigned long long key = reinterpret_cast<const unsigned long long*>(a.rowcol)[(int64_t)b * a.L + l];
  const float best = key != 0ull ? lm_unkey((unsigned)(key >> 32)) : -INFINITY;
  const int besti = key != 0ull ? (int)(0xffffffffu - (unsigned)(key & 0xffffffffull)) : 0x7fffffff;
  __syncthreads();                                     // (every thread has read the slot before thread 0 writes the answer into it)
  int* out = a.rowcol + ((int64_t)b * a.L + l) * 2;
  float* ncc_out = (a.ncc != nullptr) ? a.ncc + (int64_t)b * a.L + l : nullptr;
  if (!(best > -INFINITY)) {   // nothing inside the label (or an all -inf / NaN map)
    if (threadIdx.x == 0) {
      out[0] = -1;
      out[1] = -1;
      if (ncc_out) *ncc_out = 0.f;
    }
    return;
  }
  const int row = besti / W, col = besti - row * W;
  // window values y (reflect padded) and template t, fp32 like the reference; sums in fp64
  float y[3], t[3];
  double sy = 0.0, st = 0.0;
  const float s2 = a.sigma * a.sigma;
  const float kexp = 1.f / (s2 * -2.f), knorm = 1.f / (2.f * 3.14159265358979323846f * s2);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int e = threadIdx.x + 256 * k;
    y[k] = 0.f;
    t[k] = 0.f;
    if (e < LM_N) {
      const int wy = e / LM_T, wx = e - wy * LM_T;
      y[k] = heat[(int64_t)reflect(row - LM_R + wy, H) * W + reflect(col - LM_R + wx, W)];
      const float dx = (float)(wx - LM_R), dy = (float)(wy - LM_R);
      t[k] = expf((dx * dx + dy * dy) * kexp) * knorm;
      sy += (double)y[k];
      st += (double)t[k];
    }
  }
  const float my = (float)(block_sum(sy, red) / LM_N), mt = (float)(block_sum(st, red) / LM_N);
  double syy = 0.0, stt = 0.0, sty = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if ((int)threadIdx.x + 256 * k < LM_N) {
      const float yz = y[k] - my, tz = t[k] - mt;
      syy += (double)(yz * yz);
      stt += (double)(tz * tz);
      sty += (double)(tz * yz);
    }
  }
  syy = block_sum(syy, red);
  stt = block_sum(stt, red);
  sty = block_sum(sty, red);
  if (threadIdx.x == 0) {
    const float sdy = sqrtf((float)(syy / (LM_N - 1))), sdt = sqrtf((float)(stt / (LM_N - 1)));
    const float ncc = (float)sty / ((float)LM_N * (sdt * sdy) + 1.0e-8f);
    const bool ok = !(ncc < a.min_ncc);
    out[0] = ok ? row : -1;
    out[1] = ok ? col : -1;
    if (ncc_out) *ncc_out = ncc;
  }
}

}  // namespace dfl

extern "C" int dfl_est_lands(const dfl_est_lands_args* a, dfl_stream_t stream) {
  DFL_REQUIRE(a != nullptr && a->heats != nullptr && a->rowcol != nullptr, "dfl_est_lands: missing pointer");
  DFL_REQUIRE(a->B > 0 && a->L > 0 && a->H > dfl::LM_R && a->W > dfl::LM_R, "dfl_est_lands: maps must be larger than the 12-pixel reflect border");
  DFL_REQUIRE((int64_t)a->H * a->W < (1ll << 31), "dfl_est_lands: map too large");
  DFL_REQUIRE(a->sigma > 0.f, "dfl_est_lands: sigma");
  DFL_REQUIRE((reinterpret_cast<uintptr_t>(a->rowcol) & 7) == 0, "dfl_est_lands: rowcol must be 8-byte aligned");
  hipStream_t hs = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(a->rowcol, 0, (size_t)a->B * a->L * 8, hs) != hipSuccess) {
    dfl::set_error("dfl_est_lands: hipMemsetAsync failed");
    return DFL_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(dfl::est_lands_scan_kernel, dim3((unsigned)dfl::LM_SCAN_NB, (unsigned)a->L, (unsigned)a->B), dim3(256), 0, hs, *a);
  hipLaunchKernelGGL(dfl::est_lands_kernel, dim3((unsigned)a->L, (unsigned)a->B), dim3(256), 0, hs, *a);
  return dfl::check_launch("dfl_est_lands");
}

// ------------------------------------------------------------------------------------------------------------------
// Hard Dice of label maps (compute_actual_dice_on_test.py:63-93): per image and label the pixel counts
// |est == l|, |gt == l|, |both|; dice = 2*both / (est + gt), 1.0 when the label is absent from both.  Integer counting
// (LDS + global integer atomics): bit-exact and order independent.
namespace dfl {

__global__ void __launch_bounds__(256) hard_dice_count_kernel(const unsigned char* __restrict__ est,
                                                              const unsigned char* __restrict__ gt, int64_t hw, int C,
                                                              unsigned long long* __restrict__ counts) {
  __shared__ unsigned int loc[256 * 3];
  const int b = blockIdx.y;
  for (int i = threadIdx.x; i < 3 * C; i += 256) loc[i] = 0u;
  __syncthreads();
  const unsigned char* e = est + (int64_t)b * hw;
  const unsigned char* g = gt + (int64_t)b * hw;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (int64_t)gridDim.x * 256) {
    const int le = e[i], lg = g[i];
    if (le < C) atomicAdd(&loc[le * 3 + 0], 1u);
    if (lg < C) atomicAdd(&loc[lg * 3 + 1], 1u);
    if (le == lg && le < C) atomicAdd(&loc[le * 3 + 2], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 3 * C; i += 256)
    if (loc[i] != 0u) atomicAdd(&counts[(int64_t)b * C * 3 + i], (unsigned long long)loc[i]);
}

__global__ void hard_dice_final_kernel(const unsigned long long* __restrict__ counts, double* __restrict__ dice, int B, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * (C - 1)) return;
  const int b = i / (C - 1), l = 1 + i % (C - 1);   // background excluded
  const unsigned long long* c = counts + ((int64_t)b * C + l) * 3;
  const double tot = (double)c[0] + (double)c[1];
  dice[i] = tot > 0.1 ? 2.0 * (double)c[2] / tot : 1.0;
}

}  // namespace dfl

extern "C" int dfl_hard_dice(const unsigned char* est, const unsigned char* gt, int64_t pixels_per_image, int32_t B,
                             int32_t C, int64_t* counts, double* dice, dfl_stream_t stream) {
  DFL_REQUIRE(est && gt && counts && dice, "dfl_hard_dice: missing pointer");
  DFL_REQUIRE(B > 0 && C >= 2 && C <= 256 && pixels_per_image > 0, "dfl_hard_dice: bad sizes");
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (hipMemsetAsync(counts, 0, (size_t)B * C * 3 * sizeof(int64_t), s) != hipSuccess) {
    dfl::set_error("dfl_hard_dice: hipMemsetAsync failed");
    return DFL_ERR_LAUNCH;
  }
  int64_t gx = dfl::ceil_div(pixels_per_image, 256 * 16);
  if (gx > 256) gx = 256;
  hipLaunchKernelGGL(dfl::hard_dice_count_kernel, dim3((unsigned)gx, (unsigned)B), dim3(256), 0, s, est, gt,
                     pixels_per_image, (int)C, reinterpret_cast<unsigned long long*>(counts));
  int rc = dfl::check_launch("dfl_hard_dice");
  if (rc != DFL_OK) return rc;
  const int n = B * (C - 1);
  hipLaunchKernelGGL(dfl::hard_dice_final_kernel, dim3((unsigned)dfl::ceil_div(n, 64)), dim3(64), 0, s,
                     reinterpret_cast<const unsigned long long*>(counts), dice, (int)B, (int)C);
  return dfl::check_launch("dfl_hard_dice");
}
