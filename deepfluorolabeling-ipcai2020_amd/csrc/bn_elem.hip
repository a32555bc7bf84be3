// Streaming (HBM-bound) kernels around the convolutions: BatchNorm finalize / backward, per-channel reductions,
// affine copy, 2x2 max-pool, weight re-layout, SGD.  All operate on NHWC rows [M][C] with float4 (= 4 channels)
// accesses when C, ld and the pointers allow it, 4-byte accesses otherwise.
// Reference behaviour: nn.BatchNorm2d (train_test_code/unet.py:215,222), nn.ReLU (:213,220), F.max_pool2d (:169),
// torch.optim.SGD (train.py:333-334).  Contracts: include/dfl_hip.h.
#include "common.h"

namespace dfl {

// ------------------------------------------------------------------------------------------------ helpers
struct RowGeom {
  int vec;    // 1: float4 units
  int units;  // units per row (C/4 or C)
  int UX;     // threads along the row (power of two <= 256)
  int RY;     // rows per pass = 256 / UX
  int gy;     // grid.y = ceil(units / UX)
};

static RowGeom row_geom(int C, bool vec_ok) {
  RowGeom g;
  g.vec = (vec_ok && C % 4 == 0) ? 1 : 0;
  g.units = g.vec ? C / 4 : C;
  int ux = 1;
  while (ux * 2 <= g.units && ux * 2 <= 256) ux *= 2;
  g.UX = ux;
  g.RY = 256 / ux;
  g.gy = (int)ceil_div(g.units, ux);
  return g;
}

static int rowblocks(int64_t M, int C) {
  int64_t nb = ceil_div(M * (int64_t)C, 8192);
  if (nb > 2048) nb = 2048;
  if (nb > M) nb = M;
  if (nb < 1) nb = 1;
  return (int)nb;
}

__device__ __forceinline__ float4 ld4(const float* p, bool vec, int c, int C) {
  if (vec) return *reinterpret_cast<const float4*>(p);
  (void)c; (void)C;
  return make_float4(p[0], 0.f, 0.f, 0.f);
}

// ------------------------------------------------------------------------------------------------ colstats
// partials[blk][0][c] = sum a, partials[blk][1][c] = sum a*b over the block's rows.
template <int VEC>
__global__ void __launch_bounds__(256) colstats_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ partials, int64_t M, int C, int lda, int ldb,
                                                      int UX, int rows_per_block) {
  __shared__ float red[2][256][VEC ? 4 : 1];
  const int RY = 256 / UX;
  const int ux = threadIdx.x % UX, uy = threadIdx.x / UX;
  const int unit = blockIdx.y * UX + ux;
  const int c = VEC ? unit * 4 : unit;
  const bool cok = c < C;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > M) r1 = M;
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  if (cok) {
    for (int64_t r = r0 + uy; r < r1; r += RY) {
      if constexpr (VEC) {
        const float4 va = *reinterpret_cast<const float4*>(a + r * lda + c);
        const float4 vb = b ? *reinterpret_cast<const float4*>(b + r * ldb + c) : va;
        s1[0] += va.x; s1[1] += va.y; s1[2] += va.z; s1[3] += va.w;
        s2[0] = fmaf(va.x, vb.x, s2[0]); s2[1] = fmaf(va.y, vb.y, s2[1]);
        s2[2] = fmaf(va.z, vb.z, s2[2]); s2[3] = fmaf(va.w, vb.w, s2[3]);
      } else {
        const float va = a[r * lda + c];
        const float vb = b ? b[r * ldb + c] : va;
        s1[0] += va;
        s2[0] = fmaf(va, vb, s2[0]);
      }
    }
  }
  constexpr int W = VEC ? 4 : 1;
#pragma unroll
  for (int j = 0; j < W; ++j) {
    red[0][threadIdx.x][j] = s1[j];
    red[1][threadIdx.x][j] = s2[j];
  }
  __syncthreads();
  if (uy == 0 && cok) {
#pragma unroll
    for (int j = 0; j < W; ++j) {
      float t1 = 0.f, t2 = 0.f;
      for (int y = 0; y < RY; ++y) {
        t1 += red[0][y * UX + ux][j];
        t2 += red[1][y * UX + ux][j];
      }
      if (c + j < C) {
        partials[((int64_t)blockIdx.x * 2 + 0) * C + c + j] = t1;
        partials[((int64_t)blockIdx.x * 2 + 1) * C + c + j] = t2;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ BN finalize
// One workgroup per `cpb` channels (8 for wide layers down to 1 for narrow ones, so that even a 32-channel layer with
// thousands of partial rows spreads over tens of workgroups): 256/cpb lanes walk the partial rows, fp64 tree over them.
__device__ __forceinline__ void sum_partials_f64(const float* __restrict__ partials, int nblocks, int C, int cpb,
                                                 double* out1, double* out2, double (*red)[256]) {
  const int cl = threadIdx.x % cpb, rl = threadIdx.x / cpb, lanes = 256 / cpb;
  const int c = blockIdx.x * cpb + cl;
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    // four rows (eight loads) in flight per lane: these kernels are one memory round trip after the other, nothing else
    int r = rl;
    for (; r + 3 * lanes < nblocks; r += 4 * lanes) {
      float v1[4], v2[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        v1[u] = partials[((int64_t)(r + u * lanes) * 2 + 0) * C + c];
        v2[u] = partials[((int64_t)(r + u * lanes) * 2 + 1) * C + c];
      }
      s1 += ((double)v1[0] + (double)v1[1]) + ((double)v1[2] + (double)v1[3]);
      s2 += ((double)v2[0] + (double)v2[1]) + ((double)v2[2] + (double)v2[3]);
    }
    for (; r < nblocks; r += lanes) {
      s1 += (double)partials[((int64_t)r * 2 + 0) * C + c];
      s2 += (double)partials[((int64_t)r * 2 + 1) * C + c];
    }
  }
  // lanes of one channel inside a wave (lane = row lane * cpb + channel: xor offsets >= cpb keep the channel), then the four waves
  for (int off = 32; off >= cpb; off >>= 1) {
    s1 += __shfl_xor(s1, off, 64);
    s2 += __shfl_xor(s2, off, 64);
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (lane < cpb) {
    red[0][wave * 8 + lane] = s1;
    red[1][wave * 8 + lane] = s2;
  }
  __syncthreads();
  *out1 = (red[0][cl] + red[0][8 + cl]) + (red[0][16 + cl] + red[0][24 + cl]);
  *out2 = (red[1][cl] + red[1][8 + cl]) + (red[1][16 + cl] + red[1][24 + cl]);
}

static inline int finalize_cpb(int C) { return C >= 512 ? 8 : (C >= 256 ? 4 : (C >= 128 ? 2 : 1)); }

__global__ void __launch_bounds__(256) bn_finalize_kernel(const dfl_bn_finalize_args a, int cpb) {
  __shared__ double red[2][256];
  const int c = blockIdx.x * cpb + (threadIdx.x % cpb);
  double s1, s2;
  sum_partials_f64(a.partials, a.nblocks, a.C, cpb, &s1, &s2, red);
  if (threadIdx.x < cpb && c < a.C) {
    const double cnt = (double)a.count;
    const double mean = s1 / cnt;
    double var = s2 / cnt - mean * mean;  // biased variance used for normalisation
    if (var < 0.0) var = 0.0;
    const double invstd = 1.0 / sqrt(var + (double)a.eps);
    const float scale = (float)((double)a.gamma[c] * invstd);
    a.scale[c] = scale;
    a.shift[c] = (float)((double)a.beta[c] - mean * (double)a.gamma[c] * invstd);
    a.save_mean[c] = (float)mean;
    a.save_invstd[c] = (float)invstd;
    if (a.running_mean != nullptr) {
      const double mom = (double)a.momentum;
      const double unbiased = (cnt > 1.0) ? var * cnt / (cnt - 1.0) : var;
      a.running_mean[c] = (float)((1.0 - mom) * (double)a.running_mean[c] + mom * mean);
      a.running_var[c] = (float)((1.0 - mom) * (double)a.running_var[c] + mom * unbiased);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.num_batches_tracked != nullptr) *a.num_batches_tracked += 1;
}

// All "live" BatchNorm layers of a forward pass in one launch (dfl_bn_finalize_live): blockIdx.y = layer, a thread = a channel.
__global__ void __launch_bounds__(256) bn_finalize_live_kernel(const dfl_bn_live_job* __restrict__ jobs) {
  const dfl_bn_live_job j = jobs[blockIdx.y];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c < j.C) {
    float scale, shift;
    double mean, var;
    bn_live_affine(j.totals, j.gamma, j.beta, (double)j.count, j.eps, j.C, c, &scale, &shift, &mean, &var);
    j.scale[c] = scale;
    j.shift[c] = shift;
    j.save_mean[c] = (float)mean;
    j.save_invstd[c] = (float)(1.0 / sqrt(var + (double)j.eps));
    if (j.running_mean != nullptr) {
      const double mom = (double)j.momentum, cnt = (double)j.count;
      const double unbiased = (cnt > 1.0) ? var * cnt / (cnt - 1.0) : var;
      j.running_mean[c] = (float)((1.0 - mom) * (double)j.running_mean[c] + mom * mean);
      j.running_var[c] = (float)((1.0 - mom) * (double)j.running_var[c] + mom * unbiased);
    }
  }
  if (blockIdx.x == 0 && threadIdx.x == 0 && j.num_batches_tracked != nullptr) *j.num_batches_tracked += 1;
}

__global__ void __launch_bounds__(256) bn_bwd_finalize_live_kernel(const dfl_bn_bwd_live_job* __restrict__ jobs) {
  const dfl_bn_bwd_live_job j = jobs[blockIdx.y];
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= j.C) return;
  double sdy = 0.0, sdyr = 0.0;
#pragma unroll
  for (int r = 0; r < DFL_BN_R; ++r) {
    sdy += j.totals[(int64_t)(r * 2 + 0) * j.C + c];
    sdyr += j.totals[(int64_t)(r * 2 + 1) * j.C + c];
  }
  const double mean = (double)j.save_mean[c], invstd = (double)j.save_invstd[c];
  j.dgamma[c] = (float)(invstd * (sdyr - mean * sdy));
  j.dbeta[c] = (float)sdy;
  if (j.sum_out != nullptr) j.sum_out[c] = (float)sdy;
}

__global__ void bn_eval_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ rm, const float* __restrict__ rv, float* __restrict__ scale,
                               float* __restrict__ shift, int C, float eps) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c < C) {
    const float invstd = 1.0f / sqrtf(rv[c] + eps);
    const float s = gamma[c] * invstd;
    scale[c] = s;
    shift[c] = beta[c] - rm[c] * s;
  }
}

// dy-side finalize: dgamma, dbeta and the affine form of BatchNorm+ReLU backward.
__global__ void __launch_bounds__(256) bn_bwd_finalize_kernel(const dfl_bn_bwd_finalize_args a, int cpb) {
  __shared__ double red[2][256];
  const int c = blockIdx.x * cpb + (threadIdx.x % cpb);
  double sdy, sdyr;
  sum_partials_f64(a.partials, a.nblocks, a.C, cpb, &sdy, &sdyr, red);
  if (threadIdx.x < cpb && c < a.C) {
    const double cnt = (double)a.count;
    const double mean = (double)a.save_mean[c], invstd = (double)a.save_invstd[c], g = (double)a.gamma[c];
    const double sdyx = invstd * (sdyr - mean * sdy);  // sum dy * xhat
    a.dgamma[c] = (float)sdyx;
    a.dbeta[c] = (float)sdy;
    const double s = g * invstd;
    // count == 0: the statistics were constants (eval mode, running mean / variance): no batch-mean terms
    const double c1 = a.count > 0 ? sdy / cnt : 0.0, c2 = a.count > 0 ? sdyx / cnt : 0.0;
    // dr = s*(dy - c1 - xhat*c2),  xhat = (r - mean)*invstd
    a.coef[0 * a.C + c] = (float)s;
    a.coef[1 * a.C + c] = (float)(-s * c2 * invstd);
    a.coef[2 * a.C + c] = (float)(-s * c1 + s * c2 * invstd * mean);
  }
}

// dpre = [r > 0] * (A*dy + B*r + C); partials[blk][c] = column sums of dpre.
template <int VEC>
__global__ void __launch_bounds__(256) bn_relu_bwd_kernel(const dfl_bn_relu_bwd_args a, int UX, int rows_per_block) {
  __shared__ float red[256][VEC ? 4 : 1];
  const int RY = 256 / UX;
  const int ux = threadIdx.x % UX, uy = threadIdx.x / UX;
  const int unit = blockIdx.y * UX + ux;
  const int c = VEC ? unit * 4 : unit;
  const int C = a.C;
  const bool cok = c < C;
  constexpr int W = VEC ? 4 : 1;
  float cA[W], cB[W], cC[W], s[W];
#pragma unroll
  for (int j = 0; j < W; ++j) {
    cA[j] = 1.f; cB[j] = 0.f; cC[j] = 0.f; s[j] = 0.f;
    if (a.coef != nullptr && cok && c + j < C) {
      cA[j] = a.coef[c + j];
      cB[j] = a.coef[C + c + j];
      cC[j] = a.coef[2 * C + c + j];
    }
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > a.M) r1 = a.M;
  if (cok) {
    for (int64_t r = r0 + uy; r < r1; r += RY) {
      if constexpr (VEC) {
        const float4 dy = *reinterpret_cast<const float4*>(a.dy + r * a.lddy + c);
        const float4 rv = *reinterpret_cast<const float4*>(a.r + r * a.ldr + c);
        float4 o;
        o.x = rv.x > 0.f ? fmaf(cA[0], dy.x, fmaf(cB[0], rv.x, cC[0])) : 0.f;
        o.y = rv.y > 0.f ? fmaf(cA[1], dy.y, fmaf(cB[1], rv.y, cC[1])) : 0.f;
        o.z = rv.z > 0.f ? fmaf(cA[2], dy.z, fmaf(cB[2], rv.z, cC[2])) : 0.f;
        o.w = rv.w > 0.f ? fmaf(cA[3], dy.w, fmaf(cB[3], rv.w, cC[3])) : 0.f;
        if (a.split_out) {   // hi4 | lo4 bf16 in the float4's slot (consumers: split-bf16 GEMMs only)
          uint2 parts[2];
          split_bf16<2>(o, parts);
          *reinterpret_cast<uint4*>(a.dpre + r * a.ldo + c) = make_uint4(parts[0].x, parts[0].y, parts[1].x, parts[1].y);
        } else {
          *reinterpret_cast<float4*>(a.dpre + r * a.ldo + c) = o;
        }
        s[0] += o.x; s[1] += o.y; s[2] += o.z; s[3] += o.w;
      } else {
        const float dy = a.dy[r * a.lddy + c], rv = a.r[r * a.ldr + c];
        const float o = rv > 0.f ? fmaf(cA[0], dy, fmaf(cB[0], rv, cC[0])) : 0.f;
        a.dpre[r * a.ldo + c] = o;
        s[0] += o;
      }
    }
  }
  if (a.partials == nullptr) return;
#pragma unroll
  for (int j = 0; j < W; ++j) red[threadIdx.x][j] = s[j];
  __syncthreads();
  if (uy == 0 && cok) {
#pragma unroll
    for (int j = 0; j < W; ++j) {
      float t = 0.f;
      for (int y = 0; y < RY; ++y) t += red[y * UX + ux][j];
      if (c + j < C) a.partials[(int64_t)blockIdx.x * C + c + j] = t;
    }
  }
}

// out[c] = sum_b partials[b*stride + c] in fp64
__global__ void __launch_bounds__(256) reduce_partials_kernel(const float* __restrict__ partials, float* __restrict__ out,
                                                             int nblocks, int stride, int C) {
  __shared__ double red[256];
  const int cl = threadIdx.x & 7, rl = threadIdx.x >> 3;
  const int c = blockIdx.x * 8 + cl;
  double s = 0.0;
  if (c < C)
    for (int r = rl; r < nblocks; r += 32) s += (double)partials[(int64_t)r * stride + c];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int off = 16; off >= 1; off >>= 1) {
    if (rl < off) red[threadIdx.x] += red[threadIdx.x + off * 8];
    __syncthreads();
  }
  if (threadIdx.x < 8 && c < C) out[c] = (float)red[threadIdx.x];
}

// Batched sums (dfl_reduce_batch): blockIdx.x -> (job, block of the job) by binary search over the job table.
// A thread owns 4 consecutive outputs and reads them as one float4 per slice (n % 4 == 0 and 16-byte aligned rows: every
// weight-gradient and bias job; otherwise four scalar loads), four slices in flight, fp64 accumulation in a fixed order.
// Many slices (count >= 32): a workgroup takes 32 outputs (one 128-byte line per slice) x 32 slice lanes, eight slices of a
// lane in flight, and adds the lanes up through LDS.  Few slices: 1024 outputs per workgroup, the slices walked in order.
#ifndef DFL_RB_O
#define DFL_RB_O 32
#endif
constexpr int RB_T = 256, RB_O = DFL_RB_O, RB_S = 4 * RB_T / RB_O, RB_WIDE_MIN = 32;
static inline int reduce_job_blocks(int64_t n, int count) {
  return (int)(count >= RB_WIDE_MIN ? ceil_div(n, RB_O) : ceil_div(n, 4 * RB_T));
}

__device__ __forceinline__ void rb_load4(const float* __restrict__ p, int64_t i, int64_t n, bool vec, double* a) {
  if (vec) {
    const float4 v = *reinterpret_cast<const float4*>(p + i);
    a[0] += (double)v.x; a[1] += (double)v.y; a[2] += (double)v.z; a[3] += (double)v.w;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (i + e < n) a[e] += (double)p[i + e];
  }
}

__device__ __forceinline__ void rb_store4(const dfl_reduce_job& j, int64_t i, const double* v) {
#pragma unroll
  for (int e = 0; e < 4; ++e)
    if (i + e < j.n) {
      const int64_t ii = i + e;
      j.dst[j.T > 1 ? (ii % (j.n / j.T)) * j.T + ii / (j.n / j.T) : ii] = (float)v[e];
    }
}

__global__ void __launch_bounds__(RB_T) reduce_batch_kernel(const dfl_reduce_job* __restrict__ jobs, int njobs) {
  __shared__ double red[RB_S][RB_O + 2];
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const dfl_reduce_job j = jobs[lo];
  const int b = (int)blockIdx.x - j.first_block;
  const bool vec = (j.n & 3) == 0 && (j.stride & 3) == 0 && (reinterpret_cast<uintptr_t>(j.src) & 15) == 0;
  if (j.count >= RB_WIDE_MIN) {
    const int o4 = threadIdx.x % (RB_O / 4), sl = threadIdx.x / (RB_O / 4);
    const int64_t i = (int64_t)b * RB_O + 4 * o4;
    double a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0}, a3[4] = {0, 0, 0, 0};
    if (i < j.n) {
      int k = sl;
      if (vec) {
        for (; k + 7 * RB_S < j.count; k += 8 * RB_S) {      // eight 16-byte loads in flight
          float4 v[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(j.src + (int64_t)(k + u * RB_S) * j.stride + i);
#pragma unroll
          for (int u = 0; u < 8; u += 4) {
            a0[0] += (double)v[u].x; a0[1] += (double)v[u].y; a0[2] += (double)v[u].z; a0[3] += (double)v[u].w;
            a1[0] += (double)v[u + 1].x; a1[1] += (double)v[u + 1].y; a1[2] += (double)v[u + 1].z; a1[3] += (double)v[u + 1].w;
            a2[0] += (double)v[u + 2].x; a2[1] += (double)v[u + 2].y; a2[2] += (double)v[u + 2].z; a2[3] += (double)v[u + 2].w;
            a3[0] += (double)v[u + 3].x; a3[1] += (double)v[u + 3].y; a3[2] += (double)v[u + 3].z; a3[3] += (double)v[u + 3].w;
          }
        }
      }
      for (; k + 3 * RB_S < j.count; k += 4 * RB_S) {
        rb_load4(j.src + (int64_t)k * j.stride, i, j.n, vec, a0);
        rb_load4(j.src + (int64_t)(k + RB_S) * j.stride, i, j.n, vec, a1);
        rb_load4(j.src + (int64_t)(k + 2 * RB_S) * j.stride, i, j.n, vec, a2);
        rb_load4(j.src + (int64_t)(k + 3 * RB_S) * j.stride, i, j.n, vec, a3);
      }
      for (; k < j.count; k += RB_S) rb_load4(j.src + (int64_t)k * j.stride, i, j.n, vec, a0);
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) red[sl][4 * o4 + e] = (a0[e] + a1[e]) + (a2[e] + a3[e]);
    __syncthreads();
    for (int off = RB_S / 2; off >= 1; off >>= 1) {
      if (sl < off) {
#pragma unroll
        for (int e = 0; e < 4; ++e) red[sl][4 * o4 + e] += red[sl + off][4 * o4 + e];
      }
      __syncthreads();
    }
    if (sl == 0 && i < j.n) rb_store4(j, i, &red[0][4 * o4]);
  } else {
    const int64_t i = ((int64_t)b * RB_T + threadIdx.x) * 4;
    if (i < j.n) {
      double s[4] = {0, 0, 0, 0};
      for (int k = 0; k < j.count; ++k) rb_load4(j.src + (int64_t)k * j.stride, i, j.n, vec, s);
      rb_store4(j, i, s);
    }
  }
}

// ------------------------------------------------------------------------------------------------ affine copy
template <int VEC>
__global__ void __launch_bounds__(256) affine_copy_kernel(const dfl_affine_copy_args a, int64_t total_units, int cq) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_units; i += stride) {
    const int u = (int)(i % cq);
    int64_t pix = i / cq;
    const int xw = (int)(pix % a.W);
    pix /= a.W;
    const int yh = (int)(pix % a.H);
    const int n = (int)(pix / a.H);
    const int c = VEC ? u * 4 : u;
    const float* src = a.x + (((int64_t)n * a.xH + a.xoy + yh) * a.xW + a.xox + xw) * a.ldx + c;
    float* dst = a.y + (((int64_t)n * a.yH + a.yoy + yh) * a.yW + a.yox + xw) * a.ldy + c;
    if constexpr (VEC) {
      float4 v = *reinterpret_cast<const float4*>(src);
      if (a.scale != nullptr) {
        const float4 sc = *reinterpret_cast<const float4*>(a.scale + c);
        const float4 sh = *reinterpret_cast<const float4*>(a.shift + c);
        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
        v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
      }
      if (a.accumulate) {
        const float4 o = *reinterpret_cast<const float4*>(dst);
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
      }
      *reinterpret_cast<float4*>(dst) = v;
    } else {
      float v = *src;
      if (a.scale != nullptr) v = fmaf(v, a.scale[c], a.shift[c]);
      if (a.accumulate) v += *dst;
      *dst = v;
    }
  }
}

// ------------------------------------------------------------------------------------------------ max pool
template <int VEC>
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const dfl_pool_args a, int64_t total_units, int cq) {
  const int Ho = a.H / 2, Wo = a.W / 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_units; i += stride) {
    const int u = (int)(i % cq);
    int64_t pix = i / cq;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    const int c = VEC ? u * 4 : u;
    const float* s00 = a.x + (((int64_t)n * a.H + 2 * oy) * a.W + 2 * ox) * a.ldx + c;
    const float* s10 = s00 + (int64_t)a.W * a.ldx;
    float* dst = a.y + (((int64_t)n * Ho + oy) * Wo + ox) * a.ldy + c;
    if constexpr (VEC) {
      const float4 v0 = *reinterpret_cast<const float4*>(s00), v1 = *reinterpret_cast<const float4*>(s00 + a.ldx);
      const float4 v2 = *reinterpret_cast<const float4*>(s10), v3 = *reinterpret_cast<const float4*>(s10 + a.ldx);
      float4 m;
      m.x = fmaxf(fmaxf(v0.x, v1.x), fmaxf(v2.x, v3.x));
      m.y = fmaxf(fmaxf(v0.y, v1.y), fmaxf(v2.y, v3.y));
      m.z = fmaxf(fmaxf(v0.z, v1.z), fmaxf(v2.z, v3.z));
      m.w = fmaxf(fmaxf(v0.w, v1.w), fmaxf(v2.w, v3.w));
      *reinterpret_cast<float4*>(dst) = m;
    } else {
      *dst = fmaxf(fmaxf(s00[0], s00[a.ldx]), fmaxf(s10[0], s10[a.ldx]));
    }
  }
}

__device__ __forceinline__ int first_max4(float v0, float v1, float v2, float v3) {
  int k = 0;
  float m = v0;
  if (v1 > m) { m = v1; k = 1; }
  if (v2 > m) { m = v2; k = 2; }
  if (v3 > m) { k = 3; }
  return k;
}

template <int VEC>
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const dfl_pool_args a, int64_t total_units, int cq) {
  const int Ho = a.H / 2, Wo = a.W / 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_units; i += stride) {
    const int u = (int)(i % cq);
    int64_t pix = i / cq;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    const int c = VEC ? u * 4 : u;
    const int64_t ipix = ((int64_t)n * a.H + 2 * oy) * a.W + 2 * ox;
    const float* s00 = a.x + ipix * a.ldx + c;
    const float* s10 = s00 + (int64_t)a.W * a.ldx;
    float* d00 = a.dx + ipix * a.lddx + c;
    float* d10 = d00 + (int64_t)a.W * a.lddx;
    const float* g = a.y + (((int64_t)n * Ho + oy) * Wo + ox) * a.ldy + c;
    constexpr int W = VEC ? 4 : 1;
#pragma unroll
    for (int j = 0; j < W; ++j) {
      const int k = first_max4(s00[j], s00[a.ldx + j], s10[j], s10[a.ldx + j]);
      float* d = (k == 0) ? d00 + j : (k == 1) ? d00 + a.lddx + j : (k == 2) ? d10 + j : d10 + a.lddx + j;
      *d += g[j];
    }
  }
}

// ------------------------------------------------------------------------------------------------ bf16 tensors (math mode 4)
// The same streaming kernels for bf16 activations: units of 8 channels (16 bytes), fp32 arithmetic, one rounding at the
// store.  Sums are taken from the values as stored.
typedef unsigned int bu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void unpack8(const bu32x4 w, float* f) {
  f[0] = __uint_as_float(w.x << 16); f[1] = __uint_as_float(w.x & 0xffff0000u);
  f[2] = __uint_as_float(w.y << 16); f[3] = __uint_as_float(w.y & 0xffff0000u);
  f[4] = __uint_as_float(w.z << 16); f[5] = __uint_as_float(w.z & 0xffff0000u);
  f[6] = __uint_as_float(w.w << 16); f[7] = __uint_as_float(w.w & 0xffff0000u);
}
__device__ __forceinline__ bu32x4 pack8(const float* f) {
  bu32x4 w;
  w.x = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){f[0], f[1]}, bf16x2_t));
  w.y = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){f[2], f[3]}, bf16x2_t));
  w.z = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){f[4], f[5]}, bf16x2_t));
  w.w = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){f[6], f[7]}, bf16x2_t));
  return w;
}
__device__ __forceinline__ bu32x4 ld8(const float* base, int64_t elem) {
  return *reinterpret_cast<const bu32x4*>(reinterpret_cast<const unsigned short*>(base) + elem);
}
__device__ __forceinline__ void st8(float* base, int64_t elem, bu32x4 w) {
  *reinterpret_cast<bu32x4*>(reinterpret_cast<unsigned short*>(base) + elem) = w;
}

static RowGeom row_geom8(int C) {
  RowGeom g;
  g.vec = 1;
  g.units = C / 8;
  int ux = 1;
  while (ux * 2 <= g.units && ux * 2 <= 256) ux *= 2;
  g.UX = ux;
  g.RY = 256 / ux;
  g.gy = (int)ceil_div(g.units, ux);
  return g;
}

// per-unit partial sums of a workgroup -> partials row (red: [256][8])
__device__ __forceinline__ void reduce_units8(float (*red)[8], const float* s, int ux, int uy, int UX, int RY, bool cok,
                                              float* out, int c) {
#pragma unroll
  for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = s[j];
  __syncthreads();
  if (uy == 0 && cok) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float t = 0.f;
      for (int y = 0; y < RY; ++y) t += red[y * UX + ux][j];
      out[c + j] = t;
    }
  }
  __syncthreads();
}

__global__ void __launch_bounds__(256) colstats_bf16_kernel(const dfl_colstats_args a, int UX, int rows_per_block) {
  __shared__ float red[256][8];
  const int RY = 256 / UX;
  const int ux = threadIdx.x % UX, uy = threadIdx.x / UX;
  const int c = (blockIdx.y * UX + ux) * 8;
  const bool cok = c < a.C;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > a.M) r1 = a.M;
  float s1[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, s2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (cok) {
    for (int64_t r = r0 + uy; r < r1; r += RY) {
      float va[8], vb[8];
      unpack8(ld8(a.a, r * a.lda + c), va);
      if (a.b != nullptr) unpack8(ld8(a.b, r * a.ldb + c), vb);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        s1[j] += va[j];
        s2[j] = fmaf(va[j], a.b != nullptr ? vb[j] : va[j], s2[j]);
      }
    }
  }
  reduce_units8(red, s1, ux, uy, UX, RY, cok, a.partials + ((int64_t)blockIdx.x * 2 + 0) * a.C, c);
  reduce_units8(red, s2, ux, uy, UX, RY, cok, a.partials + ((int64_t)blockIdx.x * 2 + 1) * a.C, c);
}

__global__ void __launch_bounds__(256) bn_relu_bwd_bf16_kernel(const dfl_bn_relu_bwd_args a, int UX, int rows_per_block) {
  __shared__ float red[256][8];
  const int RY = 256 / UX;
  const int ux = threadIdx.x % UX, uy = threadIdx.x / UX;
  const int C = a.C;
  const int c = (blockIdx.y * UX + ux) * 8;
  const bool cok = c < C;
  float cA[8], cB[8], cC[8], s[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    cA[j] = 1.f; cB[j] = 0.f; cC[j] = 0.f; s[j] = 0.f;
    if (a.coef != nullptr && cok) {
      cA[j] = a.coef[c + j];
      cB[j] = a.coef[C + c + j];
      cC[j] = a.coef[2 * C + c + j];
    }
  }
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  int64_t r1 = r0 + rows_per_block;
  if (r1 > a.M) r1 = a.M;
  if (cok) {
    auto one = [&](const bu32x4 wdy, const bu32x4 wr, int64_t r) {
      float dy[8], rv[8], o[8];
      unpack8(wdy, dy);
      unpack8(wr, rv);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = rv[j] > 0.f ? fmaf(cA[j], dy[j], fmaf(cB[j], rv[j], cC[j])) : 0.f;
      const bu32x4 w = pack8(o);
      st8(a.dpre, r * a.ldo + c, w);
      unpack8(w, o);
#pragma unroll
      for (int j = 0; j < 8; ++j) s[j] += o[j];
    };
    int64_t r = r0 + uy;
    for (; r + 3 * RY < r1; r += 4 * RY) {     // four rows in flight per thread
      const bu32x4 d0 = ld8(a.dy, r * a.lddy + c), d1 = ld8(a.dy, (r + RY) * a.lddy + c);
      const bu32x4 d2 = ld8(a.dy, (r + 2 * RY) * a.lddy + c), d3 = ld8(a.dy, (r + 3 * RY) * a.lddy + c);
      const bu32x4 q0 = ld8(a.r, r * a.ldr + c), q1 = ld8(a.r, (r + RY) * a.ldr + c);
      const bu32x4 q2 = ld8(a.r, (r + 2 * RY) * a.ldr + c), q3 = ld8(a.r, (r + 3 * RY) * a.ldr + c);
      one(d0, q0, r);
      one(d1, q1, r + RY);
      one(d2, q2, r + 2 * RY);
      one(d3, q3, r + 3 * RY);
    }
    for (; r < r1; r += RY) one(ld8(a.dy, r * a.lddy + c), ld8(a.r, r * a.ldr + c), r);
  }
  if (a.partials == nullptr) return;
  reduce_units8(red, s, ux, uy, UX, RY, cok, a.partials + (int64_t)blockIdx.x * C, c);
}

__global__ void __launch_bounds__(256) affine_copy_bf16_kernel(const dfl_affine_copy_args a, int64_t total_units, int cq) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_units; i += stride) {
    const int u = (int)(i % cq);
    int64_t pix = i / cq;
    const int xw = (int)(pix % a.W);
    pix /= a.W;
    const int yh = (int)(pix % a.H);
    const int n = (int)(pix / a.H);
    const int c = u * 8;
    const int64_t so = (((int64_t)n * a.xH + a.xoy + yh) * a.xW + a.xox + xw) * a.ldx + c;
    const int64_t dof = (((int64_t)n * a.yH + a.yoy + yh) * a.yW + a.yox + xw) * a.ldy + c;
    float v[8];
    unpack8(ld8(a.x, so), v);
    if (a.scale != nullptr) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], a.scale[c + j], a.shift[c + j]);
    }
    if (a.accumulate) {
      float o[8];
      unpack8(ld8(a.y, dof), o);
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] += o[j];
    }
    st8(a.y, dof, pack8(v));
  }
}

__global__ void __launch_bounds__(256) maxpool_fwd_bf16_kernel(const dfl_pool_args a, int64_t total_units, int cq) {
  const int Ho = a.H / 2, Wo = a.W / 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_units; i += stride) {
    const int u = (int)(i % cq);
    int64_t pix = i / cq;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    const int c = u * 8;
    const int64_t s00 = (((int64_t)n * a.H + 2 * oy) * a.W + 2 * ox) * a.ldx + c;
    const int64_t s10 = s00 + (int64_t)a.W * a.ldx;
    float v0[8], v1[8], v2[8], v3[8], m[8];
    unpack8(ld8(a.x, s00), v0);
    unpack8(ld8(a.x, s00 + a.ldx), v1);
    unpack8(ld8(a.x, s10), v2);
    unpack8(ld8(a.x, s10 + a.ldx), v3);
#pragma unroll
    for (int j = 0; j < 8; ++j) m[j] = fmaxf(fmaxf(v0[j], v1[j]), fmaxf(v2[j], v3[j]));
    st8(a.y, (((int64_t)n * Ho + oy) * Wo + ox) * a.ldy + c, pack8(m));
  }
}

__global__ void __launch_bounds__(256) maxpool_bwd_bf16_kernel(const dfl_pool_args a, int64_t total_units, int cq) {
  const int Ho = a.H / 2, Wo = a.W / 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_units; i += stride) {
    const int u = (int)(i % cq);
    int64_t pix = i / cq;
    const int ox = (int)(pix % Wo);
    pix /= Wo;
    const int oy = (int)(pix % Ho);
    const int n = (int)(pix / Ho);
    const int c = u * 8;
    const int64_t ipix = ((int64_t)n * a.H + 2 * oy) * a.W + 2 * ox;
    const int64_t s00 = ipix * a.ldx + c, s10 = s00 + (int64_t)a.W * a.ldx;
    const int64_t d00 = ipix * a.lddx + c, d10 = d00 + (int64_t)a.W * a.lddx;
    float v0[8], v1[8], v2[8], v3[8], g[8], e0[8], e1[8], e2[8], e3[8];
    unpack8(ld8(a.x, s00), v0);
    unpack8(ld8(a.x, s00 + a.ldx), v1);
    unpack8(ld8(a.x, s10), v2);
    unpack8(ld8(a.x, s10 + a.ldx), v3);
    unpack8(ld8(a.y, (((int64_t)n * Ho + oy) * Wo + ox) * a.ldy + c), g);
    unpack8(ld8(a.dx, d00), e0);
    unpack8(ld8(a.dx, d00 + a.lddx), e1);
    unpack8(ld8(a.dx, d10), e2);
    unpack8(ld8(a.dx, d10 + a.lddx), e3);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = first_max4(v0[j], v1[j], v2[j], v3[j]);
      e0[j] += k == 0 ? g[j] : 0.f;
      e1[j] += k == 1 ? g[j] : 0.f;
      e2[j] += k == 2 ? g[j] : 0.f;
      e3[j] += k == 3 ? g[j] : 0.f;
    }
    st8(a.dx, d00, pack8(e0));
    st8(a.dx, d00 + a.lddx, pack8(e1));
    st8(a.dx, d10, pack8(e2));
    st8(a.dx, d10 + a.lddx, pack8(e3));
  }
}

// ------------------------------------------------------------------------------------------------ weight pack
// GEMM operand layout of dfl_conv2d: w[kq][n][r] = W(k = 4*kq + r, n), zero for k >= K ("quad-packed": one float4 is four
// consecutive k of one output column).  The source is always a contiguous [A][B][C] parameter (C = KH*KW) and the
// (k, n) <-> (a, b, c) mapping is one of three kinds (include/dfl_hip.h).  Fast path: LDS-tiled -- a 32(A) x 32(B) x C
// tile is read as 32 runs of 32*C contiguous floats and written as 512-byte runs; element-wise fallback otherwise.
constexpr int PK_T = 32, PK_CMAX = 9;

__device__ __forceinline__ float pack_src(const dfl_pack_job& j, int k, int n) {
  int a, b, c;
  if (j.kind == 1) {
    c = k / j.B; b = k - c * j.B; a = n;
  } else if (j.kind == 2) {
    const int cp = k / j.A;
    a = k - cp * j.A; b = n; c = j.flip ? (j.C - 1 - cp) : cp;
  } else {
    a = k; c = n / j.B; b = n - c * j.B;
  }
  return j.src[((int64_t)a * j.B + b) * j.C + c];
}

// element r of quad slot q: a float, or (split format) hi bf16 at half-word r and lo bf16 at half-word 4 + r of the slot
// (the bf16 chunk layout, split = 2, has its own path in pack_kernel)
__device__ __forceinline__ void pack_put(float* dst, int64_t q, int r, float v, int split) {
  if (!split) {
    dst[q * 4 + r] = v;
  } else {
    const __bf16 h = (__bf16)v;
    const __bf16 l = (__bf16)(v - (float)h);
    unsigned short* d16 = reinterpret_cast<unsigned short*>(dst) + q * 8;
    d16[r] = __builtin_bit_cast(unsigned short, h);
    d16[4 + r] = __builtin_bit_cast(unsigned short, l);
  }
}

__global__ void __launch_bounds__(256) pack_kernel(const dfl_pack_job* __restrict__ jobs) {
  __shared__ float tile[PK_T][PK_T * PK_CMAX + 1];
  const dfl_pack_job j = jobs[blockIdx.y];
  const int A = j.A, B = j.B, Cc = j.C;
  const int K = (j.kind == 1) ? Cc * B : (j.kind == 2 ? Cc * A : A);
  const int N = (j.kind == 1) ? A : (j.kind == 2 ? B : Cc * B);
  if (j.split == 2 && Cc <= PK_CMAX && ((j.kind == 1) ? (B % 16 == 0) : (A % 16 == 0))) {
    // bf16 chunk layout [K/16][N][16], LDS-tiled: a 32(A) x 32(B) x C tile is read as 32 runs of 32*C contiguous floats;
    // a thread then builds whole cells (16 consecutive k of one column = 32 bytes) and consecutive threads take consecutive
    // columns: coalesced on both sides (the element-wise form below gathers 4-byte words 4*B*C bytes apart)
    const int tb = (B + PK_T - 1) / PK_T, ta = (A + PK_T - 1) / PK_T;
    const int run = PK_T * Cc;
    unsigned short* dst16 = reinterpret_cast<unsigned short*>(j.dst);
    for (int tidx = blockIdx.x; tidx < ta * tb; tidx += gridDim.x) {
      const int a0 = (tidx / tb) * PK_T, b0 = (tidx % tb) * PK_T;
      const int nb = min(PK_T, B - b0), na = min(PK_T, A - a0);
      __syncthreads();
      for (int e = threadIdx.x; e < PK_T * run; e += 256) {
        const int ar = e / run, q = e - ar * run;
        if (ar < na && q < nb * Cc) tile[ar][q] = j.src[((int64_t)(a0 + ar) * B + b0) * Cc + q];
      }
      __syncthreads();
      for (int e = threadIdx.x; e < 64 * Cc; e += 256) {       // cells of this tile: 32 columns x 2 blocks of 16 k x C taps
        const int x = e & 31, blk = (e >> 5) & 1, cp = e >> 6;
        float f[16];
        int64_t cell;
        bool ok;
        if (j.kind == 1) {          // k = c*B + b (16 consecutive b), n = a
          ok = x < na && 16 * blk < nb;
#pragma unroll
          for (int r = 0; r < 16; ++r) f[r] = tile[x][(16 * blk + r) * Cc + cp];
          cell = (int64_t)((cp * B + b0) / 16 + blk) * N + a0 + x;
        } else if (j.kind == 2) {   // k = c'*A + a (16 consecutive a), n = b
          const int c = j.flip ? (Cc - 1 - cp) : cp;
          ok = x < nb && 16 * blk < na;
#pragma unroll
          for (int r = 0; r < 16; ++r) f[r] = tile[16 * blk + r][x * Cc + c];
          cell = (int64_t)((cp * A + a0) / 16 + blk) * N + b0 + x;
        } else {                    // k = a (16 consecutive a), n = c*B + b
          ok = x < nb && 16 * blk < na;
#pragma unroll
          for (int r = 0; r < 16; ++r) f[r] = tile[16 * blk + r][x * Cc + cp];
          cell = (int64_t)(a0 / 16 + blk) * N + cp * B + b0 + x;
        }
        if (ok) {
          bu32x4* d = reinterpret_cast<bu32x4*>(dst16 + cell * 16);
          d[0] = pack8(f);
          d[1] = pack8(f + 8);
        }
      }
    }
    return;
  }
  if (j.split == 2) {
    // bf16 chunk layout [ceil(K/16)][N][16]: thread = one (chunk, column) cell of 32 bytes; consecutive threads take
    // consecutive columns (coalesced 32-byte stores; the sources are small enough to live in L2)
    const int Kc = (K + 15) / 16;
    const int64_t cells = (int64_t)Kc * N;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < cells; i += stride) {
      const int n = (int)(i % N), kc = (int)(i / N);
      float f[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = kc * 16 + r;
        f[r] = k < K ? pack_src(j, k, n) : 0.f;
      }
      bu32x4* d = reinterpret_cast<bu32x4*>(reinterpret_cast<unsigned short*>(j.dst) + i * 16);
      d[0] = pack8(f);
      d[1] = pack8(f + 8);
    }
    return;
  }
  const bool tiled = Cc <= PK_CMAX && ((j.kind == 1) ? (B % 4 == 0) : (A % 4 == 0));
  if (tiled) {
    const int tb = (B + PK_T - 1) / PK_T, ta = (A + PK_T - 1) / PK_T;
    const int run = PK_T * Cc;
    for (int tidx = blockIdx.x; tidx < ta * tb; tidx += gridDim.x) {
      const int a0 = (tidx / tb) * PK_T, b0 = (tidx % tb) * PK_T;
      const int nb = min(PK_T, B - b0), na = min(PK_T, A - a0);
      __syncthreads();
      for (int e = threadIdx.x; e < PK_T * run; e += 256) {   // 32 rows of 32*C contiguous source floats
        const int ar = e / run, q = e - ar * run;
        if (ar < na && q < nb * Cc) tile[ar][q] = j.src[((int64_t)(a0 + ar) * B + b0) * Cc + q];
      }
      __syncthreads();
      for (int e = threadIdx.x; e < PK_T * run; e += 256) {
        const int r = e & 3, x = (e >> 2) & 31, rest = e >> 7;   // rest in [0, 8*C)
        if (j.kind == 1) {          // k = c*B + b, n = a: quads run along b, columns along a
          const int c = rest >> 3, bq = rest & 7, bb = 4 * bq + r, ar = x;
          if (ar < na && bb < nb)
            pack_put(j.dst, (int64_t)((c * B + b0) / 4 + bq) * N + a0 + ar, r, tile[ar][bb * Cc + c], j.split);
        } else if (j.kind == 2) {   // k = c'*A + a, n = b: quads along a, columns along b
          const int cp = rest >> 3, aq = rest & 7, ar = 4 * aq + r, bb = x;
          const int c = j.flip ? (Cc - 1 - cp) : cp;
          if (ar < na && bb < nb)
            pack_put(j.dst, (int64_t)((cp * A + a0) / 4 + aq) * N + b0 + bb, r, tile[ar][bb * Cc + c], j.split);
        } else {                    // k = a, n = c*B + b
          const int c = rest >> 3, aq = rest & 7, ar = 4 * aq + r, bb = x;
          if (ar < na && bb < nb)
            pack_put(j.dst, (int64_t)(a0 / 4 + aq) * N + c * B + b0 + bb, r, tile[ar][bb * Cc + c], j.split);
        }
      }
    }
    return;
  }
  const int Kq = (K + 3) / 4;
  const int64_t total = (int64_t)Kq * N * 4;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int r = (int)(i & 3);
    const int64_t t = i >> 2;
    const int n = (int)(t % N), kq = (int)(t / N);
    const int k = 4 * kq + r;
    pack_put(j.dst, i >> 2, r, (k < K) ? pack_src(j, k, n) : 0.f, j.split);
  }
}

// ---- tiled dual-layout pack (dfl_pack_weights_tiled; round 4) ------------------------------------------------------------
// One workgroup per 32(A) x 32(B) x C tile of ALL jobs of the launch (flat list: no idle blocks for the small layers, which a
// (1024, jobs) grid spends 9 of 10 blocks on), the tile read with 16-byte loads, and BOTH bf16 chunk layouts of the parameter --
// the forward operand and the data-gradient operand -- written from the same LDS tile: the fp32 master is read once.
__device__ __forceinline__ void pack_emit_cells(const float (*tile)[PK_T * PK_CMAX + 1], unsigned short* dst16, int kind, int flip,
                                                int A, int B, int Cc, int a0, int b0) {
  const int N = (kind == 1) ? A : (kind == 2 ? B : Cc * B);
  for (int e = threadIdx.x; e < 64 * Cc; e += 256) {           // cells of this tile: 32 columns x 2 blocks of 16 k x C taps
    const int x = e & 31, blk = (e >> 5) & 1, cp = e >> 6;
    float f[16];
    int64_t cell;
    if (kind == 1) {                // k = c*B + b (16 consecutive b), n = a
#pragma unroll
      for (int r = 0; r < 16; ++r) f[r] = tile[x][(16 * blk + r) * Cc + cp];
      cell = (int64_t)((cp * B + b0) / 16 + blk) * N + a0 + x;
    } else if (kind == 2) {         // k = c'*A + a (16 consecutive a), n = b
      const int c = flip ? (Cc - 1 - cp) : cp;
#pragma unroll
      for (int r = 0; r < 16; ++r) f[r] = tile[16 * blk + r][x * Cc + c];
      cell = (int64_t)((cp * A + a0) / 16 + blk) * N + b0 + x;
    } else {                        // k = a (16 consecutive a), n = c*B + b
#pragma unroll
      for (int r = 0; r < 16; ++r) f[r] = tile[16 * blk + r][x * Cc + cp];
      cell = (int64_t)(a0 / 16 + blk) * N + cp * B + b0 + x;
    }
    bu32x4* d = reinterpret_cast<bu32x4*>(dst16 + cell * 16);
    d[0] = pack8(f);
    d[1] = pack8(f + 8);
  }
}

// ... and the quad layouts of the fp32-tensor arithmetics (4 floats, or 4 hi | 4 lo bf16: pack_put) from the same tile -- the
// element order of pack_kernel's tiled branch for a full 32 x 32 tile (round 5: the parity modes take the one-pass update too)
__device__ __forceinline__ void pack_emit_quads(const float (*tile)[PK_T * PK_CMAX + 1], float* dst, int kind, int flip, int split,
                                                int A, int B, int Cc, int a0, int b0) {
  const int N = (kind == 1) ? A : (kind == 2 ? B : Cc * B);
  const int run = PK_T * Cc;
  for (int e = threadIdx.x; e < PK_T * run; e += 256) {
    const int r = e & 3, x = (e >> 2) & 31, rest = e >> 7;   // rest in [0, 8*C)
    if (kind == 1) {            // k = c*B + b, n = a: quads run along b, columns along a
      const int c = rest >> 3, bq = rest & 7, bb = 4 * bq + r, ar = x;
      pack_put(dst, (int64_t)((c * B + b0) / 4 + bq) * N + a0 + ar, r, tile[ar][bb * Cc + c], split);
    } else if (kind == 2) {     // k = c'*A + a, n = b: quads along a, columns along b
      const int cp = rest >> 3, aq = rest & 7, ar = 4 * aq + r, bb = x;
      const int c = flip ? (Cc - 1 - cp) : cp;
      pack_put(dst, (int64_t)((cp * A + a0) / 4 + aq) * N + b0 + bb, r, tile[ar][bb * Cc + c], split);
    } else {                    // k = a, n = c*B + b
      const int c = rest >> 3, aq = rest & 7, ar = 4 * aq + r, bb = x;
      pack_put(dst, (int64_t)(a0 / 4 + aq) * N + c * B + b0 + bb, r, tile[ar][bb * Cc + c], split);
    }
  }
}
__device__ __forceinline__ void pack_emit(const float (*tile)[PK_T * PK_CMAX + 1], float* dst, int kind, int flip, int split, int A, int B,
                                          int Cc, int a0, int b0) {
  if (split == 2) pack_emit_cells(tile, reinterpret_cast<unsigned short*>(dst), kind, flip, A, B, Cc, a0, b0);
  else pack_emit_quads(tile, dst, kind, flip, split, A, B, Cc, a0, b0);
}

__global__ void __launch_bounds__(256) pack_tiles_kernel(const dfl_pack_job* __restrict__ jobs, int njobs) {
  __shared__ float tile[PK_T][PK_T * PK_CMAX + 1];
  // which job: the last one whose first_tile <= blockIdx.x (binary search; the few records stay in the scalar cache)
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_tile <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const dfl_pack_job j = jobs[lo];
  const int A = j.A, B = j.B, Cc = j.C;
  const int tb = B / PK_T;
  const int tidx = (int)blockIdx.x - j.first_tile;
  const int a0 = (tidx / tb) * PK_T, b0 = (tidx % tb) * PK_T;
  const int run4 = 8 * Cc;                                     // float4 per tile row (32 * C floats)
  for (int e = threadIdx.x; e < PK_T * run4; e += 256) {
    const int ar = e / run4, q4 = e - ar * run4;
    const float4 v = *reinterpret_cast<const float4*>(j.src + ((int64_t)(a0 + ar) * B + b0) * Cc + 4 * q4);
    float* t = &tile[ar][4 * q4];
    t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
  }
  __syncthreads();
  pack_emit(tile, j.dst, j.kind, j.flip, j.split, A, B, Cc, a0, b0);
  if (j.dst2 != nullptr) pack_emit(tile, j.dst2, j.kind2, j.flip2, j.split2, A, B, Cc, a0, b0);
}

// ------------------------------------------------------------------------------------------------ SGD
__global__ void __launch_bounds__(256) sgd_kernel(float* __restrict__ p, const float* __restrict__ grad,
                                                 float* __restrict__ buf, int64_t n, float lr, float mom, float wd,
                                                 float gscale, int nesterov, int first) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float w = p[i];
    float g = fmaf(wd, w, grad[i] * gscale);
    if (mom != 0.f) {
      const float b = first ? g : fmaf(mom, buf[i], g);
      buf[i] = b;
      g = nesterov ? fmaf(mom, b, g) : b;
    }
    p[i] = fmaf(-lr, g, w);
  }
}

__device__ __forceinline__ float sgd_update(float w, float gr, float* b, float lr, float mom, float wd, float gscale, int nesterov) {
  float g = fmaf(wd, w, gr * gscale);                          // (the arithmetic of sgd_kernel, word for word)
  if (mom != 0.f) {
    const float nb = fmaf(mom, *b, g);
    *b = nb;
    g = nesterov ? fmaf(mom, nb, g) : nb;
  }
  return fmaf(-lr, g, w);
}

// dfl_sgd_pack_tiled: pack_tiles_kernel whose workgroups update their tile of the master before they emit its layouts -- the
// weights are read once per step (sgd_kernel + pack_tiles_kernel: twice) and one launch goes.
__global__ void __launch_bounds__(256) sgd_pack_tiles_kernel(dfl_sgd_pack_args a) {
  __shared__ float tile[PK_T][PK_T * PK_CMAX + 1];
  const dfl_pack_job* __restrict__ jobs = a.jobs_dev;
  int lo = 0, hi = a.njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_tile <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const dfl_pack_job j = jobs[lo];
  const int tidx = (int)blockIdx.x - j.first_tile;
  float* __restrict__ P = const_cast<float*>(j.src);
  const float* __restrict__ G = j.src + a.grad_delta;
  float* __restrict__ Bf = const_cast<float*>(j.src) + a.buf_delta;
  const bool hasb = a.momentum != 0.f;
  if (j.kind == DFL_PACK_PLAIN) {
    const int64_t i0 = (int64_t)tidx * DFL_SGD_PLAIN_TILE;
    const int64_t i1 = min((int64_t)j.A, i0 + DFL_SGD_PLAIN_TILE);
    for (int64_t i = i0 + threadIdx.x; i < i1; i += 256) {
      float b = hasb ? Bf[i] : 0.f;
      P[i] = sgd_update(P[i], G[i], &b, a.lr, a.momentum, a.weight_decay, a.grad_scale, a.nesterov);
      if (hasb) Bf[i] = b;
    }
    return;
  }
  const int A = j.A, B = j.B, Cc = j.C;
  const int tb = B / PK_T;
  const int a0 = (tidx / tb) * PK_T, b0 = (tidx % tb) * PK_T;
  const int run4 = 8 * Cc;
  for (int e = threadIdx.x; e < PK_T * run4; e += 256) {
    const int ar = e / run4, q4 = e - ar * run4;
    const int64_t o = ((int64_t)(a0 + ar) * B + b0) * Cc + 4 * q4;
    const float4 w = *reinterpret_cast<const float4*>(P + o);
    const float4 g = *reinterpret_cast<const float4*>(G + o);
    float4 b = hasb ? *reinterpret_cast<const float4*>(Bf + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 v;
    v.x = sgd_update(w.x, g.x, &b.x, a.lr, a.momentum, a.weight_decay, a.grad_scale, a.nesterov);
    v.y = sgd_update(w.y, g.y, &b.y, a.lr, a.momentum, a.weight_decay, a.grad_scale, a.nesterov);
    v.z = sgd_update(w.z, g.z, &b.z, a.lr, a.momentum, a.weight_decay, a.grad_scale, a.nesterov);
    v.w = sgd_update(w.w, g.w, &b.w, a.lr, a.momentum, a.weight_decay, a.grad_scale, a.nesterov);
    *reinterpret_cast<float4*>(P + o) = v;
    if (hasb) *reinterpret_cast<float4*>(Bf + o) = b;
    float* t = &tile[ar][4 * q4];
    t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
  }
  __syncthreads();
  pack_emit(tile, j.dst, j.kind, j.flip, j.split, A, B, Cc, a0, b0);
  if (j.dst2 != nullptr) pack_emit(tile, j.dst2, j.kind2, j.flip2, j.split2, A, B, Cc, a0, b0);
}

static unsigned stream_grid(int64_t units) {
  int64_t b = ceil_div(units, 256);
  if (b > 8192) b = 8192;
  if (b < 1) b = 1;
  return (unsigned)b;
}

}  // namespace dfl

using namespace dfl;

extern "C" int dfl_rowblock_count(int64_t M, int32_t C) { return rowblocks(M, C); }

extern "C" int dfl_colstats(const dfl_colstats_args* a, dfl_stream_t stream) {
  DFL_REQUIRE(a && a->a && a->partials && a->M > 0 && a->C > 0, "dfl_colstats: bad args");
  DFL_REQUIRE(a->nblocks == rowblocks(a->M, a->C), "dfl_colstats: nblocks must be dfl_rowblock_count(M, C)");
  if (a->bf16) {
    DFL_REQUIRE(a->C % 8 == 0 && a->lda % 8 == 0 && aligned16(a->a) && (a->b == nullptr || (a->ldb % 8 == 0 && aligned16(a->b))),
                "dfl_colstats (bf16): C and ld must be multiples of 8, tensors 16-byte aligned");
    const RowGeom g8 = row_geom8(a->C);
    hipLaunchKernelGGL(colstats_bf16_kernel, dim3((unsigned)a->nblocks, (unsigned)g8.gy), dim3(256), 0, static_cast<hipStream_t>(stream),
                       *a, g8.UX, (int)ceil_div(a->M, a->nblocks));
    return check_launch("dfl_colstats");
  }
  const bool vec_ok = a->lda % 4 == 0 && aligned16(a->a) && (a->b == nullptr || (a->ldb % 4 == 0 && aligned16(a->b)));
  const RowGeom g = row_geom(a->C, vec_ok);
  const int rpb = (int)ceil_div(a->M, a->nblocks);
  dim3 grid((unsigned)a->nblocks, (unsigned)g.gy);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (g.vec)
    hipLaunchKernelGGL(colstats_kernel<1>, grid, dim3(256), 0, s, a->a, a->b, a->partials, a->M, a->C, a->lda, a->ldb, g.UX, rpb);
  else
    hipLaunchKernelGGL(colstats_kernel<0>, grid, dim3(256), 0, s, a->a, a->b, a->partials, a->M, a->C, a->lda, a->ldb, g.UX, rpb);
  return check_launch("dfl_colstats");
}

extern "C" int dfl_bn_finalize(const dfl_bn_finalize_args* a, dfl_stream_t stream) {
  DFL_REQUIRE(a && a->partials && a->gamma && a->beta && a->scale && a->shift && a->save_mean && a->save_invstd,
              "dfl_bn_finalize: missing pointer");
  DFL_REQUIRE(a->C > 0 && a->nblocks > 0 && a->count > 0, "dfl_bn_finalize: bad sizes");
  DFL_REQUIRE((a->running_mean == nullptr) == (a->running_var == nullptr), "dfl_bn_finalize: running stats go together");
  const int cpb = finalize_cpb(a->C);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)ceil_div(a->C, cpb)), dim3(256), 0, static_cast<hipStream_t>(stream), *a,
                     cpb);
  return check_launch("dfl_bn_finalize");
}

extern "C" int dfl_bn_finalize_live(const dfl_bn_live_job* jobs_dev, int32_t njobs, int32_t max_C, dfl_stream_t stream) {
  DFL_REQUIRE(jobs_dev && njobs > 0 && max_C > 0, "dfl_bn_finalize_live: bad args");
  hipLaunchKernelGGL(bn_finalize_live_kernel, dim3((unsigned)ceil_div(max_C, 256), (unsigned)njobs), dim3(256), 0,
                     static_cast<hipStream_t>(stream), jobs_dev);
  return check_launch("dfl_bn_finalize_live");
}

extern "C" int dfl_bn_bwd_finalize_live(const dfl_bn_bwd_live_job* jobs_dev, int32_t njobs, int32_t max_C, dfl_stream_t stream) {
  DFL_REQUIRE(jobs_dev && njobs > 0 && max_C > 0, "dfl_bn_bwd_finalize_live: bad args");
  hipLaunchKernelGGL(bn_bwd_finalize_live_kernel, dim3((unsigned)ceil_div(max_C, 256), (unsigned)njobs), dim3(256), 0,
                     static_cast<hipStream_t>(stream), jobs_dev);
  return check_launch("dfl_bn_bwd_finalize_live");
}

extern "C" int dfl_bn_eval_prepare(const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, float* scale, float* shift, int32_t C, float eps,
                                   dfl_stream_t stream) {
  DFL_REQUIRE(gamma && beta && running_mean && running_var && scale && shift && C > 0, "dfl_bn_eval_prepare: bad args");
  hipLaunchKernelGGL(bn_eval_kernel, dim3((unsigned)ceil_div(C, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     gamma, beta, running_mean, running_var, scale, shift, (int)C, eps);
  return check_launch("dfl_bn_eval_prepare");
}

extern "C" int dfl_bn_bwd_finalize(const dfl_bn_bwd_finalize_args* a, dfl_stream_t stream) {
  DFL_REQUIRE(a && a->partials && a->gamma && a->save_mean && a->save_invstd && a->dgamma && a->dbeta && a->coef,
              "dfl_bn_bwd_finalize: missing pointer");
  DFL_REQUIRE(a->C > 0 && a->nblocks > 0 && a->count >= 0, "dfl_bn_bwd_finalize: bad sizes");
  const int cpb = finalize_cpb(a->C);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)ceil_div(a->C, cpb)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), *a, cpb);
  return check_launch("dfl_bn_bwd_finalize");
}

extern "C" int dfl_bn_relu_bwd_apply(const dfl_bn_relu_bwd_args* a, dfl_stream_t stream) {
  DFL_REQUIRE(a && a->dy && a->r && a->dpre && a->M > 0 && a->C > 0, "dfl_bn_relu_bwd_apply: bad args");
  DFL_REQUIRE(a->nblocks == rowblocks(a->M, a->C), "dfl_bn_relu_bwd_apply: nblocks must be dfl_rowblock_count(M, C)");
  if (a->bf16) {
    DFL_REQUIRE(a->C % 8 == 0 && a->lddy % 8 == 0 && a->ldr % 8 == 0 && a->ldo % 8 == 0 && aligned16(a->dy) && aligned16(a->r) &&
                    aligned16(a->dpre) && !a->split_out,
                "dfl_bn_relu_bwd_apply (bf16): C and ld must be multiples of 8, tensors 16-byte aligned, no split output");
    const RowGeom g8 = row_geom8(a->C);
    hipLaunchKernelGGL(bn_relu_bwd_bf16_kernel, dim3((unsigned)a->nblocks, (unsigned)g8.gy), dim3(256), 0,
                       static_cast<hipStream_t>(stream), *a, g8.UX, (int)ceil_div(a->M, a->nblocks));
    return check_launch("dfl_bn_relu_bwd_apply");
  }
  const bool vec_ok = a->lddy % 4 == 0 && a->ldr % 4 == 0 && a->ldo % 4 == 0 && aligned16(a->dy) && aligned16(a->r) &&
                      aligned16(a->dpre);
  const RowGeom g = row_geom(a->C, vec_ok);
  DFL_REQUIRE(!a->split_out || g.vec, "dfl_bn_relu_bwd_apply: split_out needs the vector layout (C, ld % 4 == 0, 16-byte alignment)");
  const int rpb = (int)ceil_div(a->M, a->nblocks);
  dim3 grid((unsigned)a->nblocks, (unsigned)g.gy);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (g.vec)
    hipLaunchKernelGGL(bn_relu_bwd_kernel<1>, grid, dim3(256), 0, s, *a, g.UX, rpb);
  else
    hipLaunchKernelGGL(bn_relu_bwd_kernel<0>, grid, dim3(256), 0, s, *a, g.UX, rpb);
  return check_launch("dfl_bn_relu_bwd_apply");
}

extern "C" int dfl_reduce_partials(const float* partials, float* out, int32_t nblocks, int32_t stride, int32_t C,
                                   dfl_stream_t stream) {
  DFL_REQUIRE(partials && out && nblocks > 0 && C > 0 && stride >= C, "dfl_reduce_partials: bad args");
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)ceil_div(C, 8)), dim3(256), 0, static_cast<hipStream_t>(stream),
                     partials, out, (int)nblocks, (int)stride, (int)C);
  return check_launch("dfl_reduce_partials");
}

extern "C" int dfl_reduce_job_blocks(int64_t n, int32_t count) {
  DFL_REQUIRE(n > 0 && count > 0, "dfl_reduce_job_blocks: bad args");
  return reduce_job_blocks(n, count);
}

extern "C" int dfl_reduce_batch(const dfl_reduce_job* jobs_dev, int32_t njobs, int32_t total_blocks, dfl_stream_t stream) {
  DFL_REQUIRE(jobs_dev && njobs > 0 && total_blocks > 0, "dfl_reduce_batch: bad args");
  hipLaunchKernelGGL(reduce_batch_kernel, dim3((unsigned)total_blocks), dim3(RB_T), 0, static_cast<hipStream_t>(stream),
                     jobs_dev, (int)njobs);
  return check_launch("dfl_reduce_batch");
}

extern "C" int dfl_affine_copy(const dfl_affine_copy_args* a, dfl_stream_t stream) {
  DFL_REQUIRE(a && a->x && a->y && a->N > 0 && a->H > 0 && a->W > 0 && a->C > 0, "dfl_affine_copy: bad args");
  DFL_REQUIRE(a->xoy >= 0 && a->xox >= 0 && a->xoy + a->H <= a->xH && a->xox + a->W <= a->xW, "dfl_affine_copy: source window");
  DFL_REQUIRE(a->yoy >= 0 && a->yox >= 0 && a->yoy + a->H <= a->yH && a->yox + a->W <= a->yW, "dfl_affine_copy: dest window");
  DFL_REQUIRE((a->scale == nullptr) == (a->shift == nullptr), "dfl_affine_copy: scale/shift go together");
  if (a->bf16) {
    DFL_REQUIRE(a->C % 8 == 0 && a->ldx % 8 == 0 && a->ldy % 8 == 0 && aligned16(a->x) && aligned16(a->y),
                "dfl_affine_copy (bf16): C and ld must be multiples of 8, tensors 16-byte aligned");
    const int cq8 = a->C / 8;
    const int64_t tot8 = (int64_t)a->N * a->H * a->W * cq8;
    hipLaunchKernelGGL(affine_copy_bf16_kernel, dim3(stream_grid(tot8)), dim3(256), 0, static_cast<hipStream_t>(stream), *a, tot8, cq8);
    return check_launch("dfl_affine_copy");
  }
  const bool vec = a->C % 4 == 0 && a->ldx % 4 == 0 && a->ldy % 4 == 0 && aligned16(a->x) && aligned16(a->y) &&
                   (a->scale == nullptr || (aligned16(a->scale) && aligned16(a->shift)));
  const int cq = vec ? a->C / 4 : a->C;
  const int64_t total = (int64_t)a->N * a->H * a->W * cq;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (vec)
    hipLaunchKernelGGL(affine_copy_kernel<1>, dim3(stream_grid(total)), dim3(256), 0, s, *a, total, cq);
  else
    hipLaunchKernelGGL(affine_copy_kernel<0>, dim3(stream_grid(total)), dim3(256), 0, s, *a, total, cq);
  return check_launch("dfl_affine_copy");
}

static int pool_common(const dfl_pool_args* a, bool bwd, bool* vec, int* cq, int64_t* total) {
  DFL_REQUIRE(a && a->x && a->y && a->N > 0 && a->H >= 2 && a->W >= 2 && a->C > 0, "dfl_maxpool2x2: bad args");
  DFL_REQUIRE(!bwd || a->dx != nullptr, "dfl_maxpool2x2_bwd: dx required");
  *vec = a->C % 4 == 0 && a->ldx % 4 == 0 && a->ldy % 4 == 0 && aligned16(a->x) && aligned16(a->y) &&
         (!bwd || (a->lddx % 4 == 0 && aligned16(a->dx)));
  *cq = *vec ? a->C / 4 : a->C;
  *total = (int64_t)a->N * (a->H / 2) * (a->W / 2) * *cq;
  return DFL_OK;
}

static int pool_bf16_check(const dfl_pool_args* a, bool bwd, int* cq, int64_t* total) {
  DFL_REQUIRE(a->C % 8 == 0 && a->ldx % 8 == 0 && a->ldy % 8 == 0 && aligned16(a->x) && aligned16(a->y) &&
                  (!bwd || (a->lddx % 8 == 0 && aligned16(a->dx))),
              "dfl_maxpool2x2 (bf16): C and ld must be multiples of 8, tensors 16-byte aligned");
  *cq = a->C / 8;
  *total = (int64_t)a->N * (a->H / 2) * (a->W / 2) * *cq;
  return DFL_OK;
}

extern "C" int dfl_maxpool2x2_fwd(const dfl_pool_args* a, dfl_stream_t stream) {
  bool vec; int cq; int64_t total;
  int rc = pool_common(a, false, &vec, &cq, &total);
  if (rc != DFL_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a->bf16) {
    rc = pool_bf16_check(a, false, &cq, &total);
    if (rc != DFL_OK) return rc;
    hipLaunchKernelGGL(maxpool_fwd_bf16_kernel, dim3(stream_grid(total)), dim3(256), 0, s, *a, total, cq);
    return check_launch("dfl_maxpool2x2_fwd");
  }
  if (vec) hipLaunchKernelGGL(maxpool_fwd_kernel<1>, dim3(stream_grid(total)), dim3(256), 0, s, *a, total, cq);
  else hipLaunchKernelGGL(maxpool_fwd_kernel<0>, dim3(stream_grid(total)), dim3(256), 0, s, *a, total, cq);
  return check_launch("dfl_maxpool2x2_fwd");
}

extern "C" int dfl_maxpool2x2_bwd(const dfl_pool_args* a, dfl_stream_t stream) {
  bool vec; int cq; int64_t total;
  int rc = pool_common(a, true, &vec, &cq, &total);
  if (rc != DFL_OK) return rc;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (a->bf16) {
    rc = pool_bf16_check(a, true, &cq, &total);
    if (rc != DFL_OK) return rc;
    hipLaunchKernelGGL(maxpool_bwd_bf16_kernel, dim3(stream_grid(total)), dim3(256), 0, s, *a, total, cq);
    return check_launch("dfl_maxpool2x2_bwd");
  }
  if (vec) hipLaunchKernelGGL(maxpool_bwd_kernel<1>, dim3(stream_grid(total)), dim3(256), 0, s, *a, total, cq);
  else hipLaunchKernelGGL(maxpool_bwd_kernel<0>, dim3(stream_grid(total)), dim3(256), 0, s, *a, total, cq);
  return check_launch("dfl_maxpool2x2_bwd");
}

extern "C" int dfl_pack_weights(const dfl_pack_job* jobs_dev, int32_t njobs, int64_t max_elems, dfl_stream_t stream) {
  DFL_REQUIRE(jobs_dev && njobs > 0 && max_elems > 0, "dfl_pack_weights: bad args");
  int64_t bx = ceil_div(max_elems, 256 * 8);
  if (bx > 1024) bx = 1024;
  if (bx < 1) bx = 1;
  hipLaunchKernelGGL(pack_kernel, dim3((unsigned)bx, (unsigned)njobs), dim3(256), 0, static_cast<hipStream_t>(stream), jobs_dev);
  return check_launch("dfl_pack_weights");
}

extern "C" int dfl_pack_weights_tiled(const dfl_pack_job* jobs_dev, int32_t njobs, int32_t total_tiles, dfl_stream_t stream) {
  DFL_REQUIRE(jobs_dev && njobs > 0 && total_tiles > 0, "dfl_pack_weights_tiled: bad args");
  hipLaunchKernelGGL(pack_tiles_kernel, dim3((unsigned)total_tiles), dim3(256), 0, static_cast<hipStream_t>(stream), jobs_dev, (int)njobs);
  return check_launch("dfl_pack_weights_tiled");
}

extern "C" int dfl_sgd_pack_tiled(const dfl_sgd_pack_args* a, dfl_stream_t stream) {
  DFL_REQUIRE(a != nullptr && a->jobs_dev != nullptr && a->njobs > 0 && a->total_tiles > 0, "dfl_sgd_pack_tiled: empty job list");
  DFL_REQUIRE(a->grad_delta % 4 == 0 && a->buf_delta % 4 == 0, "dfl_sgd_pack_tiled: gradient / momentum arenas not 16-byte congruent with the parameters");
  DFL_REQUIRE(a->momentum >= 0.f && (!a->nesterov || a->momentum > 0.f), "dfl_sgd_pack_tiled: nesterov needs a momentum");
  hipLaunchKernelGGL(sgd_pack_tiles_kernel, dim3((unsigned)a->total_tiles), dim3(256), 0, static_cast<hipStream_t>(stream), *a);
  return check_launch("dfl_sgd_pack_tiled");
}

extern "C" int dfl_sgd_step(float* p, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum,
                            float weight_decay, float grad_scale, int32_t nesterov, int32_t first_step,
                            dfl_stream_t stream) {
  DFL_REQUIRE(p && grad && n > 0, "dfl_sgd_step: bad args");
  DFL_REQUIRE(momentum == 0.f || momentum_buf != nullptr, "dfl_sgd_step: momentum buffer required");
  hipLaunchKernelGGL(sgd_kernel, dim3(stream_grid(n)), dim3(256), 0, static_cast<hipStream_t>(stream), p, grad,
                     momentum_buf, n, lr, momentum, weight_decay, grad_scale, (int)nesterov, (int)first_step);
  return check_launch("dfl_sgd_step");
}
