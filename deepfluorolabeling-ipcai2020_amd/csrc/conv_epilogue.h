// Pieces shared by the convolution kernels (conv_gemm.hip, conv_rows.hip): kernel argument block, bounds-checked
// buffer loads, the simple epilogue (bias, ReLU, NHWC store, per-channel statistics) and the statistics reduction.
#pragma once
#include "common.h"

namespace dfl {

constexpr int KC = 16;
constexpr uint32_t OOB = 0x80000000u;   // buffer offset beyond any tensor we accept (< 2 GiB): the load returns 0

struct ConvK {
  dfl_conv_args a;
  int Mtot, Ktot, Hg, Wg, Cout;
  int fast;          // MODE 1 preconditions hold
  int so_simple;
  int splits, cps;   // split-K: number of K slices and chunks per slice
  uint32_t x_bytes, w_bytes, y_bytes, so_bytes;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rs, uint32_t voff, uint32_t soff) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// MFMA accumulator tiles of one wave -> y: bias, ReLU, plain NHWC store, statistics of the stored value (sum v, sum v*u
// with u = v or a partner tensor) accumulated into s1 / s2 per column of the lane.  Out-of-range rows / columns get an
// out-of-range buffer offset: the store is dropped by the hardware, no branches.
template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void conv_epilogue_simple(const ConvK& p, f32x16 (&acc)[TM][TN], float (&s1)[TN], float (&s2)[TN],
                                                     int m0, int n0, int wm, int wn, int li, int lh) {
  const dfl_conv_args& a = p.a;
  const int Ntot = a.Ntot;
  __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)p.y_bytes, 0x00020000);
  uint32_t cb[TN];
  float cbias[TN];
  bool cok[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int n = n0 + wn * (TN * 32) + j * 32 + li;
    cok[j] = n < Ntot;
    cb[j] = (uint32_t)n * 4u;
    cbias[j] = (a.bias != nullptr && cok[j]) ? a.bias[n] : 0.f;
  }
  const uint32_t ldyb = (uint32_t)a.ldy * 4u;
  if (a.stat_other == nullptr) {
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int mb = m0 + wm * (TM * 32) + i * 32 + 8 * g + 4 * lh;
        const uint32_t rowb = (uint32_t)mb * ldyb;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const bool rok = (mb + rr) < p.Mtot;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            float v = acc[i][j][4 * g + rr] + cbias[j];
            if (a.relu) v = fmaxf(v, 0.f);
            const bool ok = rok && cok[j];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsY, ok ? rowb + (uint32_t)rr * ldyb + cb[j] : OOB, 0, 0);
            const float vm = ok ? v : 0.f;
            s1[j] += vm;
            s2[j] = fmaf(vm, vm, s2[j]);
          }
        }
      }
    }
  } else {
    // statistics against a partner tensor u (sum v, sum v*u): u comes through a bounds-checked descriptor, the
    // loads of a whole 32-row tile are issued before the first one is consumed
    __amdgpu_buffer_rsrc_t rsU = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.stat_other), 0, (int)p.so_bytes, 0x00020000);
    const uint32_t ldub = (uint32_t)a.ldso * 4u;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      float u[16][TN];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int mb = m0 + wm * (TM * 32) + i * 32 + 8 * g + 4 * lh;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const bool rok = (mb + rr) < p.Mtot;
#pragma unroll
          for (int j = 0; j < TN; ++j)
            u[4 * g + rr][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                rsU, (rok && cok[j]) ? (uint32_t)(mb + rr) * ldub + cb[j] : OOB, 0, 0));
        }
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int mb = m0 + wm * (TM * 32) + i * 32 + 8 * g + 4 * lh;
        const uint32_t rowb = (uint32_t)mb * ldyb;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const bool rok = (mb + rr) < p.Mtot;
#pragma unroll
          for (int j = 0; j < TN; ++j) {
            float v = acc[i][j][4 * g + rr] + cbias[j];
            if (a.relu) v = fmaxf(v, 0.f);
            const bool ok = rok && cok[j];
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rsY, ok ? rowb + (uint32_t)rr * ldyb + cb[j] : OOB, 0, 0);
            const float vm = ok ? v : 0.f;
            s1[j] += vm;
            s2[j] = fmaf(vm, u[4 * g + rr][j], s2[j]);
          }
        }
      }
    }
  }
}

// Per-column sums of the workgroup -> one row of stat_partials (dfl_bn_finalize adds the rows in fp64).  Reuses the
// staging LDS: every MFMA read of it lies behind the last loop barrier.
template <int WM, int WN, int TM, int TN>
__device__ __forceinline__ void conv_stats_tail(const ConvK& p, const float (&s1)[TN], const float (&s2)[TN], float* smem,
                                                int tid, int n0, int wm, int wn, int li, int lh) {
  constexpr int NT = WM * WN * 64, BN = WN * TN * 32;
  const dfl_conv_args& a = p.a;
  const int Ntot = a.Ntot;
  float* red = smem;  // [WM][2][BN]
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const float t1 = s1[j] + xor32(s1[j]);
    const float t2 = s2[j] + xor32(s2[j]);
    if (lh == 0) {
      const int col = wn * (TN * 32) + j * 32 + li;
      red[(wm * 2 + 0) * BN + col] = t1;
      red[(wm * 2 + 1) * BN + col] = t2;
    }
  }
  __syncthreads();
  for (int idx = tid; idx < 2 * BN; idx += NT) {
    const int which = idx / BN, col = idx - which * BN;
    const int n = n0 + col;
    if (n < Ntot) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) s += red[(w * 2 + which) * BN + col];
      if (a.stat_totals != nullptr) bn_live_add(a.stat_totals, (int)blockIdx.x, which, Ntot, n, s);     // live statistics (include/dfl_hip.h)
      else a.stat_partials[((int64_t)blockIdx.x * 2 + which) * Ntot + n] = s;
    }
  }
}

}  // namespace dfl
