// Ablation micro-benchmark for the fp32 MFMA gather-GEMM main loop (tuning aid, not part of the library).
// Variants add one ingredient at a time so the cost of each shows up as a drop in TFLOP/s:
//   V0 MFMA only (operands in registers)        V1 + operand reads from LDS (ds_read_b32)
//   V2 + LDS refill (ds_write) + one barrier per chunk     V3 + global loads feeding the refill (prefetch 1 chunk)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/exp_mfma.cpp -o /tmp/exp_mfma ; run: /tmp/exp_mfma
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int KC = 16;

template <int VAR, int WM, int WN, int TM, int TN, int OCC>
__global__ void __launch_bounds__(WM* WN * 64, OCC) k(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                     int nchunks, int ldx) {
  constexpr int NT = WM * WN * 64, BM = WM * TM * 32, BN = WN * TN * 32, LDA = BM + 4, LDB = BN + 4;
  constexpr int QA = BM * 4 / NT, NQB = KC * BN / 4, QB = (NQB + NT - 1) / NT, BQ = BN / 4;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* As = smem;
  float* Bs = smem + 2 * KC * LDA;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / WN, wn = wave % WN, li = lane & 31, lh = lane >> 5;
  f32x16 acc[TM][TN];
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j)
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  // fill LDS once
  for (int i = tid; i < 2 * KC * LDA + 2 * KC * LDB; i += NT) smem[i] = 0.001f * (float)(i & 15);
  __syncthreads();
  float4 ra[QA], rb[QB];
  const int aq = tid & 3;
  const float* xp = x + ((size_t)blockIdx.x * BM + (tid >> 2)) * ldx + 4 * aq;
  auto load = [&](int ch) {
#pragma unroll
    for (int r = 0; r < QA; ++r) {
      if (VAR >= 5) {   // 3x3 gather on a [pixels][64] tensor, image width 96: chunk -> (tap, channel block)
        const int tap = (ch * KC) / ldx, c = (ch * KC) % ldx;
        const long off = ((long)(tap / 3) * 96 + (tap % 3)) * ldx + c;
        ra[r] = *reinterpret_cast<const float4*>(xp + (size_t)r * (NT / 4) * ldx + off);
      } else {
        ra[r] = *reinterpret_cast<const float4*>(xp + (size_t)r * (NT / 4) * ldx + (ch * KC) % ldx);
      }
    }
#pragma unroll
    for (int r = 0; r < QB; ++r) {
      const int idx = (tid + r * NT) % NQB;
      rb[r] = *reinterpret_cast<const float4*>(w + ((size_t)ch * KC + idx / BQ) * BN + 4 * (idx % BQ));
    }
  };
  auto store = [&](int buf) {
    float* Ab = As + buf * KC * LDA;
#pragma unroll
    for (int r = 0; r < QA; ++r) {
      const int row = (tid >> 2) + r * (NT / 4);
      Ab[(4 * aq + 0) * LDA + row] = ra[r].x;
      Ab[(4 * aq + 1) * LDA + row] = ra[r].y;
      Ab[(4 * aq + 2) * LDA + row] = ra[r].z;
      Ab[(4 * aq + 3) * LDA + row] = ra[r].w;
    }
    float* Bb = Bs + buf * KC * LDB;
#pragma unroll
    for (int r = 0; r < QB; ++r) {
      const int idx = tid + r * NT;
      if (idx < NQB) *reinterpret_cast<float4*>(Bb + (idx / BQ) * LDB + 4 * (idx % BQ)) = rb[r];
    }
  };
  for (int r = 0; r < QA; ++r) ra[r] = make_float4(1.f, 2.f, 3.f, 4.f);
  for (int r = 0; r < QB; ++r) rb[r] = make_float4(1.f, 2.f, 3.f, 4.f);
  float av0 = 0.5f + lane, bv0 = 0.25f + lane;
  for (int ch = 0; ch < nchunks; ++ch) {
    const int buf = ch & 1;
    if (VAR >= 3) load(ch + 1 < nchunks ? ch + 1 : ch);
    __builtin_amdgcn_sched_barrier(0);
    const float* Ab = As + buf * KC * LDA + wm * (TM * 32) + li;
    const float* Bb = Bs + buf * KC * LDB + wn * (TN * 32) + li;
#pragma unroll
    for (int kk = 0; kk < KC / 2; ++kk) {
      float av[TM], bv[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) av[i] = (VAR >= 1) ? Ab[(2 * kk + lh) * LDA + i * 32] : av0;
#pragma unroll
      for (int j = 0; j < TN; ++j) bv[j] = (VAR >= 1) ? Bb[(2 * kk + lh) * LDB + j * 32] : bv0;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (VAR >= 2) {
      store(buf ^ 1);
      __syncthreads();
    }
  }
  if (VAR >= 4) {   // real tile store: [BM rows][BN] floats per block, 128-byte segments per half wave
    for (int i = 0; i < TM; ++i)
      for (int j = 0; j < TN; ++j)
        for (int r = 0; r < 16; ++r) {
          const int row = wm * TM * 32 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, col = wn * TN * 32 + j * 32 + li;
          y[((size_t)blockIdx.x * BM + row) * BN + col] = fmaxf(acc[i][j][r], 0.f);
        }
    return;
  }
  float s = 0.f;
  for (int i = 0; i < TM; ++i)
    for (int j = 0; j < TN; ++j)
      for (int r = 0; r < 16; ++r) s += acc[i][j][r];
  y[(size_t)blockIdx.x * NT + tid] = s;
}

template <int VAR, int WM, int WN, int TM, int TN, int OCC>
void run(const char* name, int blocks, int nchunks, const float* x, const float* w, float* y, int ldx) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
  const size_t lds = (size_t)(2 * KC * (BM + 4) + 2 * KC * (BN + 4)) * 4;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<VAR, WM, WN, TM, TN, OCC>), dim3(blocks), dim3(WM * WN * 64), lds, 0, x, w, y, nchunks, ldx);
  hipEventRecord(e0);
  const int reps = 10;
  for (int it = 0; it < reps; ++it) hipLaunchKernelGGL((k<VAR, WM, WN, TM, TN, OCC>), dim3(blocks), dim3(WM * WN * 64), lds, 0, x, w, y, nchunks, ldx);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  const double fl = 2.0 * blocks * (double)BM * BN * KC * nchunks;
  printf("%-34s blocks %5d chunks %4d : %8.4f ms  %7.1f TFLOP/s\n", name, blocks, nchunks, ms, fl / ms / 1e9);
}

int main() {
  const int ldx = 64;
  float *x, *w, *y;
  hipMalloc(&x, (size_t)4096 * 256 * ldx * 4 + 65536);
  hipMalloc(&w, (size_t)1024 * 128 * 4 * 16);
  hipMalloc(&y, (size_t)4608 * 256 * 128 * 4);
  hipMemset(x, 0, (size_t)4096 * 256 * ldx * 4 + 65536);
  hipMemset(w, 0, (size_t)1024 * 128 * 4 * 16);
  run<4, 2, 2, 2, 1, 5>("V4 +tile store      128x64", 1152, 36, x, w, y, ldx);
  run<5, 2, 2, 2, 1, 5>("V5 +3x3 gather 37MB 128x64", 1152, 36, x, w, y, ldx);
  run<4, 2, 2, 2, 1, 5>("V4 +tile store      128x64 K288", 1152, 18, x, w, y, ldx);
  run<5, 4, 1, 2, 1, 5>("V5 +3x3 gather      256x32 K576", 2304, 36, x, w, y, ldx);
  run<5, 2, 2, 1, 1, 6>("V5 +3x3 gather      64x64 K576", 2304, 36, x, w, y, ldx);
  for (int blocks : {1152}) {
    run<0, 2, 2, 2, 1, 5>("V0 mfma only        128x64", blocks, 36, x, w, y, ldx);
    run<1, 2, 2, 2, 1, 5>("V1 +ds_read         128x64", blocks, 36, x, w, y, ldx);
    run<2, 2, 2, 2, 1, 5>("V2 +ds_write+barrier 128x64", blocks, 36, x, w, y, ldx);
    run<3, 2, 2, 2, 1, 5>("V3 +global loads    128x64", blocks, 36, x, w, y, ldx);
  }
  run<0, 2, 2, 2, 2, 2>("V0 mfma only        128x128", 512, 72, x, w, y, ldx);
  run<1, 2, 2, 2, 2, 2>("V1 +ds_read         128x128", 512, 72, x, w, y, ldx);
  run<2, 2, 2, 2, 2, 2>("V2 +ds_write+barrier 128x128", 512, 72, x, w, y, ldx);
  run<3, 2, 2, 2, 2, 2>("V3 +global loads    128x128", 512, 72, x, w, y, ldx);
  run<0, 2, 2, 1, 1, 6>("V0 mfma only        64x64", 1536, 72, x, w, y, ldx);
  run<1, 2, 2, 1, 1, 6>("V1 +ds_read         64x64", 1536, 72, x, w, y, ldx);
  run<2, 2, 2, 1, 1, 6>("V2 +ds_write+barrier 64x64", 1536, 72, x, w, y, ldx);
  run<3, 2, 2, 1, 1, 6>("V3 +global loads    64x64", 1536, 72, x, w, y, ldx);
  run<3, 2, 2, 1, 1, 6>("V3 +global loads    64x64", 4608, 36, x, w, y, ldx);
  return 0;
}
