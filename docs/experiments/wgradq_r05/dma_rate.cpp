// Fill-rate probe: how fast can ONE workgroup per CU stream memory into its LDS (LDS-DMA) or its registers (plain 16-byte
// buffer loads), as a function of waves per workgroup and pieces in flight per wave?  Source: contiguous per workgroup
// (HBM-sized working set) or one 4 MB region shared by all (L2 / MALL resident).
//   hipcc --offload-arch=gfx950 -O2 docs/experiments/dma_rate.cpp -o docs/experiments/bin/dma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(i32x4 rs, unsigned voff, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_byte) : "memory");
}

template <int DEPTH>
__device__ __forceinline__ void wait_depth() {
  if constexpr (DEPTH == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (DEPTH == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (DEPTH == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (DEPTH == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if constexpr (DEPTH == 16) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(31)" ::: "memory");
}

// every wave streams `pieces` KiB: piece i of wave w of workgroup b at byte (b * stride_wg + (i * nwaves + w) * 1024) % span
template <int DEPTH, bool LDS>
__global__ void __launch_bounds__(1024) k(const float* g, unsigned long long nbytes, unsigned stride_wg, unsigned span, int pieces, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const unsigned long long ga = (unsigned long long)g;
  i32x4 rs;
  rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ga);
  rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(ga >> 32) & 0xffff);
  rs.z = __builtin_amdgcn_readfirstlane((int)(unsigned)nbytes);
  rs.w = 0x00020000;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const unsigned base = blockIdx.x * stride_wg;
  u32x4 acc = {0, 0, 0, 0};
  for (int i = 0; i < pieces; ++i) {
    unsigned off = (unsigned)((i * nw + wave) * 1024) % span;
    unsigned voff = (base + off) % (unsigned)nbytes + lane * 16;
    if constexpr (LDS) {
      dma16(rs, voff, (unsigned)(((i % DEPTH) * nw + wave) * 1024));
      wait_depth<DEPTH>();
    } else {
      __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, (int)nbytes, 0x00020000);
      u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r2, voff, 0, 0);
      acc += v;
    }
  }
  if constexpr (LDS) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0 && smem[17] == 123 && sink) sink[0] = 1.f;
  } else {
    if (acc.x + acc.y + acc.z + acc.w == 0x12345u && sink) sink[0] = 1.f;
  }
}

template <int DEPTH, bool LDS>
double run(const float* g, unsigned long long nbytes, int nwg, int nwaves, unsigned stride_wg, unsigned span, int pieces) {
  auto kk = k<DEPTH, LDS>;
  (void)hipFuncSetAttribute((const void*)kk, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  const size_t lds = LDS ? (size_t)DEPTH * nwaves * 1024 : 0;
  if (lds > 160 * 1024) return -1;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(kk, dim3(nwg), dim3(nwaves * 64), lds, 0, g, nbytes, stride_wg, span, pieces, (float*)nullptr);
  (void)hipEventRecord(e0);
  const int reps = 5;
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kk, dim3(nwg), dim3(nwaves * 64), lds, 0, g, nbytes, stride_wg, span, pieces, (float*)nullptr);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  return (double)nwg * nwaves * pieces * 1024.0 * reps / (ms * 1e-3) / 1e12;
}

int main() {
  const unsigned long long nbytes = 1ull << 30;     // 1 GiB
  float* g;
  (void)hipMalloc(&g, nbytes);
  (void)hipMemset(g, 1, nbytes);
  const int nwg = 256;
  printf("TB/s, 256 workgroups (one per CU); columns: pieces in flight per wave 1 2 4 8 16 32\n");
  for (int hbm = 1; hbm >= 0; --hbm) {
    const unsigned stride = hbm ? 4u << 20 : 16384u, span = hbm ? 4u << 20 : 4u << 20;   // HBM: 4 MB of its own per workgroup; cached: all share 4 MB
    for (int lds = 1; lds >= 0; --lds)
      for (int nw : {4, 8, 12, 16}) {
        const int pieces = hbm ? (4 << 10) / nw : (4 << 10) / nw;      // 4 MB per workgroup per launch
        printf("%s %s waves %2d:", hbm ? "HBM   " : "cached", lds ? "LDS-DMA" : "regs   ", nw);
        double r[6];
        if (lds) {
          r[0] = run<1, true>(g, nbytes, nwg, nw, stride, span, pieces); r[1] = run<2, true>(g, nbytes, nwg, nw, stride, span, pieces);
          r[2] = run<4, true>(g, nbytes, nwg, nw, stride, span, pieces); r[3] = run<8, true>(g, nbytes, nwg, nw, stride, span, pieces);
          r[4] = run<16, true>(g, nbytes, nwg, nw, stride, span, pieces); r[5] = run<32, true>(g, nbytes, nwg, nw, stride, span, pieces);
        } else {
          r[0] = run<1, false>(g, nbytes, nwg, nw, stride, span, pieces);
          for (int i = 1; i < 6; ++i) r[i] = -1;
        }
        for (int i = 0; i < 6; ++i) printf(" %6.2f", r[i]);
        printf("\n");
      }
  }
  return 0;
}
