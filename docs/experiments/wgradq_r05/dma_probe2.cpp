// Probe of the inline-asm form of gfx950's LDS-DMA (buffer_load_dwordx4 ... offen lds with M0 written in the same statement):
// LDS destinations beyond 64 KB and 128 KB, out-of-range lanes, counted vmcnt waits, several waves.
//   hipcc --offload-arch=gfx950 -O2 docs/experiments/dma_probe2.cpp -o docs/experiments/bin/dma_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void dma16(i32x4 rs, unsigned voff, unsigned lds_byte) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(rs), "s"(lds_byte) : "memory");
}

__global__ void __launch_bounds__(256) k(const float* g, float* out, int nfloat, int lds_floats) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  for (int i = threadIdx.x; i < lds_floats; i += blockDim.x) smem[i] = -1.f;
  __syncthreads();
  const unsigned long long ga = (unsigned long long)g;
  i32x4 rs;                                      // raw buffer: base, stride 0, num_records (bytes), flags as make_buffer_rsrc
  rs.x = __builtin_amdgcn_readfirstlane((int)(unsigned)ga);
  rs.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(ga >> 32) & 0xffff);
  rs.z = __builtin_amdgcn_readfirstlane(nfloat * 4);
  rs.w = 0x00020000;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // three destinations: 0, 70 KB, 150 KB (+ wave KB); source: granule permuted, lanes 5/6 out of range
  const unsigned dst[3] = {0u, 70u * 1024u, 150u * 1024u};
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    unsigned voff = (unsigned)(((j * 4 + wave) * 64 + (63 - lane)) * 16);
    if (lane == 5 || lane == 6) voff = 0x80000000u;
    dma16(rs, voff, dst[j] + (unsigned)wave * 1024u);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < lds_floats; i += blockDim.x) out[i] = smem[i];
}

int main() {
  const int n = 1 << 16, ldsf = 160 * 1024 / 4;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *g, *o;
  (void)hipMalloc(&g, n * 4);
  (void)hipMalloc(&o, ldsf * 4);
  (void)hipMemcpy(g, h.data(), n * 4, hipMemcpyHostToDevice);
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k, dim3(1), dim3(256), 160 * 1024, 0, g, o, n, ldsf);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { printf("DMA_PROBE2 launch failed: %s\n", hipGetErrorString(e)); return 1; }
  std::vector<float> r(ldsf);
  (void)hipMemcpy(r.data(), o, ldsf * 4, hipMemcpyDeviceToHost);
  int bad = 0, touched = 0;
  const unsigned dst[3] = {0u, 70u * 1024u, 150u * 1024u};
  for (int j = 0; j < 3; ++j)
    for (int w = 0; w < 4; ++w)
      for (int l = 0; l < 64; ++l) {
        const float* p = &r[(dst[j] + w * 1024 + l * 16) / 4];
        const bool oob = l == 5 || l == 6;
        const float want = oob ? 0.f : (float)(((j * 4 + w) * 64 + (63 - l)) * 4);
        const bool ok = p[0] == want && p[3] == (oob ? 0.f : want + 3);
        if (!ok) { if (bad < 8) printf("dst %d wave %d lane %d: got %g..%g want %g\n", j, w, l, p[0], p[3], want); ++bad; }
      }
  for (int i = 0; i < ldsf; ++i) touched += r[i] != -1.f;
  printf("floats touched: %d (expected %d)\n", touched, 3 * 4 * 256);
  printf(bad ? "DMA_PROBE2 mismatches: %d\n" : "DMA_PROBE2 ok: asm LDS-DMA reaches 0 / 70 KB / 150 KB, out-of-range lanes write zeros (%d)\n", bad);
  return bad != 0;
}
