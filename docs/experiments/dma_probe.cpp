// Probe of gfx950's buffer_load_dwordx4 ... lds (global -> LDS without VGPRs): where does lane l's 16 bytes land, what do
// out-of-range lanes write, does an instruction offset move the LDS address or the global one?
//   hipcc --offload-arch=gfx950 -O2 -x hip tools/exp/dma_probe.cpp -o tools/exp/bin/dma_probe && tools/exp/bin/dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__global__ void k(const float* g, float* out, int nfloat) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) smem[i] = -1.f;
  __syncthreads();
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, nfloat * 4, 0x00020000);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // lane l fetches granule (63 - l) of its wave's 1 KiB; lanes 5 and 6 go out of range
  unsigned voff = (unsigned)((wave * 64 + (63 - lane)) * 16);
  if (lane == 5 || lane == 6) voff = 0x80000000u;
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + wave * 256), 16, voff, 0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = smem[i];
}

int main() {
  const int n = 4096;
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = (float)i;
  float *g, *o;
  hipMalloc(&g, n * 4);
  hipMalloc(&o, 1024 * 4);
  hipMemcpy(g, h.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(128), 4096, 0, g, o, n);
  std::vector<float> r(1024);
  hipMemcpy(r.data(), o, 1024 * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int w = 0; w < 2; ++w)
    for (int l = 0; l < 64; ++l) {
      const float* p = &r[w * 256 + l * 4];
      const float want = (l == 5 || l == 6) ? 0.f : (float)((w * 64 + (63 - l)) * 4);
      const bool ok = p[0] == want && (l == 5 || l == 6 ? p[3] == 0.f : p[3] == want + 3);
      if (!ok) { if (bad < 8) printf("wave %d lane %d: got %g %g %g %g want %g..\n", w, l, p[0], p[1], p[2], p[3], want); ++bad; }
    }
  printf("untouched tail: %g %g\n", r[512], r[1023]);
  printf(bad ? "DMA_PROBE mismatches: %d\n" : "DMA_PROBE ok: lane l lands at base + 16*l, out-of-range lanes write zeros (%d)\n", bad);
  return bad != 0;
}
