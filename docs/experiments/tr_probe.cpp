// Probe of ds_read_b64_tr_b16 semantics on gfx950: lds[i] = i (16-bit), every lane passes the address of 4 contiguous
// elements; prints what each lane receives.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(s4* out, int pitch) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x;
  const int i = l & 15, g = l >> 4;
  // group g: 4 rows (row = i/4) of `pitch` elements, 16 columns starting at 16*g... each lane: row i/4, quad i%4
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + (i / 4) * pitch + g * 16 + (i % 4) * 4));
  out[l] = v;
}
int main() {
  s4* d;
  hipMalloc(&d, 64 * sizeof(s4));
  const int pitch = 100;
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, pitch);
  s4 h[64];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %5d %5d %5d %5d   (expect col %d of rows 0..3: %d %d %d %d)\n", l, h[l].x, h[l].y, h[l].z, h[l].w,
                                      l, (l >> 4) * 16 + (l & 15), pitch + (l >> 4) * 16 + (l & 15), 2 * pitch + (l >> 4) * 16 + (l & 15), 3 * pitch + (l >> 4) * 16 + (l & 15));
  return 0;
}
