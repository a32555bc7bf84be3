// Grid barrier probe for a persistent multi-stage kernel (round 5): 256 workgroups x 512 threads, NB stages; in every stage a
// thread writes a value that a workgroup of ANOTHER XCD reads in the next one (coherence check), between them a grid barrier:
//   0  one atomic counter, every wave fences (agent-scope release before, acquire after)
//   1  one atomic counter, only wave 0 fences at agent scope (the others meet it at the workgroup barrier)
//   2  one atomic counter, no fences at all (cost of the atomics alone; the data check may fail)
//   3  flags, all-to-all: workgroup b stores flag[b] = epoch, its threads 0..nwg-1 poll one flag each; wave 0 fences
//   4  like 3 without fences
//   5  like 1 with s_sleep(8) in the poll loop
//   6  like 2 (no fences), but the data go through write-through stores / cache-bypassing loads (relaxed agent-scope atomics: sc1)
//   7  like 1 with the release fence only (buffer_wbl2 sc1)        8  like 1 with the acquire fence only (buffer_inv sc1)
// hipcc --offload-arch=gfx950 -O3 -o docs/experiments/bin/barrier_probe docs/experiments/infer192_r05/barrier_probe.cpp
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>
__device__ __forceinline__ void grid_barrier(unsigned* bar, unsigned* flags, unsigned epoch, int nwg) {
  if (MODE == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * nwg) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  } else if (MODE == 1 || MODE == 2 || MODE == 5 || MODE == 6 || MODE == 7 || MODE == 8) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x < 64) {
      if (MODE == 1 || MODE == 5 || MODE == 7) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch * nwg) __builtin_amdgcn_s_sleep(MODE == 5 ? 8 : 1);
      }
      if (MODE == 1 || MODE == 5 || MODE == 8) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __syncthreads();
    if (threadIdx.x < 64) {
      if (MODE == 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      if (threadIdx.x == 0) __hip_atomic_store(flags + blockIdx.x, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if ((int)threadIdx.x < nwg) {
      while (__hip_atomic_load(flags + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < epoch) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    if (threadIdx.x < 64 && MODE == 3) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  }
}

template <int MODE>
__global__ void __launch_bounds__(512, 1) probe(unsigned* bar, unsigned* flags, int* data, int nb, int* errs) {
  const int nwg = gridDim.x, b = blockIdx.x;
  int bad = 0;
  for (int s = 0; s < nb; ++s) {
    if (MODE == 6) __hip_atomic_store(data + ((s & 1) * nwg + b) * 512 + threadIdx.x, s * 1000 + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else data[((s & 1) * nwg + b) * 512 + threadIdx.x] = s * 1000 + b;
    grid_barrier<MODE>(bar, flags, (unsigned)(s + 1), nwg);
    const int o = (b + 1 + s) % nwg;
    const int v = MODE == 6 ? __hip_atomic_load(data + ((s & 1) * nwg + o) * 512 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                            : data[((s & 1) * nwg + o) * 512 + threadIdx.x];
    if (v != s * 1000 + o) ++bad;
  }
  if (bad) atomicAdd(errs, bad);
}

int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 100;
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int nwg = prop.multiProcessorCount;
  unsigned *bar, *flags;
  int *data, *errs;
  hipMalloc(&bar, 4);
  hipMalloc(&flags, 4096);
  hipMalloc(&data, 2 * nwg * 512 * 4);
  hipMalloc(&errs, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int mode = 0; mode < 9; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipMemset(bar, 0, 4);
      hipMemset(flags, 0, 4096);
      hipMemset(errs, 0, 4);
      hipMemset(data, 0xff, 2 * nwg * 512 * 4);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      switch (mode) {
        case 0: hipLaunchKernelGGL(probe<0>, dim3(nwg), dim3(512), 0, 0, bar, flags, data, nb, errs); break;
        case 1: hipLaunchKernelGGL(probe<1>, dim3(nwg), dim3(512), 0, 0, bar, flags, data, nb, errs); break;
        case 2: hipLaunchKernelGGL(probe<2>, dim3(nwg), dim3(512), 0, 0, bar, flags, data, nb, errs); break;
        case 3: hipLaunchKernelGGL(probe<3>, dim3(nwg), dim3(512), 0, 0, bar, flags, data, nb, errs); break;
        case 4: hipLaunchKernelGGL(probe<4>, dim3(nwg), dim3(512), 0, 0, bar, flags, data, nb, errs); break;
        case 5: hipLaunchKernelGGL(probe<5>, dim3(nwg), dim3(512), 0, 0, bar, flags, data, nb, errs); break;
        case 6: hipLaunchKernelGGL(probe<6>, dim3(nwg), dim3(512), 0, 0, bar, flags, data, nb, errs); break;
        case 7: hipLaunchKernelGGL(probe<7>, dim3(nwg), dim3(512), 0, 0, bar, flags, data, nb, errs); break;
        default: hipLaunchKernelGGL(probe<8>, dim3(nwg), dim3(512), 0, 0, bar, flags, data, nb, errs); break;
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0.f;
      hipEventElapsedTime(&ms, e0, e1);
      int h = 0;
      hipMemcpy(&h, errs, 4, hipMemcpyDeviceToHost);
      printf("mode %d: %d workgroups, %d stages: %.1f us total, %.2f us per stage, stale reads %d\n", mode, nwg, nb, ms * 1e3, ms * 1e3 / nb, h);
    }
  }
  return 0;
}
