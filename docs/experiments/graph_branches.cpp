// Does a hipGraph with two independent branches run them concurrently on gfx950 / ROCm 7.2, and at what cost per edge?
// Two chains of N small kernels (each ~5 us: a dependent load -> reduce -> store, 32 workgroups) captured (a) on one stream,
// (b) on two streams forked / joined with events inside ONE capture; plus a "big + small" mix: a 40 us kernel filling the chip
// beside a chain of eight small ones.   hipcc --offload-arch=gfx950 -O2 graph_branches.cpp -o graph_branches && ./graph_branches
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void small_k(const float* in, float* out, int n) {
  __shared__ float red[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) s += in[blockIdx.x * n + i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) out[blockIdx.x] = red[0];
}
__global__ void big_k(float* buf, int iters) {
  float v = buf[blockIdx.x * 256 + threadIdx.x];
  for (int i = 0; i < iters; ++i) v = fmaf(v, 1.0001f, 0.5f);
  buf[blockIdx.x * 256 + threadIdx.x] = v;
}

int main() {
  const int N = 20, n = 4096, blocks = 32;
  float *a, *b, *big;
  CK(hipMalloc(&a, (size_t)blocks * n * 4 * 2)); CK(hipMalloc(&b, (size_t)blocks * n * 4 * 2)); CK(hipMalloc(&big, 1024 * 256 * 4));
  CK(hipMemset(a, 0, (size_t)blocks * n * 8)); CK(hipMemset(b, 0, (size_t)blocks * n * 8)); CK(hipMemset(big, 0, 1024 * 256 * 4));
  hipStream_t s0, s1;
  CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t fork, join, t0, t1;
  CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  auto chain = [&](hipStream_t s, float* p, int cnt) { for (int i = 0; i < cnt; ++i) hipLaunchKernelGGL(small_k, dim3(blocks), dim3(256), 0, s, p, p + blocks * n, n); };
  struct Case { const char* name; int kind; };
  const Case cases[] = {{"serial: 2 x 20 small kernels on one stream", 0}, {"branches: 20 + 20 small kernels on two streams", 1},
                        {"serial: 8 x (big + small)", 2}, {"branches: 8 big || 8 small", 3}, {"one chain of 20 small", 4}, {"8 big alone", 5}};
  for (const Case& c : cases) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    if (c.kind == 0) { chain(s0, a, N); chain(s0, b, N); }
    if (c.kind == 4) { chain(s0, a, N); }
    if (c.kind == 1) {
      CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0));
      chain(s0, a, N); chain(s1, b, N);
      CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0));
    }
    if (c.kind == 2) for (int i = 0; i < 8; ++i) { hipLaunchKernelGGL(big_k, dim3(1024), dim3(256), 0, s0, big, 6000); chain(s0, a, 1); }
    if (c.kind == 5) for (int i = 0; i < 8; ++i) { hipLaunchKernelGGL(big_k, dim3(1024), dim3(256), 0, s0, big, 6000); }
    if (c.kind == 3) {
      CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0));
      for (int i = 0; i < 8; ++i) hipLaunchKernelGGL(big_k, dim3(1024), dim3(256), 0, s0, big, 6000);
      chain(s1, a, 8);
      CK(hipEventRecord(join, s1)); CK(hipStreamWaitEvent(s0, join, 0));
    }
    CK(hipStreamEndCapture(s0, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int w = 0; w < 5; ++w) CK(hipGraphLaunch(ge, s0));
    CK(hipStreamSynchronize(s0));
    CK(hipEventRecord(t0, s0));
    const int reps = 50;
    for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, s0));
    CK(hipEventRecord(t1, s0)); CK(hipStreamSynchronize(s0));
    float ms; CK(hipEventElapsedTime(&ms, t0, t1));
    size_t nn = 0; hipGraphGetNodes(g, nullptr, &nn);
    printf("%-52s %8.1f us per launch (%zu nodes)\n", c.name, ms / reps * 1e3, nn);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
  }
  return 0;
}
// Measured (MI355X, ROCm 7.2, round 4):
//   serial: 2 x 20 small kernels on one stream              148.5 us per launch (40 nodes)
//   branches: 20 + 20 small kernels on two streams          142.3 us per launch (40 nodes)
//   serial: 8 x (big + small)                               952.0 us
//   branches: 8 big || 8 small                              888.6 us
//   one chain of 20 small                                    78.8 us          8 big alone   922.0 us
// A graph's branches are NOT executed concurrently: the nodes of both branches run one after the other (3.9 us per small
// kernel either way).  Putting a layer's weight gradient, its BatchNorm finalize chain or the small 1x1 / 2x2 kernels on a
// second branch of the backward graph therefore cannot overlap them with the 3x3 kernels.
