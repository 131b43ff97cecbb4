// Dependent-launch floor on gfx950: N back-to-back kernels on one stream (eager and as a HIP graph),
// empty vs tiny-memory kernels, 1 vs 256 workgroups.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_empty() {}
__global__ void k_touch(float* p) { p[blockIdx.x * blockDim.x + threadIdx.x] += 1.f; }
int main() {
  float* d; CK(hipMalloc(&d, 1 << 22)); CK(hipMemset(d, 0, 1 << 22));
  hipStream_t s; CK(hipStreamCreate(&s));
  const int N = 2000;
  for (int mode = 0; mode < 4; ++mode) {
    const int blocks = (mode & 1) ? 256 : 1;
    const bool touch = mode & 2;
    auto launch = [&]() { if (touch) hipLaunchKernelGGL(k_touch, dim3(blocks), dim3(256), 0, s, d); else hipLaunchKernelGGL(k_empty, dim3(blocks), dim3(256), 0, s); };
    for (int i = 0; i < 100; ++i) launch();
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < N; ++i) launch();
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("eager  %s blocks %3d: %.2f us per kernel\n", touch ? "touch" : "empty", blocks, ms * 1e3 / N);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) launch();
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("graph  %s blocks %3d: %.2f us per kernel\n", touch ? "touch" : "empty", blocks, ms * 1e3 / N);
  }
  return 0;
}
