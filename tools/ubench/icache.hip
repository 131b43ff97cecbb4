// Is a launch slower when its CODE is cold?  (LAB_NOTES Finding 14: the slow boxes of the pool run launches of few workgroups
// 1.3-2x slower with every bandwidth / latency / launch probe of calib.hip unchanged - and calib.hip's probes all repeat ONE kernel,
// whose code stays in the instruction caches, while the UNet step launches ~40 different kernels one after the other.)
// 32 instantiations of one kernel with ~14 KB of straight-line code each (unique literals), 64 or 512 workgroups of 256 threads:
//   same   : one instantiation repeated                    (instruction caches warm)
//   rotate : the 32 instantiations round-robin             (instruction caches cold, code warm in L2: 450 KB in all)
//   flush  : round-robin with a 256-MB read between launches (code evicted from the L2s too, as by the step's 1.7 GB of weights);
//            the flush kernel's own time (measured alone) is subtracted
// all as captured graphs.  Build on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/ubench/icache.hip -o /tmp/icache
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int ID>
__global__ __launch_bounds__(256) void k_code(float* buf) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  float a = buf[t], b = a + 1.f, c = a + 2.f, d = a + 3.f;
#pragma unroll
  for (int i = 0; i < 448; ++i) {   // 4 independent chains x 448 FMAs with 32-bit literals: ~14 KB of code per instantiation
    a = __builtin_fmaf(a, 1.0f + 1e-6f * (float)(ID * 4096 + i * 4 + 0), 1e-7f * (float)(i + ID));
    b = __builtin_fmaf(b, 1.0f + 1e-6f * (float)(ID * 4096 + i * 4 + 1), 2e-7f * (float)(i + ID));
    c = __builtin_fmaf(c, 1.0f + 1e-6f * (float)(ID * 4096 + i * 4 + 2), 3e-7f * (float)(i + ID));
    d = __builtin_fmaf(d, 1.0f + 1e-6f * (float)(ID * 4096 + i * 4 + 3), 4e-7f * (float)(i + ID));
  }
  buf[t] = (a + b) + (c + d);
}
__global__ __launch_bounds__(256) void k_flush(const float4* __restrict__ src, float* sink, size_t n4) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 v = src[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) sink[0] = s;
}

typedef void (*kern_t)(float*);
template <int... I>
static std::vector<kern_t> table(std::integer_sequence<int, I...>) { return {k_code<I>...}; }

int main() {
  const std::vector<kern_t> ks = table(std::make_integer_sequence<int, 32>{});
  float* buf; CK(hipMalloc(&buf, 512 * 256 * 4)); CK(hipMemset(buf, 0, 512 * 256 * 4));
  const size_t fbytes = (size_t)256 << 20;
  float4* big; CK(hipMalloc(&big, fbytes)); CK(hipMemset(big, 1, fbytes));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 320;
  auto run = [&](int wgs, int mode, float* us) -> int {   // mode 0 same, 1 rotate, 2 rotate + flush, 3 flush only
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) {
      if (mode >= 2) hipLaunchKernelGGL(k_flush, dim3(1024), dim3(256), 0, s, big, buf, fbytes / 16);
      if (mode != 3) hipLaunchKernelGGL(ks[mode == 0 ? 0 : i % 32], dim3(wgs), dim3(256), 0, s, buf);
    }
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    *us = best * 1e3f / N;
    (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    return 0;
  };
  for (int wgs : {64, 512}) {
    float same, rot, fl, fonly;
    if (run(wgs, 0, &same) || run(wgs, 1, &rot) || run(wgs, 2, &fl) || run(wgs, 3, &fonly)) return 1;
    printf("icache %3d workgroups: same kernel %.2f us per launch, 32 kernels round-robin %.2f, round-robin behind a 256-MB flush %.2f "
           "(flush alone %.2f -> %.2f)\n", wgs, same, rot, fl, fonly, fl - fonly);
  }
  return 0;
}
