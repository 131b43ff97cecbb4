// What turns the 1.5 us dependent-launch floor into the ~4 us seen in the UNet graph?
// alternate two different kernels; 264-B by-value kernarg; kernels with a big (never executed) code
// body; 48 KB dynamic LDS; 1 thread; 2048 blocks; a kernel that writes 8 MB before a trivial one.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Big { float v[64]; int n; float* p; };
__global__ void k_a(float* p) { if (p && threadIdx.x == 9999) p[0] = 1.f; }
__global__ void k_b(float* p) { if (p && threadIdx.x == 9998) p[1] = 2.f; }
__global__ void k_big(Big b) { if (b.n == 12345) b.p[0] = b.v[3]; }
__global__ void k_lds(float* p) { extern __shared__ float sm[]; if (threadIdx.x == 9999) { sm[0] = 1; p[0] = sm[1]; } }
__global__ void k_write(float4* p) { p[blockIdx.x * 256 + threadIdx.x] = float4{1, 2, 3, 4}; }
template <int N> __global__ void k_code(float* p, int sel) {
  float x = p[threadIdx.x];
  if (sel == 77) {
#pragma unroll
    for (int i = 0; i < N; ++i) x = x * 1.0001f + (float)i;
  }
  if (sel == 78) p[threadIdx.x] = x;
}
int main() {
  float* d; CK(hipMalloc(&d, 64 << 20)); CK(hipMemset(d, 0, 64 << 20));
  hipStream_t s; CK(hipStreamCreate(&s));
  const int N = 2000;
  Big big{}; big.p = d;
  CK(hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  const char* names[] = {"same kernel", "alternate 2 kernels", "264-B kernarg", "alternate 2 big-code kernels", "48 KB dyn LDS",
                         "1 thread", "2048 blocks", "8 MB writer + trivial (per pair)"};
  for (int mode = 0; mode < 8; ++mode) {
    auto launch = [&](int i) {
      switch (mode) {
        case 0: hipLaunchKernelGGL(k_a, dim3(256), dim3(256), 0, s, d); break;
        case 1: if (i & 1) hipLaunchKernelGGL(k_a, dim3(256), dim3(256), 0, s, d); else hipLaunchKernelGGL(k_b, dim3(256), dim3(256), 0, s, d); break;
        case 2: hipLaunchKernelGGL(k_big, dim3(256), dim3(256), 0, s, big); break;
        case 3: if (i & 1) hipLaunchKernelGGL(k_code<4000>, dim3(256), dim3(256), 0, s, d, 0); else hipLaunchKernelGGL(k_code<4001>, dim3(256), dim3(256), 0, s, d, 0); break;
        case 4: hipLaunchKernelGGL(k_lds, dim3(256), dim3(256), 49152, s, d); break;
        case 5: hipLaunchKernelGGL(k_a, dim3(1), dim3(1), 0, s, d); break;
        case 6: hipLaunchKernelGGL(k_a, dim3(2048), dim3(256), 0, s, d); break;
        case 7: if (i & 1) hipLaunchKernelGGL(k_a, dim3(1), dim3(64), 0, s, d); else hipLaunchKernelGGL(k_write, dim3(2048), dim3(256), 0, s, (float4*)d); break;
      }
    };
    for (int i = 0; i < 100; ++i) launch(i);
    CK(hipStreamSynchronize(s));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; ++i) launch(i);
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    float ms;
    CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("graph %-34s: %.2f us per kernel\n", names[mode], ms * 1e3 / N);
  }
  return 0;
}
