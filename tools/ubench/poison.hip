// Fill every CU's LDS and a wave's worth of VGPRs with NaN bit patterns (what a fresh box may hold): a kernel that reads LDS it never
// wrote - harmless behind a zero weight on benign leftovers - turns the result into NaN afterwards.  Run before a test process:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/poison.hip -o tools/ubench/poison && tools/ubench/poison
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k_poison(unsigned* sink) {
  extern __shared__ unsigned lds[];
  const unsigned nan = 0x7fc00000u ^ (threadIdx.x << 3);   // quiet NaNs (fp32); as two fp16: 0x7fc0 = NaN, low half arbitrary
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += blockDim.x) lds[i] = (i & 1) ? 0x7e007e00u : nan;   // 0x7e00 = fp16 NaN
  __syncthreads();
  unsigned acc = 0;
  for (int i = threadIdx.x; i < 160 * 1024 / 4; i += blockDim.x) acc ^= lds[i];
  if (acc == 0x12345u) sink[0] = acc;
}
int main() {
  unsigned* sink;
  if (hipMalloc(&sink, 4096) != hipSuccess) return 1;
  hipFuncSetAttribute((const void*)k_poison, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(k_poison, dim3(1024), dim3(1024), 160 * 1024, 0, sink);
  hipError_t e = hipDeviceSynchronize();
  printf("poison: %s\n", hipGetErrorString(e));
  return e != hipSuccess;
}
