// Micro-benchmarks calibrating the cost model used for kernel design on gfx950:
// s_barrier round trip, __syncthreads, MFMA issue rate (1 and 2 waves per SIMD), ds_read_b128.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void k_barrier(long long* out, int iters) {
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) __builtin_amdgcn_s_barrier();
    else if (MODE == 1) __syncthreads();
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  long long t1 = clock64();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int NACC>
__global__ void k_mfma(long long* out, float* sink, int iters) {
  half8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.01f); }
  floatx16 acc[NACC];
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int n = 0; n < NACC; ++n) acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[n], 0, 0, 0);
  }
  long long t1 = clock64();
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int r = 0; r < 16; ++r) s += acc[n][r];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

__global__ void k_ldsread(long long* out, float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[32768];
  for (int i = threadIdx.x; i < 32768; i += blockDim.x) lds[i] = (_Float16)i;
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const _Float16* p = lds + (lane & 31) * 64 + ((((lane >> 5)) ^ (((lane & 31) >> 1) & 7)) * 8);
  half8 v[16];
  float s = 0.f;
  long long t0 = clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) v[k] = *reinterpret_cast<const half8*>(p + (k & 3) * 2048 + (k >> 2) * 8192);
#pragma unroll
    for (int k = 0; k < 16; ++k) asm volatile("" ::"v"(v[k]));
  }
  long long t1 = clock64();
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

int main() {
  long long* d; float* sink; long long h;
  CK(hipMalloc(&d, 64)); CK(hipMalloc(&sink, 1 << 22));
  const int it = 2000;
  for (int threads : {64, 256, 512, 1024}) {
    for (int mode = 0; mode < 3; ++mode) {
      for (int blocks : {1, 256, 1024}) {
        if (mode == 0) hipLaunchKernelGGL(k_barrier<0>, dim3(blocks), dim3(threads), 0, 0, d, it);
        if (mode == 1) hipLaunchKernelGGL(k_barrier<1>, dim3(blocks), dim3(threads), 0, 0, d, it);
        if (mode == 2) hipLaunchKernelGGL(k_barrier<2>, dim3(blocks), dim3(threads), 0, 0, d, it);
        CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
        printf("barrier mode %d threads %4d blocks %4d: %.1f cycles/iter\n", mode, threads, blocks, (double)h / it);
      }
    }
  }
  for (int threads : {256, 512}) for (int blocks : {1, 256, 512}) {
    hipLaunchKernelGGL(k_mfma<4>, dim3(blocks), dim3(threads), 0, 0, d, sink, it);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    printf("mfma 4 acc threads %d blocks %d: %.1f cycles per MFMA (per wave)\n", threads, blocks, (double)h / it / 4);
    hipLaunchKernelGGL(k_mfma<1>, dim3(blocks), dim3(threads), 0, 0, d, sink, it);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    printf("mfma 1 acc (dependent) threads %d blocks %d: %.1f cycles per MFMA\n", threads, blocks, (double)h / it);
  }
  for (int threads : {64, 256, 512}) for (int blocks : {1, 256}) {
    hipLaunchKernelGGL(k_ldsread, dim3(blocks), dim3(threads), 0, 0, d, sink, it);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    printf("16 x ds_read_b128 threads %d blocks %d: %.1f cycles per 16 reads\n", threads, blocks, (double)h / it);
  }
  return 0;
}
