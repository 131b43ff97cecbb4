// HBM/L2 -> LDS fill-rate probe for gfx950: what does one CU sustain with global_load_lds_dwordx4
// (LDS-DMA) vs global_load_dwordx4 + ds_write_b128, as a function of loads in flight per wave and
// workgroups per CU?  Access pattern = the implicit-GEMM A tile: 8 lanes cover one 128-B row
// segment, rows 640 B apart (C = 320 fp16 channels), every workgroup walks its own 128-row window.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float float4v __attribute__((ext_vector_type(4)));

// MODE 0: LDS-DMA, wait all + barrier per batch (2-stage GEMM loop shape)
// MODE 1: LDS-DMA, keep D in flight (wait for the older half only), barrier per batch
// MODE 2: register loads + ds_write_b128, wait all + barrier per batch
template <int MODE, int D>
__global__ __launch_bounds__(256) void k_fill(const char* __restrict__ src, size_t window, int rowstride, int iters,
                                              long long* out, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = src + (size_t)blockIdx.x * window;
  // one "load" = 64 lanes x 16 B = 8 rows x 128 B
  const size_t lane_off = (size_t)(lane >> 3) * rowstride + (lane & 7) * 16;
  float4v acc = {0, 0, 0, 0};
  long long t0 = clock64();
  size_t pos = 0;
  for (int it = 0; it < iters; ++it) {
    char* stage = lds + (it & 1) * (D * 4 * 1024);
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const size_t row0 = (size_t)((wave * D + i) * 8) * rowstride;
      const char* p = base + ((pos + row0) % window) + lane_off;
      if (MODE == 2) {
        const float4v v = *reinterpret_cast<const float4v*>(p);
        *reinterpret_cast<float4v*>(stage + (wave * D + i) * 1024 + lane * 16) = v;
      } else {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)p,
                                         (__attribute__((address_space(3))) void*)(stage + (wave * D + i) * 1024), 16, 0, 0);
      }
    }
    pos += 128;   // next 64-channel K slice of the same rows
    if (MODE == 3) {
      // handled below
    } else if (MODE == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    acc[0] += *reinterpret_cast<float*>(lds + ((it * 64 + tid * 4) & 4095));
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  long long t1 = clock64();
  sink[blockIdx.x * 256 + tid] = acc[0];
  if (tid == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

// MODE 3: plain global_load_dwordx4 into VGPRs, 2 batches of D in flight, no LDS at all (TA -> VGPR rate)
// MODE 4: as 3 but every load instruction covers 1 KB contiguous (no row gather)
template <int MODE, int D>
__global__ __launch_bounds__(256) void k_regs(const char* __restrict__ src, size_t window, int rowstride, int iters,
                                              long long* out, float* sink) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const char* base = src + (size_t)blockIdx.x * window;
  const size_t lane_off = MODE == 4 ? (size_t)lane * 16 : (size_t)(lane >> 3) * rowstride + (lane & 7) * 16;
  float4v acc = {0, 0, 0, 0};
  float4v buf[2][D];
  long long t0 = clock64();
  size_t pos = 0;
  auto issue = [&](int b) {
#pragma unroll
    for (int i = 0; i < D; ++i) {
      const size_t row0 = MODE == 4 ? (size_t)(wave * D + i) * 1024 : (size_t)((wave * D + i) * 8) * rowstride;
      buf[b][i] = *reinterpret_cast<const float4v*>(base + ((pos + row0) % window) + lane_off);
    }
    pos += 128;
  };
  issue(0);
  for (int it = 0; it < iters; it += 2) {
    issue(1);
#pragma unroll
    for (int i = 0; i < D; ++i) acc += buf[0][i];
    issue(0);
#pragma unroll
    for (int i = 0; i < D; ++i) acc += buf[1][i];
  }
  long long t1 = clock64();
  sink[blockIdx.x * 256 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
  if (tid == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int MODE, int D>
int run(const char* src, size_t bytes, long long* d, float* sink) {
  const int iters = 400;
  for (int blocks : {256, 512, 1024}) {
    const size_t window = 128 * 640;   // 128 rows x 640 B: reused -> L2 resident, like a GEMM A panel
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    if constexpr (MODE >= 3) {
      auto k = k_regs<MODE, D>;
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, src, window, 640, 10, d, sink);
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, src, window, 640, iters, d, sink);
    } else {
    auto k = k_fill<MODE, D>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * D * 4096));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 2 * D * 4096, 0, src, window, 640, 10, d, sink);
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 2 * D * 4096, 0, src, window, 640, iters, d, sink);
    }
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    long long h; CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    const double total = (double)blocks * iters * D * 4 * 1024;
    printf("mode %d D=%2d (%3d KB/batch/WG) WGs %4d: %7.1f cycles/batch, %6.2f TB/s chip, %6.1f GB/s per CU\n", MODE, D,
           D * 4, blocks, (double)h / iters, total / ms / 1e9, total / ms / 1e6 / 256);
  }
  return 0;
}

int main() {
  char* src; long long* d; float* sink;
  const size_t bytes = (size_t)1024 * 128 * 640 + (1 << 20);
  CK(hipMalloc(&src, bytes)); CK(hipMemset(src, 1, bytes));
  CK(hipMalloc(&d, 64)); CK(hipMalloc(&sink, 1 << 22));
  run<0, 8>(src, bytes, d, sink);
  run<1, 8>(src, bytes, d, sink);
  run<2, 8>(src, bytes, d, sink);
  run<3, 4>(src, bytes, d, sink);
  run<3, 8>(src, bytes, d, sink);
  run<4, 8>(src, bytes, d, sink);
  return 0;
}
