// VALU issue-rate probes for the attention softmax on gfx950: cycles per wave64 instruction
// for v_exp_f32, v_fma_f32, v_pk_fma_f32, v_max3_f32, v_cvt_pk_f16_f32, v_pk_mul_f32, alone and
// with MFMAs of the same wave in flight (does the matrix pipe overlap the VALU of one wave?).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float float2v __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// OP: 0 exp, 1 fma, 2 pk_fma, 3 max3, 4 cvt_pk, 5 pk_mul, 6 add ; MF: MFMAs interleaved per 32 ops
template <int OP, int MF>
__global__ void k_valu(long long* out, float* sink, int iters) {
  float v[32];
  for (int i = 0; i < 32; ++i) v[i] = threadIdx.x * 1e-3f + i * 0.01f;
  half8 a, b;
  for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x * 0.001f + e); b[e] = (_Float16)(e * 0.01f); }
  floatx16 acc[4];
  for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) acc[n][r] = 0.f;
  const float s = 0.999f, m = 1e-4f;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int n = 0; n < MF; ++n)
      asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[n & 3]) : "v"(a), "v"(b));
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      if (OP == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
      if (OP == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(s), "v"(m));
      if (OP == 3) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(s), "v"(m));
      if (OP == 6) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(m));
    }
    if (OP == 2 || OP == 5) {
#pragma unroll
      for (int i = 0; i < 32; i += 2) {
        float2v p = {v[i], v[i + 1]}, ss = {s, s}, mm = {m, m};
        if (OP == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(ss), "v"(mm));
        if (OP == 5) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p) : "v"(ss));
        v[i] = p[0]; v[i + 1] = p[1];
      }
    }
    if (OP == 4) {
#pragma unroll
      for (int i = 0; i < 32; i += 2) {
        unsigned r;
        asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r) : "v"(v[i]), "v"(v[i + 1]));
        v[i] = __uint_as_float(r);
      }
    }
  }
  long long t1 = clock64();
  float x = 0.f;
  for (int i = 0; i < 32; ++i) x += v[i];
  for (int n = 0; n < 4; ++n) for (int r = 0; r < 16; ++r) x += acc[n][r];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = x;
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int OP, int MF>
int run(const char* name, int nops, long long* d, float* sink) {
  long long h;
  const int it = 2000;
  for (int threads : {256, 512}) {
    hipLaunchKernelGGL((k_valu<OP, MF>), dim3(256), dim3(threads), 0, 0, d, sink, it);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost));
    printf("%-12s + %d mfma, %d waves/SIMD: %7.1f cycles per iteration (%d VALU ops -> %.2f cycles/op if VALU-bound; MFMA alone %d)\n",
           name, MF, threads / 256, (double)h / it, nops, (double)h / it / nops, MF * 32);
  }
  return 0;
}

int main() {
  long long* d; float* sink;
  CK(hipMalloc(&d, 64)); CK(hipMalloc(&sink, 1 << 22));
  run<0, 0>("v_exp_f32", 32, d, sink);
  run<1, 0>("v_fma_f32", 32, d, sink);
  run<2, 0>("v_pk_fma_f32", 16, d, sink);
  run<3, 0>("v_max3_f32", 32, d, sink);
  run<4, 0>("v_cvt_pk_f16", 16, d, sink);
  run<5, 0>("v_pk_mul_f32", 16, d, sink);
  run<6, 0>("v_add_f32", 32, d, sink);
  run<0, 4>("v_exp_f32", 32, d, sink);
  run<1, 4>("v_fma_f32", 32, d, sink);
  run<1, 8>("v_fma_f32", 32, d, sink);
  run<6, 2>("v_add_f32", 32, d, sink);
  run<0, 16>("v_exp_f32", 32, d, sink);
  return 0;
}
