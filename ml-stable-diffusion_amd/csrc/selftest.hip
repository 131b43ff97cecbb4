// Hardware self-check: verifies on the running GPU the two lane-layout facts every kernel in
// this library is built on (cdna_hip_programming.md section 3):
//   bit 0  v_mfma_f32_32x32x16_f16 operand/accumulator lane mapping
//          A: lane l holds A[i=l&31][k=8*(l>>5)+e], B: B[k=8*(l>>5)+e][j=l&31],
//          D: lane l reg r holds D[i=(r&3)+8*(r>>2)+4*(l>>5)][j=l&31]
//   bit 1  the DPP butterfly (quad_perm xor1, xor2, row_half_mirror, row_mirror) + the
//          v_permlane16_swap xor-16 step used for the ORIGINAL-attention row reductions
//   bit 2  the v_permlane32_swap xor-32 exchange of the SPLIT_EINSUM softmax
// Asymmetric integer data so a transposed or permuted mapping cannot pass by accident.
#include "kernels.h"

namespace sd {
namespace {

__global__ void mfma_probe(const half_t* __restrict__ A, const half_t* __restrict__ B, float* __restrict__ D) {
  const int l = threadIdx.x;
  half8 a, b;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int k = 8 * (l >> 5) + e;
    a[e] = A[(l & 31) * 16 + k];     // A[32][16] row-major
    b[e] = B[k * 32 + (l & 31)];     // B[16][32] row-major
  }
  floatx16 c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), j = l & 31;
    D[i * 32 + j] = c[r];
  }
}

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}

__device__ __forceinline__ void swap16(float v, float& x, float& y) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];   // scalars first: bit-casting the vector-element lvalue reads lane 0 twice
  x = __uint_as_float(r0);
  y = __uint_as_float(r1);
}
__device__ __forceinline__ void swap32(float v, float& x, float& y) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];   // scalars first: bit-casting the vector-element lvalue reads lane 0 twice
  x = __uint_as_float(r0);
  y = __uint_as_float(r1);
}

// out_x32[l] = in[l] - 2 * in[l ^ 32] would be wrong under any other pairing; the swap returns
// {low-half value, high-half value} in every lane, so own/partner are picked by (l >> 5)
__global__ void dpp_probe(const float* __restrict__ in, float* __restrict__ out_max, float* __restrict__ out_sum,
                          float* __restrict__ out_x32) {
  const int l = threadIdx.x;
  float v = in[l];
  float x, y;
  float m = v;
  m = fmaxf(m, dpp_mov<0xB1>(m));
  m = fmaxf(m, dpp_mov<0x4E>(m));
  m = fmaxf(m, dpp_mov<0x141>(m));
  m = fmaxf(m, dpp_mov<0x140>(m));
  swap16(m, x, y);
  m = fmaxf(x, y);
  float s = v;
  s += dpp_mov<0xB1>(s);
  s += dpp_mov<0x4E>(s);
  s += dpp_mov<0x141>(s);
  s += dpp_mov<0x140>(s);
  swap16(s, x, y);
  s = x + y;
  out_max[l] = m;
  out_sum[l] = s;
  swap32(v, x, y);
  out_x32[l] = x - 2.f * y;      // every lane: in[l & 31] - 2 * in[(l & 31) + 32]
}

}  // namespace

int selftest_mfma() {
  int result = 0;
  half_t hA[32 * 16], hB[16 * 32];
  float ref[32 * 32], got[32 * 32];
  for (int i = 0; i < 32; ++i)
    for (int k = 0; k < 16; ++k) hA[i * 16 + k] = (half_t)(float)((i * 3 + k * 7) % 11 - 5);
  for (int k = 0; k < 16; ++k)
    for (int j = 0; j < 32; ++j) hB[k * 32 + j] = (half_t)(float)((k * 5 + j * 2) % 13 - 6);
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      float s = 0.f;
      for (int k = 0; k < 16; ++k) s += (float)hA[i * 16 + k] * (float)hB[k * 32 + j];
      ref[i * 32 + j] = s;
    }
  half_t *dA, *dB;
  float *dD, *dIn, *dMax, *dSum, *dX32;
  SD_HIP(hipMalloc(&dA, sizeof(hA)));
  SD_HIP(hipMalloc(&dB, sizeof(hB)));
  SD_HIP(hipMalloc(&dD, sizeof(got)));
  SD_HIP(hipMalloc(&dIn, 64 * 4));
  SD_HIP(hipMalloc(&dMax, 64 * 4));
  SD_HIP(hipMalloc(&dSum, 64 * 4));
  SD_HIP(hipMalloc(&dX32, 64 * 4));
  SD_HIP(hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice));
  SD_HIP(hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(mfma_probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  SD_HIP(hipDeviceSynchronize());
  SD_HIP(hipMemcpy(got, dD, sizeof(got), hipMemcpyDeviceToHost));
  int bad = 0;
  for (int i = 0; i < 32 * 32; ++i)
    if (got[i] != ref[i]) {
      if (bad < 8) fprintf(stderr, "[sd selftest] mfma D[%d][%d] = %g, expected %g\n", i / 32, i % 32, got[i], ref[i]);
      ++bad;
    }
  if (bad) result |= 1;

  float hin[64], hmax[64], hsum[64], hx32[64];
  for (int l = 0; l < 64; ++l) hin[l] = (float)((l * 37) % 101) - 50.f;
  SD_HIP(hipMemcpy(dIn, hin, sizeof(hin), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(dpp_probe, dim3(1), dim3(64), 0, 0, dIn, dMax, dSum, dX32);
  SD_HIP(hipDeviceSynchronize());
  SD_HIP(hipMemcpy(hmax, dMax, sizeof(hmax), hipMemcpyDeviceToHost));
  SD_HIP(hipMemcpy(hsum, dSum, sizeof(hsum), hipMemcpyDeviceToHost));
  SD_HIP(hipMemcpy(hx32, dX32, sizeof(hx32), hipMemcpyDeviceToHost));
  bad = 0;
  for (int hf = 0; hf < 2; ++hf) {
    float m = -1e30f, s = 0.f;
    for (int l = 0; l < 32; ++l) {
      m = fmaxf(m, hin[hf * 32 + l]);
      s += hin[hf * 32 + l];
    }
    for (int l = 0; l < 32; ++l)
      if (hmax[hf * 32 + l] != m || hsum[hf * 32 + l] != s) {
        if (bad < 8)
          fprintf(stderr, "[sd selftest] dpp lane %d: max %g (want %g) sum %g (want %g)\n", hf * 32 + l,
                  hmax[hf * 32 + l], m, hsum[hf * 32 + l], s);
        ++bad;
      }
  }
  if (bad) result |= 2;
  bad = 0;
  for (int l = 0; l < 64; ++l) {
    const float want = hin[l & 31] - 2.f * hin[(l & 31) + 32];
    if (hx32[l] != want) {
      if (bad < 8) fprintf(stderr, "[sd selftest] permlane32_swap lane %d: %g (want %g)\n", l, hx32[l], want);
      ++bad;
    }
  }
  if (bad) result |= 4;
  (void)hipFree(dA);
  (void)hipFree(dB);
  (void)hipFree(dD);
  (void)hipFree(dIn);
  (void)hipFree(dMax);
  (void)hipFree(dSum);
  (void)hipFree(dX32);
  return result;
}

}  // namespace sd
