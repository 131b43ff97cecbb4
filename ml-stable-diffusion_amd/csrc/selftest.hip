// Hardware self-check: verifies on the running GPU the two lane-layout facts every kernel in
// this library is built on (cdna_hip_programming.md section 3):
//   bit 0  v_mfma_f32_32x32x16_f16 operand/accumulator lane mapping
//          A: lane l holds A[i=l&31][k=8*(l>>5)+e], B: B[k=8*(l>>5)+e][j=l&31],
//          D: lane l reg r holds D[i=(r&3)+8*(r>>2)+4*(l>>5)][j=l&31]
//   bit 1  the DPP butterfly (quad_perm xor1, xor2, row_half_mirror, row_mirror) + xor-16
//          ds_bpermute used for the ORIGINAL-attention row reductions
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

__global__ void dpp_probe(const float* __restrict__ in, float* __restrict__ out_max, float* __restrict__ out_sum) {
  const int l = threadIdx.x;
  float v = in[l];
  float m = v;
  m = fmaxf(m, dpp_mov<0xB1>(m));
  m = fmaxf(m, dpp_mov<0x4E>(m));
  m = fmaxf(m, dpp_mov<0x141>(m));
  m = fmaxf(m, dpp_mov<0x140>(m));
  m = fmaxf(m, __shfl_xor(m, 16));
  float s = v;
  s += dpp_mov<0xB1>(s);
  s += dpp_mov<0x4E>(s);
  s += dpp_mov<0x141>(s);
  s += dpp_mov<0x140>(s);
  s += __shfl_xor(s, 16);
  out_max[l] = m;
  out_sum[l] = s;
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
  float *dD, *dIn, *dMax, *dSum;
  SD_HIP(hipMalloc(&dA, sizeof(hA)));
  SD_HIP(hipMalloc(&dB, sizeof(hB)));
  SD_HIP(hipMalloc(&dD, sizeof(got)));
  SD_HIP(hipMalloc(&dIn, 64 * 4));
  SD_HIP(hipMalloc(&dMax, 64 * 4));
  SD_HIP(hipMalloc(&dSum, 64 * 4));
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

  float hin[64], hmax[64], hsum[64];
  for (int l = 0; l < 64; ++l) hin[l] = (float)((l * 37) % 101) - 50.f;
  SD_HIP(hipMemcpy(dIn, hin, sizeof(hin), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(dpp_probe, dim3(1), dim3(64), 0, 0, dIn, dMax, dSum);
  SD_HIP(hipDeviceSynchronize());
  SD_HIP(hipMemcpy(hmax, dMax, sizeof(hmax), hipMemcpyDeviceToHost));
  SD_HIP(hipMemcpy(hsum, dSum, sizeof(hsum), hipMemcpyDeviceToHost));
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
  (void)hipFree(dA);
  (void)hipFree(dB);
  (void)hipFree(dD);
  (void)hipFree(dIn);
  (void)hipFree(dMax);
  (void)hipFree(dSum);
  return result;
}

}  // namespace sd
