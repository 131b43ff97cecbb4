// CLIP text-encoder kernels that the UNet path does not already have: token + position embedding
// gather, causal multi-head self-attention over the 77-token prompt, the MLP activations.
//
// Math (reference): the text encoders are transformers' CLIPTextModel / CLIPTextModelWithProjection,
// wrapped by python_coreml_stable_diffusion/torch2coreml.py:379-441 with the causal mask patched to
// -1e4 instead of -inf (torch2coreml.py:363-377) and called from pipeline.py:151-175.  The GEMMs
// (q/k/v/out projections, fc1, fc2) and LayerNorms run on igemm.hip / norm.hip.
#include "kernels.h"

namespace sd {
namespace {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// x[s][:] = token_embedding[ids[s]][:] + position_embedding[s][:]   (CLIPTextEmbeddings.forward)
__global__ void clip_embed_kernel(const int* __restrict__ ids, const half_t* __restrict__ tok,
                                  const half_t* __restrict__ pos, half_t* __restrict__ x, int S, int D, int vocab) {
  const int s = blockIdx.x;
  int id = ids[s];
  id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
  const half_t* t = tok + (size_t)id * D;
  const half_t* p = pos + (size_t)s * D;
  for (int c = threadIdx.x * 8; c < D; c += blockDim.x * 8) {
    const half8 a = *reinterpret_cast<const half8*>(t + c), b = *reinterpret_cast<const half8*>(p + c);
    half8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)((float)a[e] + (float)b[e]);
    *reinterpret_cast<half8*>(x + (size_t)s * D + c) = o;
  }
}

// Causal self-attention of one head per workgroup (S <= 80 tokens, d <= 128): K and V of the head sit
// in LDS, each wavefront owns query rows (wave, wave + 4, ...).  softmax(q.k * d^-0.5 + mask) with the
// reference's additive mask -1e4 on keys j > i: exp(s - 1e4 - max) underflows to exactly 0 in fp32, so
// the masked keys are simply skipped.  qkv rows are [q | k | v] of width 3*D (one stacked projection).
constexpr int CLIP_SMAX = 80, CLIP_DMAX = 128;   // 77-token prompts; 40 KB of K|V per head
__global__ __launch_bounds__(256) void clip_attention_kernel(const half_t* __restrict__ qkv, half_t* __restrict__ out,
                                                             int S, int D, int d, float scale) {
  __shared__ __attribute__((aligned(16))) half_t Ks[CLIP_SMAX * CLIP_DMAX];
  __shared__ __attribute__((aligned(16))) half_t Vs[CLIP_SMAX * CLIP_DMAX];
  __shared__ float Ps[4][CLIP_SMAX];
  __shared__ float Qs[4][CLIP_DMAX];
  const int h = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int ld = 3 * D;
  const int dv = d >> 3;                          // 16-B chunks per row
  for (int i = t; i < S * dv; i += 256) {
    const int s = i / dv, c = (i - s * dv) * 8;
    *reinterpret_cast<half8*>(Ks + s * d + c) = *reinterpret_cast<const half8*>(qkv + (size_t)s * ld + D + h * d + c);
    *reinterpret_cast<half8*>(Vs + s * d + c) = *reinterpret_cast<const half8*>(qkv + (size_t)s * ld + 2 * D + h * d + c);
  }
  __syncthreads();
  for (int i = wave; i < S; i += 4) {
    for (int c = lane; c < d; c += 64) Qs[wave][c] = (float)qkv[(size_t)i * ld + h * d + c] * scale;
    __builtin_amdgcn_wave_barrier();
    float sc[2];
    float mx = -1e30f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int j = lane + 64 * r;
      float acc = -1e30f;
      if (j <= i) {
        acc = 0.f;
        for (int c = 0; c < d; ++c) acc += Qs[wave][c] * (float)Ks[j * d + c];
      }
      sc[r] = acc;
      mx = fmaxf(mx, acc);
    }
    mx = wave_max_f(mx);
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int j = lane + 64 * r;
      const float p = (j <= i) ? __expf(sc[r] - mx) : 0.f;
      if (j < CLIP_SMAX) Ps[wave][j] = p;
      sum += p;
    }
    const float inv = 1.0f / wave_sum_f(sum);
    __builtin_amdgcn_wave_barrier();
    for (int c = lane; c < d; c += 64) {
      float acc = 0.f;
      for (int j = 0; j <= i; ++j) acc += Ps[wave][j] * (float)Vs[j * d + c];
      out[(size_t)i * D + h * d + c] = (half_t)(acc * inv);
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// in-place MLP activation: 0 = quick_gelu x*sigmoid(1.702x) (CLIP ViT-L), 1 = exact-erf gelu (OpenCLIP)
__global__ void clip_act_kernel(half_t* __restrict__ x, size_t n8, int act) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n8; i += (size_t)gridDim.x * blockDim.x) {
    half8 v = reinterpret_cast<half8*>(x)[i];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float f = (float)v[e];
      v[e] = (half_t)(act == 0 ? f / (1.0f + __expf(-1.702f * f)) : 0.5f * f * (1.0f + erff(f * 0.70710678118654752f)));
    }
    reinterpret_cast<half8*>(x)[i] = v;
  }
}

}  // namespace

void launch_clip_embed(const int* ids, const half_t* tok, const half_t* pos, half_t* x, int S, int D, int vocab,
                       hipStream_t s) {
  SD_REQUIRE(D % 8 == 0, kUnsupported, "clip_embed: hidden size %d", D);
  hipLaunchKernelGGL(clip_embed_kernel, dim3(S), dim3(128), 0, s, ids, tok, pos, x, S, D, vocab);
  SD_HIP(hipGetLastError());
}

void launch_clip_attention(const half_t* qkv, half_t* out, int S, int D, int heads, hipStream_t s) {
  const int d = D / heads;
  SD_REQUIRE(S <= CLIP_SMAX && d <= CLIP_DMAX && d % 8 == 0 && heads * d == D, kUnsupported,
             "clip_attention: S=%d D=%d heads=%d", S, D, heads);
  hipLaunchKernelGGL(clip_attention_kernel, dim3(heads), dim3(256), 0, s, qkv, out, S, D, d, 1.0f / sqrtf((float)d));
  SD_HIP(hipGetLastError());
}

void launch_clip_act(half_t* x, size_t n, int act, hipStream_t s) {
  SD_REQUIRE(n % 8 == 0 && (act == 0 || act == 1), kInvalidArgument, "clip_act: n=%zu act=%d", n, act);
  const size_t n8 = n / 8;
  hipLaunchKernelGGL(clip_act_kernel, dim3((int)std::min<size_t>((n8 + 255) / 256, 2048)), dim3(256), 0, s, x, n8, act);
  SD_HIP(hipGetLastError());
}

}  // namespace sd
