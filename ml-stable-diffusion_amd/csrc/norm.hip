// K6/K7 - normalisation kernels (HBM-bound; VALU + wavefront-shuffle reductions, fp32 statistics).
//
//   layernorm:  LayerNormANE over the channel dim of BC1S (layer_norm.py:51-80; with the load hook
//               unet.py:132-138 the affine is x_hat*w + b on the checkpoint tensors), eps 1e-5,
//               biased variance computed as mean((x-mu)^2) exactly like the reference.
//               One wavefront per token row: the row lives in registers (16-B loads), two
//               shuffle-reductions, one 16-B store per lane-chunk.
//   groupnorm:  torch.nn.GroupNorm(32) of unet.py:430-451 (eps = norm_eps 1e-5), :528-531
//               (SpatialTransformer, eps hard-coded 1e-6) and :966-968.  Channels-last makes a group
//               a strided set, so statistics are a column reduction: each thread owns one 8-channel
//               chunk column over a slab of pixels; partial (sum, sumsq) are merged per group in LDS
//               in a fixed order and written per (sample, slab, group) - no atomics, so results are
//               bitwise reproducible.  The apply pass folds the slabs in its prologue and fuses the
//               affine, the optional SiLU (unet.py:473,481) and the channel concat of the up-block
//               inputs (unet.py:213-216) into a single read-modify-write.
#include <cstdlib>

#include "kernels.h"

namespace sd {
namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// partial layout [B][G][kGnMaxSlabs][2] (kernels.h); entries >= the producer's count are ignored by the fold
constexpr int LN_MAX_CHUNKS = 4;   // C <= 64 lanes * 4 chunks * 8 = 2048

__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ b, half_t* __restrict__ y, int M,
                                                        int C, float eps) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int nchunk = C >> 3;
  const half_t* xr = x + (size_t)row * C;
  float v[LN_MAX_CHUNKS][8];
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nchunk) {
      const half8 h = *reinterpret_cast<const half8*>(xr + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] = (float)h[e];
        sum += v[i][e];
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[i][e] = 0.f;
    }
  }
  const float mean = wave_sum(sum) / (float)C;
  float sq = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float dlt = v[i][e] - mean;
        sq += dlt * dlt;
      }
    }
  }
  const float rstd = rsqrtf(wave_sum(sq) / (float)C + eps);
  half_t* yr = y + (size_t)row * C;
#pragma unroll
  for (int i = 0; i < LN_MAX_CHUNKS; ++i) {
    const int ch = lane + 64 * i;
    if (ch < nchunk) {
      half8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (half_t)((v[i][e] - mean) * rstd * w[ch * 8 + e] + b[ch * 8 + e]);
      *reinterpret_cast<half8*>(yr + ch * 8) = o;
    }
  }
}

// Pass 1, grid (pixel slabs, B): deterministic partial (sum, sumsq) per (sample, slab, group).
// No atomics anywhere: bitwise run-to-run reproducibility of the whole UNet depends on it.
__global__ __launch_bounds__(256) void groupnorm_partial_kernel(const half_t* __restrict__ x0, int C0,
                                                                const half_t* __restrict__ x1, int C1,
                                                                float* __restrict__ partial, int HW, int G,
                                                                int pix_per_block) {
  __shared__ float part[256][17];          // per-thread (sum[8], sumsq[8]); +1 pad against bank conflicts
  __shared__ float chs[2048], chq[2048];   // per-channel sums of the current column block
  const int C = C0 + C1;
  const int ncol = C >> 3;                 // 8-channel chunk columns
  const int cpg = C / G;
  const int b = blockIdx.y;
  const int p0 = blockIdx.x * pix_per_block;
  const int p1 = min(p0 + pix_per_block, HW);
  const int t = threadIdx.x;
  float gs = 0.f, gq = 0.f;                // thread g < G accumulates group g across column blocks
  for (int cb = 0; cb < ncol; cb += 256) { // C <= 2048 -> one pass; C = 2560 -> two
    const int cols = min(256, ncol - cb);
    const int rows = 256 / cols;           // pixel rows in flight
    const int col = cb + t % cols;
    const int r0 = t / cols;
    float s[8], q[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
    if (r0 < rows) {
      const int c = col * 8;
      const half_t* src = (c < C0) ? x0 + c : x1 + (c - C0);
      const int Cs = (c < C0) ? C0 : C1;
      // 4 independent 16-B loads in flight per thread (a dependent one-load-per-iteration loop
      // would expose the full memory latency every iteration)
      for (int p = p0 + r0; p < p1; p += 4 * rows) {
        half8 h[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int pp = p + u * rows;
          h[u] = (pp < p1) ? *reinterpret_cast<const half8*>(src + ((size_t)b * HW + pp) * Cs) : half8{0, 0, 0, 0, 0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = (float)h[u][e];
            s[e] += f;
            q[e] += f * f;
          }
      }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      part[t][e] = s[e];
      part[t][8 + e] = q[e];
    }
    __syncthreads();
    for (int cc = t; cc < cols * 8; cc += 256) {   // fixed-order reduction over the rows in flight
      const int cl = cc >> 3, e = cc & 7;
      float a = 0.f, d = 0.f;
      for (int r = 0; r < rows; ++r) {
        a += part[r * cols + cl][e];
        d += part[r * cols + cl][8 + e];
      }
      chs[cc] = a;
      chq[cc] = d;
    }
    __syncthreads();
    if (t < G) {                                   // channels of group t inside this column block
      const int lo = max(t * cpg, cb * 8), hi = min((t + 1) * cpg, (cb + cols) * 8);
      for (int c = lo; c < hi; ++c) {
        gs += chs[c - cb * 8];
        gq += chq[c - cb * 8];
      }
    }
    __syncthreads();
  }
  if (t < G) {
    float* dst = partial + (((size_t)b * G + t) * kGnMaxSlabs + blockIdx.x) * 2;   // [b][g][slab][2]
    dst[0] = gs;
    dst[1] = gq;
  }
}

// Pass 2 / single-launch GroupNorm: the bodies live in gn_body.inc (shared with igemm.hip's side-by-side launches)
#include "gn_body.inc"

__global__ __launch_bounds__(256) void groupnorm_apply_kernel(const half_t* __restrict__ x0, int C0,
                                                              const half_t* __restrict__ x1, int C1,
                                                              const float* __restrict__ partial, int slabs,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, half_t* __restrict__ y,
                                                              int HW, int G, float eps, int silu, int pix_per_block) {
  groupnorm_apply_body(x0, C0, x1, C1, partial, slabs, gamma, beta, y, HW, G, eps, silu, pix_per_block, blockIdx.x, blockIdx.y);
}

template <int VW, int NT = 256, bool RES = false>
__global__ __launch_bounds__(NT) void groupnorm_fused_kernel(const half_t* __restrict__ x0, int C0,
                                                              const half_t* __restrict__ x1, int C1,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, half_t* __restrict__ y,
                                                              int HW, int G, float eps, int silu) {
  groupnorm_fused_body<VW, NT, RES>(x0, C0, x1, C1, gamma, beta, y, HW, G, eps, silu, blockIdx.x, blockIdx.y);
}

// One workgroup per row: row kept in registers (cols <= 256 threads * 4 chunks * 8), fp32 max / sum
// with wave shuffles + LDS across the four waves.
__global__ __launch_bounds__(256) void row_softmax_kernel(half_t* __restrict__ x, int cols, float scale_log2) {
  __shared__ float red[8];
  half_t* row = x + (size_t)blockIdx.x * cols;
  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int nchunk = cols >> 3;
  float v[4][8];
  float mx = -1e30f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ch = t + 256 * i;
    if (ch < nchunk) {
      const half8 h = *reinterpret_cast<const half8*>(row + ch * 8);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] = (float)h[e] * scale_log2;
        mx = fmaxf(mx, v[i][e]);
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  if (lane == 0) red[wave] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ch = t + 256 * i;
    if (ch < nchunk) {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v[i][e] = __builtin_amdgcn_exp2f(v[i][e] - mx);
        sum += v[i][e];
      }
    }
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wave] = sum;
  __syncthreads();
  const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ch = t + 256 * i;
    if (ch < nchunk) {
      half8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = (half_t)(v[i][e] * inv);
      *reinterpret_cast<half8*>(row + ch * 8) = o;
    }
  }
}

}  // namespace

void launch_row_softmax(half_t* x, int rows, int cols, float scale, hipStream_t s) {
  SD_REQUIRE(cols % 8 == 0 && cols <= 256 * 4 * 8, kUnsupported, "row_softmax: cols=%d (need cols %% 8 == 0, <= 8192)", cols);
  hipLaunchKernelGGL(row_softmax_kernel, dim3(rows), dim3(256), 0, s, x, cols, scale * 1.4426950408889634f);
  SD_HIP(hipGetLastError());
}

void launch_layernorm(const half_t* x, const float* w, const float* b, half_t* y, int M, int C, float eps,
                      hipStream_t s) {
  SD_REQUIRE(C % 8 == 0 && C <= 64 * 8 * LN_MAX_CHUNKS, kUnsupported, "layernorm: C=%d (need C %% 8 == 0, C <= 2048)", C);
  hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(M, 4)), dim3(256), 0, s, x, w, b, y, M, C, eps);
  SD_HIP(hipGetLastError());
}

int groupnorm_num_slabs(int B, int HW) {
  // ~256+ blocks in total keeps every CU busy; at least 16 pixels per block; <= 128 slabs
  int want = std::max(1, 512 / std::max(1, B));
  return std::max(1, std::min(std::min(want, 128), std::max(1, HW / 16)));
}

namespace {
long gn_fused_max_hw() {
  // measured crossover (tools/gn_sweep.py): the single launch wins for HW <= 256 (3.5-11 us vs 10-23 us), the
  // slab pair wins at 32x32 and above (8-16 us vs 13-208 us).  SD_GN_FUSED_MAX_HW overrides for tuning.
  static const long v = tune_env_int("SD_GN_FUSED_MAX_HW", 256);
  return v;
}
}  // namespace

bool groupnorm_wants_producer_stats(int HW, int C, int G) {
  static const bool off = tune_env_set("SD_NO_GN_PRODUCER_STATS");   // A/B switch
  if (off || G < 1 || C % G != 0) return false;
  const int cpg = C / G;
  // (round 5: also where the GroupNorm itself would be ONE launch, HW <= 256: its 64-workgroup two-pass kernel takes 7-13 us in
  // sequence, the fully parallel apply pass behind producer statistics ~5; SD_GN_STATS_SMALL=0 restores round 4's rule)
  static const bool small_too = tune_env_int("SD_GN_STATS_SMALL", 0) != 0;   // measured: no gain (LAB_NOTES.md r5)
  const bool single_launch = HW <= gn_fused_max_hw() && cpg <= 128 && cpg % 2 == 0;
  return (!single_launch || small_too) && cpg <= 64 && G <= 64;
}

size_t groupnorm_scratch_floats(int B, int HW, int G) { return (size_t)B * G * kGnMaxSlabs * 2; }

void launch_groupnorm(const half_t* x0, int C0, const half_t* x1, int C1, float* partial, const float* gamma,
                      const float* beta, half_t* y, int B, int HW, int G, float eps, int silu, hipStream_t s,
                      int producer_entries, const ConvDesc* side) {
  if (!x1) C1 = 0;
  static const bool res = tune_env_int("SD_GN_RESIDENT", 0) != 0;   // 1: round 5's register-resident single-launch form (A/B; gn_body.inc)
  const int silu_f = silu ? 1 : 0;
  const int C = C0 + C1;
  SD_REQUIRE(C % G == 0 && C0 % 8 == 0 && C1 % 8 == 0 && G <= 64, kUnsupported, "groupnorm: C0=%d C1=%d G=%d", C0, C1, G);
  const int cpg = C / G;
  if (producer_entries > 0) {   // statistics came out of the producing kernel's epilogue: one apply launch
    SD_REQUIRE(producer_entries <= kGnMaxSlabs && !x1, kInternal, "groupnorm: %d producer entries", producer_entries);
    // pixel slabs of the apply pass are independent of the producer's entries: ~512 workgroups, >= 8 pixels each
    int slabs = std::max(1, std::min(std::max(1, 512 / std::max(1, B)), std::max(1, HW / 8)));
    const int ppb = cdiv(HW, slabs);
    slabs = cdiv(HW, ppb);
    if (side) {   // the apply pass and the independent GEMM in one grid
      launch_gn_apply_side(x0, C0, x1, C1, partial, producer_entries, gamma, beta, y, B, HW, G, eps, silu, slabs, ppb, *side, s);
      return;
    }
    hipLaunchKernelGGL(groupnorm_apply_kernel, dim3(slabs, B), dim3(256), 0, s, x0, C0, x1, C1, partial, producer_entries, gamma,
                       beta, y, HW, G, eps, silu, ppb);
    SD_HIP(hipGetLastError());
    return;
  }
  const long fused_max_hw = gn_fused_max_hw();
  if (HW <= fused_max_hw && cpg <= 128 && cpg % 2 == 0) {
    dim3 grid(G, B);
    if (side) {   // the single-launch GroupNorm (64 workgroups) and the independent GEMM in one grid
      launch_gn_fused_side(cpg % 8 == 0 ? 8 : (cpg % 4 == 0 ? 4 : 2), x0, C0, x1, C1, gamma, beta, y, B, HW, G, eps, silu_f, *side, s);
      return;
    }
#define SD_GN_FUSED(VW_, NT_)                                                                                                              \
  do {                                                                                                                                     \
    if (res) hipLaunchKernelGGL((groupnorm_fused_kernel<VW_, NT_, true>), grid, dim3(NT_), 0, s, x0, C0, x1, C1, gamma, beta, y, HW, G, eps, silu_f); \
    else hipLaunchKernelGGL((groupnorm_fused_kernel<VW_, NT_, false>), grid, dim3(NT_), 0, s, x0, C0, x1, C1, gamma, beta, y, HW, G, eps, silu_f);    \
  } while (0)
    if (cpg % 8 == 0) SD_GN_FUSED(8, 256);
    else if (cpg % 4 == 0) SD_GN_FUSED(4, 256);
    else SD_GN_FUSED(2, 256);
    SD_HIP(hipGetLastError());
    return;
  }
  // 32x32 level: one 1024-thread workgroup per (group, sample) instead of the slab pair (in sequence a dependent launch costs
  // more than the second pass over a slice that is still in L2); SD_GN_WIDE=0 switches it off (A/B)
  static const bool wide = tune_env_int("SD_GN_WIDE", 1) != 0;
  if (wide && !side && HW <= 1024 && cpg >= 16 && cpg <= 48 && cpg % 4 == 0) {   // (1024-thread blocks: no side GEMM there)   // (60-channel groups measured slower: 20.8 vs 16.0 us)
    dim3 grid(G, B);
    if (cpg % 8 == 0) SD_GN_FUSED(8, 1024);
    else SD_GN_FUSED(4, 1024);
    SD_HIP(hipGetLastError());
    return;
  }
  int slabs = groupnorm_num_slabs(B, HW);
  const int ppb = cdiv(HW, slabs);
  slabs = cdiv(HW, ppb);   // <= groupnorm_num_slabs
  hipLaunchKernelGGL(groupnorm_partial_kernel, dim3(slabs, B), dim3(256), 0, s, x0, C0, x1, C1, partial, HW, G, ppb);
  if (side) {
    launch_gn_apply_side(x0, C0, x1, C1, partial, slabs, gamma, beta, y, B, HW, G, eps, silu, slabs, ppb, *side, s);
    return;
  }
  hipLaunchKernelGGL(groupnorm_apply_kernel, dim3(slabs, B), dim3(256), 0, s, x0, C0, x1, C1, partial, slabs, gamma, beta,
                     y, HW, G, eps, silu, ppb);
  SD_HIP(hipGetLastError());
}

}  // namespace sd
