// fp32 compute path of the VAE graphs (decoder and encoder).
//
// The reference keeps the stock SDXL VAE in float32 - `inputs_dtype = torch.float32`, `compute_precision = FLOAT32`
// (torch2coreml.py:570-578 decoder, :726-733 encoder) - because its activations leave the fp16 range, and the pipeline
// reads the dtype off the model (`expected_inputs['z']['dtype']`, pipeline.py:315).  The MFMA kernels of this library
// store activations in fp16, so a handle built with `compute_fp32` runs its graph on the kernels below instead: fp32
// NHWC activations, fp32 arithmetic and accumulation on the vector ALU (the gfx950 matrix pipe has no fp32-input
// rate advantage: 157 TFLOP/s for both), weights as uploaded (fp16 values, exactly representable).  Parity, not
// speed, is the point of this path: a 512x512 decode costs tens of milliseconds instead of 4.5.
//
//   conv_f32_kernel      implicit GEMM, 64 x 64 output tile per 256-thread workgroup, 4 x 4 outputs per thread, K steps
//                        of 16 staged through LDS (k-major, so the inner loop is two 16-byte LDS reads per 16 FMAs);
//                        3x3 / 1x1, stride 1 / 2, nearest-x2 upsample folded into the gather, the encoder's (0,1,0,1)
//                        padding, bias, residual; "weights" may be fp16 [N][K], fp32 [N][K] (attention: the K tokens)
//                        or fp32 [K][N] (attention: V)
//   groupnorm_f32_*      statistics in fp64 partial sums (deterministic order), then affine (+ SiLU)
//   row_softmax_f32      softmax(scale * x) per row, in place
//   layout kernels       NCHW (fp32 / fp16) <-> NHWC fp32 at the model boundary
#include "kernels.h"

namespace sd {

namespace {

constexpr int FBM = 64, FBN = 64, FBK = 16;

struct ConvF32Args {
  const float* x;
  const void* w;
  const float* bias;
  const float* res;
  float* out;
  int B, Hi, Wi, Cin, Ho, Wo, N, K, M;
  int ksize, stride, up, pad;
  int w_kind;   // 0: half [N][K]   1: float [N][K]   2: float [K][N]
};

__device__ __forceinline__ float load_w(const ConvF32Args& a, int n, int k) {
  if (n >= a.N || k >= a.K) return 0.f;
  if (a.w_kind == 0) return (float)reinterpret_cast<const half_t*>(a.w)[(size_t)n * a.K + k];
  if (a.w_kind == 1) return reinterpret_cast<const float*>(a.w)[(size_t)n * a.K + k];
  return reinterpret_cast<const float*>(a.w)[(size_t)k * a.N + n];
}

__global__ __launch_bounds__(256) void conv_f32_kernel(ConvF32Args a) {
  __shared__ float As[2][FBK][FBM + 4];
  __shared__ float Ws[2][FBK][FBN + 4];
  const int tid = threadIdx.x;
  const int m_blk = blockIdx.x * FBM, n_blk = blockIdx.y * FBN;
  // loader coordinates: row lr of the tile, 4 consecutive k starting at lk
  const int lr = tid >> 2, lk = (tid & 3) * 4;
  const int m_ld = m_blk + lr;
  int pb = 0, oy = 0, ox = 0;
  const bool m_ok = m_ld < a.M;
  if (m_ok) {
    pb = m_ld / (a.Ho * a.Wo);
    const int rem = m_ld - pb * a.Ho * a.Wo;
    oy = rem / a.Wo;
    ox = rem - oy * a.Wo;
  }
  const int Hup = a.Hi * a.up, Wup = a.Wi * a.up, ush = a.up >> 1;
  auto load_a = [&](int k) -> float {   // im2col element (m_ld, k), k = tap * Cin + c
    if (!m_ok || k >= a.K) return 0.f;
    const int tap = k / a.Cin, c = k - tap * a.Cin;
    const int ky = tap / a.ksize, kx = tap - ky * a.ksize;
    const int iy = oy * a.stride - a.pad + ky, ix = ox * a.stride - a.pad + kx;
    if (iy < 0 || iy >= Hup || ix < 0 || ix >= Wup) return 0.f;
    return a.x[(((size_t)pb * a.Hi + (iy >> ush)) * a.Wi + (ix >> ush)) * a.Cin + c];
  };
  // weight loader coordinates: K-contiguous layouts take 4 consecutive k of row wr; the [K][N] layout takes column wr of 4
  // consecutive k rows with wr fastest across lanes (coalesced along n)
  const int wr = a.w_kind == 2 ? (tid & 63) : lr, wk = a.w_kind == 2 ? (tid >> 6) * 4 : lk;
  float ra[4], rw[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) ra[e] = load_a(k0 + lk + e);
#pragma unroll
    for (int e = 0; e < 4; ++e) rw[e] = load_w(a, n_blk + wr, k0 + wk + e);
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      As[buf][lk + e][lr] = ra[e];
      Ws[buf][wk + e][wr] = rw[e];
    }
  };
  const int tx = tid & 15, ty = tid >> 4;   // 4 channels tx*4.., 4 pixels ty*4..
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  const int nk = (a.K + FBK - 1) / FBK;
  fetch(0);
  stash(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) fetch((kt + 1) * FBK);
#pragma unroll
    for (int k = 0; k < FBK; ++k) {
      const floatx4 av = *reinterpret_cast<const floatx4*>(&As[buf][k][ty * 4]);
      const floatx4 wv = *reinterpret_cast<const floatx4*>(&Ws[buf][k][tx * 4]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], wv[j], acc[i][j]);
    }
    if (kt + 1 < nk) stash(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m_blk + ty * 4 + i;
    if (m >= a.M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n_blk + tx * 4 + j;
      if (n >= a.N) continue;
      float v = acc[i][j];
      if (a.bias) v += a.bias[n];
      if (a.res) v += a.res[(size_t)m * a.N + n];
      a.out[(size_t)m * a.N + n] = v;
    }
  }
}

// ---- GroupNorm ----
constexpr int kGnF32Slabs = 64;
// grid (B * G, slabs): fp64 sum / sum of squares of this slab's pixels of one (sample, group)
__global__ __launch_bounds__(256) void groupnorm_f32_stats_kernel(const float* __restrict__ x, double* __restrict__ partial, int HW,
                                                                  int C, int G, int slabs) {
  const int bg = blockIdx.x, b = bg / G, g = bg - b * G;
  const int cpg = C / G;
  const int per = (HW + slabs - 1) / slabs;
  const int p0 = blockIdx.y * per, p1 = min(p0 + per, HW);
  double s = 0.0, q = 0.0;
  const int n = (p1 - p0) * cpg;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int p = p0 + i / cpg, c = g * cpg + i % cpg;
    const double v = (double)x[((size_t)b * HW + p) * C + c];
    s += v;
    q += v * v;
  }
  __shared__ double ss[256], sq[256];
  ss[threadIdx.x] = s;
  sq[threadIdx.x] = q;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      ss[threadIdx.x] += ss[threadIdx.x + o];
      sq[threadIdx.x] += sq[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    partial[((size_t)bg * kGnF32Slabs + blockIdx.y) * 2] = ss[0];
    partial[((size_t)bg * kGnF32Slabs + blockIdx.y) * 2 + 1] = sq[0];
  }
}
// one thread per (sample, group): mean / rstd
__global__ void groupnorm_f32_finalize_kernel(const double* __restrict__ partial, float* __restrict__ stats, int BG, int slabs,
                                              double count, float eps) {
  const int bg = blockIdx.x * blockDim.x + threadIdx.x;
  if (bg >= BG) return;
  double s = 0.0, q = 0.0;
  for (int i = 0; i < slabs; ++i) {
    s += partial[((size_t)bg * kGnF32Slabs + i) * 2];
    q += partial[((size_t)bg * kGnF32Slabs + i) * 2 + 1];
  }
  const double mean = s / count;
  double var = q / count - mean * mean;
  if (var < 0.0) var = 0.0;
  stats[bg * 2] = (float)mean;
  stats[bg * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}
__global__ __launch_bounds__(256) void groupnorm_f32_apply_kernel(const float* __restrict__ x, const float* __restrict__ stats,
                                                                  const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                  float* __restrict__ y, size_t total, int HW, int C, int G,
                                                                  int silu) {
  const int cpg = C / G;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C);
    const int b = (int)(i / ((size_t)HW * C));
    const int bg = b * G + c / cpg;
    float v = (x[i] - stats[bg * 2]) * stats[bg * 2 + 1] * gamma[c] + beta[c];
    if (silu) v = v / (1.0f + expf(-v));
    y[i] = v;
  }
}

// ---- softmax(scale * x) per row, in place; one workgroup per row ----
__global__ __launch_bounds__(256) void row_softmax_f32_kernel(float* __restrict__ x, int cols, float scale) {
  float* row = x + (size_t)blockIdx.x * cols;
  __shared__ float red[256];
  float m = -3.0e38f;
  for (int i = threadIdx.x; i < cols; i += 256) m = fmaxf(m, row[i] * scale);
  red[threadIdx.x] = m;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
    __syncthreads();
  }
  m = red[0];
  __syncthreads();
  float s = 0.f;
  for (int i = threadIdx.x; i < cols; i += 256) {
    const float e = expf(row[i] * scale - m);
    row[i] = e;
    s += e;
  }
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  const float inv = 1.0f / red[0];
  for (int i = threadIdx.x; i < cols; i += 256) row[i] *= inv;
}

// ---- layouts ----
__global__ void nchw_to_nhwc_f32_kernel(const void* __restrict__ src, int src_is_f32, float* __restrict__ dst, int B, int C, int H,
                                        int W) {
  const size_t total = (size_t)B * C * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const size_t p = i / C;                 // b*H*W + y*W + x
    const size_t b = p / ((size_t)H * W), yx = p - b * H * W;
    const size_t s = (b * C + c) * (size_t)H * W + yx;
    dst[i] = src_is_f32 ? reinterpret_cast<const float*>(src)[s] : (float)reinterpret_cast<const half_t*>(src)[s];
  }
}
__global__ void nhwc_to_nchw_f32f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int H, int W) {
  const size_t total = (size_t)B * C * H * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t yx = i % ((size_t)H * W);
    const size_t bc = i / ((size_t)H * W);
    const size_t b = bc / C, c = bc - b * C;
    dst[i] = src[(b * H * W + yx) * C + c];
  }
}

inline int grid_for(size_t total) { return (int)std::min<size_t>((total + 255) / 256, 65535); }

}  // namespace

void launch_conv_f32(const ConvF32Desc& d, hipStream_t s) {
  ConvF32Args a;
  a.x = d.x; a.w = d.w; a.bias = d.bias; a.res = d.res; a.out = d.out;
  a.B = d.B; a.Hi = d.Hi; a.Wi = d.Wi; a.Cin = d.Cin; a.Ho = d.Ho; a.Wo = d.Wo; a.N = d.N;
  a.ksize = d.ksize; a.stride = d.stride; a.up = d.up;
  a.pad = d.pad >= 0 ? d.pad : d.ksize / 2;
  a.K = d.ksize * d.ksize * d.Cin;
  a.M = d.B * d.Ho * d.Wo;
  a.w_kind = d.w_kind;
  SD_REQUIRE(d.x && d.w && d.out && a.M > 0 && a.N > 0 && a.K > 0 && (d.up == 1 || d.up == 2), kInvalidArgument, "conv_f32: bad arguments");
  const dim3 grid(cdiv(a.M, FBM), cdiv(a.N, FBN));
  SD_REQUIRE(grid.y <= 65535, kUnsupported, "conv_f32: N = %d too wide", a.N);
  hipLaunchKernelGGL(conv_f32_kernel, grid, dim3(256), 0, s, a);
  SD_HIP(hipGetLastError());
}

size_t groupnorm_f32_scratch_bytes(int B, int G) { return (size_t)B * G * (kGnF32Slabs * 2 * sizeof(double) + 2 * sizeof(float)); }

void launch_groupnorm_f32(const float* x, void* scratch, const float* gamma, const float* beta, float* y, int B, int HW, int C, int G,
                          float eps, int silu, hipStream_t s) {
  SD_REQUIRE(C % G == 0, kUnsupported, "groupnorm_f32: %d channels, %d groups", C, G);
  double* partial = reinterpret_cast<double*>(scratch);
  float* stats = reinterpret_cast<float*>(partial + (size_t)B * G * kGnF32Slabs * 2);
  const int slabs = std::max(1, std::min(kGnF32Slabs, HW / 64));
  hipLaunchKernelGGL(groupnorm_f32_stats_kernel, dim3(B * G, slabs), dim3(256), 0, s, x, partial, HW, C, G, slabs);
  hipLaunchKernelGGL(groupnorm_f32_finalize_kernel, dim3(cdiv(B * G, 64)), dim3(64), 0, s, partial, stats, B * G, slabs,
                     (double)HW * (C / G), eps);
  const size_t total = (size_t)B * HW * C;
  hipLaunchKernelGGL(groupnorm_f32_apply_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, stats, gamma, beta, y, total, HW, C, G, silu);
  SD_HIP(hipGetLastError());
}

void launch_row_softmax_f32(float* x, int rows, int cols, float scale, hipStream_t s) {
  hipLaunchKernelGGL(row_softmax_f32_kernel, dim3(rows), dim3(256), 0, s, x, cols, scale);
  SD_HIP(hipGetLastError());
}

void launch_nchw_to_nhwc_f32(const void* src, int src_is_f32, float* dst, int B, int C, int H, int W, hipStream_t s) {
  hipLaunchKernelGGL(nchw_to_nhwc_f32_kernel, dim3(grid_for((size_t)B * C * H * W)), dim3(256), 0, s, src, src_is_f32, dst, B, C, H, W);
  SD_HIP(hipGetLastError());
}

void launch_nhwc_to_nchw_f32f32(const float* src, float* dst, int B, int C, int H, int W, hipStream_t s) {
  hipLaunchKernelGGL(nhwc_to_nchw_f32f32_kernel, dim3(grid_for((size_t)B * C * H * W)), dim3(256), 0, s, src, dst, B, C, H, W);
  SD_HIP(hipGetLastError());
}

}  // namespace sd
