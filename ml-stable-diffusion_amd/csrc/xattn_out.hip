// The whole cross-attention branch of a BasicTransformerBlock as ONE launch (unet.py:586-591 around CrossAttention.forward
// :87-118): for a block of 32 query tokens,
//     q   = to_q(norm2(h1))                          LayerNormANE (layer_norm.py:51-80) folded into the projection
//     a2  = softmax(q k^T / sqrt(d)) v   per head    k / v = to_k / to_v of the prompt (hoisted out of the step)
//     h2  = h1 + to_out(a2) + b_out
// xattn.hip stops behind a2 (one workgroup per (128 tokens, head)) and `to_out` + the residual are a second launch: a 1.7-GFLOP
// GEMM that takes 11-12 us of launch floor, cold first touches and an epilogue round trip.  `to_out` mixes
// the heads, so the fusion needs a workgroup that owns ALL heads of its tokens: here a workgroup is 32 tokens x (C / 64) waves -
// wave w IS head w in the attention phase and output-channel block w in both projections:
//   phase 1  q_w^T[64][32] = Wq[64w .. 64w+64][:] . x^T : the 32 x C activation tile sits in LDS once (it is also the residual),
//            every wave streams ITS OWN 64 weight rows straight global -> VGPR in MFMA fragment order from PRE-TILED weights
//            (one fully coalesced 1-KB load per fragment, batches in flight; nothing of the weight side goes through LDS, no
//            barrier in the loop); the LayerNorm statistics of the token rows come from the same fragments (v_dot2).
//   phase 2  the 77 keys of the prompt: K / V^T fragments of head w straight from L2 into registers (they are tiny and shared by
//            every workgroup of the launch), exact softmax in registers (xattn.hip's arithmetic, line by line), a2 of head w ->
//            LDS tile [32][C].
//   phase 3  h2^T[64w .. +64][32] = Wo[64w ..][:] . a2^T + b_out + h1 : same loop as phase 1 over the LDS tile of a2.
// Two workgroup barriers in all.  Weights stay in L2 (2 x C x C x 2 B for all workgroups).  Only C = 320 / 640 (5 / 10 heads of
// 64): with 20 heads a workgroup would need 20 waves, and at M = 512 there would be 16 of them.
// PRE (five heads only): the SELF-attention's output projection in front of it, in the same launch (unet.py:588):
//   phase 0  h1^T[64w .. +64][32] = Wo1[64w ..][:] . a1^T + b1 + h0 : a1 (the self-attention's output) is the LDS tile, the result -
//            rounded to fp16 like the tensor the separate launch stores - becomes the activation tile of phase 1 and never goes to HBM.
// One more barrier; three launches (to_out, q-projection + attention, to_out) become one.
#include "kernels.h"

namespace sd {
namespace {

constexpr int XO_TOK = 32;   // query tokens per workgroup
constexpr int XO_D = 64;     // head dim
constexpr int XO_KEYS = 96;  // key capacity (3 MFMA tiles)

struct XOArgs {
  const half_t* x;
  const half8* wq_t;
  const float* q_bias;
  const float* q_colsum;
  const half_t* k;
  const half_t* vt;
  const half8* wo_t;
  const float* o_bias;
  half_t* out;
  int M, C, S, L, ldv;
  float ln_eps, scale_log2;
  // PRE: x is a1 (the self-attention's output), h0 the block input (its residual), wo1_t / o1_bias the first to_out
  const half_t* h0;
  const half8* wo1_t;
  const float* o1_bias;
};

__device__ __forceinline__ float xo_xor32_sumf(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return __uint_as_float(r0) + __uint_as_float(r1);
}
__device__ __forceinline__ float xo_xor32_maxf(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return fmaxf(__uint_as_float(r0), __uint_as_float(r1));
}

// fragment-major copy of a [N][K] row-major matrix: wt[(nb * K/16 + k16) * 64 + lane] = the 8 halves lane (l31 = lane & 31,
// hi = lane >> 5) feeds the A operand of v_mfma_f32_32x32x16_f16 for row block nb, K step k16: w[nb*32 + l31][k16*16 + hi*8 ..]
__global__ __launch_bounds__(256) void xo_retile_kernel(const half_t* __restrict__ w, half8* __restrict__ wt, int N, int K) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= N * K / 8) return;
  const int lane = idx & 63, t = idx >> 6, k16n = K >> 4;
  const int k16 = t % k16n, nb = t / k16n;
  wt[idx] = *reinterpret_cast<const half8*>(w + (size_t)(nb * 32 + (lane & 31)) * K + k16 * 16 + (lane >> 5) * 8);
}

// acc[j] += W[nb0 + j] (32 rows) . src^T over all of K: weight fragments global -> VGPR in batches of BATCH K steps, NBUF - 1
// batches ahead of the MFMAs (xo_prefetch requests the first NBUF - 1 - it may run long before the activation tile is ready);
// activation fragments from the LDS tile src [32][ROW].  STATS: row statistics of src on the side.
template <int BATCH, int NBUF>
struct XoW {
  half8 w[NBUF][2][BATCH];   // [buffer][row block][step]
};
template <int K16, int BATCH, int NBUF>
__device__ __forceinline__ void xo_prefetch(XoW<BATCH, NBUF>& r, const half8* __restrict__ wt, int nb0, int lane) {
  const half8* w0 = wt + (size_t)nb0 * K16 * 64 + lane;
  const half8* w1 = w0 + (size_t)K16 * 64;
#pragma unroll
  for (int p = 0; p < NBUF - 1; ++p)
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      r.w[p][0][i] = w0[(p * BATCH + i) * 64];
      r.w[p][1][i] = w1[(p * BATCH + i) * 64];
    }
}
template <int K16, int BATCH, int NBUF, bool STATS>
__device__ __forceinline__ void xo_gemm(XoW<BATCH, NBUF>& r, const half8* __restrict__ wt, int nb0, const half_t* src, int ROW, int lane,
                                        floatx16 (&acc)[2], float& s1, float& s2) {
  static_assert(K16 % BATCH == 0 && K16 / BATCH >= NBUF - 1, "whole batches");
  constexpr int NB = K16 / BATCH;
  const int l31 = lane & 31, hi = lane >> 5;
  const half8* w0 = wt + (size_t)nb0 * K16 * 64 + lane;
  const half8* w1 = w0 + (size_t)K16 * 64;
  const half_t* srow = src + l31 * ROW + hi * 8;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int cur = b % NBUF, nxt = (b + NBUF - 1) % NBUF;
    if (b + NBUF - 1 < NB) {
#pragma unroll
      for (int i = 0; i < BATCH; ++i) {
        r.w[nxt][0][i] = w0[((b + NBUF - 1) * BATCH + i) * 64];
        r.w[nxt][1][i] = w1[((b + NBUF - 1) * BATCH + i) * 64];
      }
    }
#pragma unroll
    for (int i = 0; i < BATCH; ++i) {
      const half8 xf = *reinterpret_cast<const half8*>(srow + (b * BATCH + i) * 16);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r.w[cur][0][i], xf, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r.w[cur][1][i], xf, acc[1], 0, 0, 0);
      if constexpr (STATS) {
        const half2v one2 = {(half_t)1.f, (half_t)1.f};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const half2v p2 = {xf[2 * e], xf[2 * e + 1]};
          s2 = __builtin_amdgcn_fdot2(p2, p2, s2, false);
          s1 = __builtin_amdgcn_fdot2(p2, one2, s1, false);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // the later batches' loads stay NBUF - 1 batches ahead, no further
  }
}

// the same over TB blocks of 32 tokens of the LDS tile (rows tb * 32 + l31): a weight fragment fetched once feeds TB MFMAs per row block
template <int K16, int BATCH, int NBUF, int TB>
__device__ __forceinline__ void xo_gemm_tb(XoW<BATCH, NBUF>& r, const half8* __restrict__ wt, int nb0, const half_t* src, int ROW, int lane,
                                           floatx16 (&acc)[TB][2]) {
  static_assert(K16 % BATCH == 0 && K16 / BATCH >= NBUF - 1, "whole batches");
  constexpr int NB = K16 / BATCH;
  const int l31 = lane & 31, hi = lane >> 5;
  const half8* w0 = wt + (size_t)nb0 * K16 * 64 + lane;
  const half8* w1 = w0 + (size_t)K16 * 64;
  const half_t* srow = src + l31 * ROW + hi * 8;
#pragma unroll
  for (int b = 0; b < NB; ++b) {
    const int cur = b % NBUF, nxt = (b + NBUF - 1) % NBUF;
    if (b + NBUF - 1 < NB) {
#pragma unroll
      for (int i = 0; i < BATCH; ++i) {
        r.w[nxt][0][i] = w0[((b + NBUF - 1) * BATCH + i) * 64];
        r.w[nxt][1][i] = w1[((b + NBUF - 1) * BATCH + i) * 64];
      }
    }
#pragma unroll
    for (int i = 0; i < BATCH; ++i)
#pragma unroll
      for (int tb = 0; tb < TB; ++tb) {
        const half8 xf = *reinterpret_cast<const half8*>(srow + tb * 32 * ROW + (b * BATCH + i) * 16);
        acc[tb][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r.w[cur][0][i], xf, acc[tb][0], 0, 0, 0);
        acc[tb][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(r.w[cur][1][i], xf, acc[tb][1], 0, 0, 0);
      }
    __builtin_amdgcn_sched_barrier(0);   // the later batches' loads stay NBUF - 1 batches ahead, no further
  }
}

constexpr size_t xo_lds_bytes(int C) {   // two activation tiles, the per-column constants, PRE: LayerNorm partials [2 C / 64][32] (sum, sumsq)
  return (size_t)2 * XO_TOK * (C + 8) * 2 + (size_t)4 * C * sizeof(float) + (size_t)2 * (C / 64) * 32 * 2 * sizeof(float);
}

template <int NW, bool PRE>
__global__ __launch_bounds__(NW * 64) void xattn_out_kernel(XOArgs a) {
  constexpr int C = NW * 64, ROW = C + 8, K16 = C / 16, NT = NW * 64;
  constexpr int BATCH = 4, NBUF = NW <= 5 ? 3 : 2;         // five heads: 256 VGPRs per wave, two batches ahead; ten: 168, one
  static_assert(!PRE || NW <= 5, "the self-attention's to_out in front: five heads only");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* xs = reinterpret_cast<half_t*>(smem);            // [32][ROW]  h1 rows: LayerNorm source, residual
  half_t* os = xs + XO_TOK * ROW;                          // [32][ROW]  a2 (all heads); PRE: a1 until phase 0 is over
  float* sconst = reinterpret_cast<float*>(os + XO_TOK * ROW);   // [C] q bias | [C] q colsum | [C] out bias | [C] first out bias (PRE)
  float2* lnp = reinterpret_cast<float2*>(sconst + 4 * C);       // PRE: [2 NW][32] norm2 partials of h1 per token

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = head = 64-channel output block
  const int l31 = lane & 31, hi = lane >> 5;
  const int m_blk = blockIdx.x * XO_TOK;
  const int b = m_blk / a.S;                               // S % 32 == 0: one sample per workgroup

  // ---- the activation tile (PRE: a1), the per-column constants (NT == C threads: 4 chunks / 1 column each), the first weight
  // batches and the prompt's K fragments of this head: every load is requested before the first one is used ----
  half8 xv[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + NT * i, row = idx / (C / 8), c8 = idx - row * (C / 8);
    xv[i] = *reinterpret_cast<const half8*>(a.x + (size_t)(m_blk + row) * C + c8 * 8);
  }
  XoW<BATCH, NBUF> wr;
  xo_prefetch<K16, BATCH, NBUF>(wr, PRE ? a.wo1_t : a.wq_t, 2 * wave, lane);
  const float qb = a.q_bias[tid], qc = a.q_colsum[tid], ob = a.o_bias[tid];
  float ob1 = 0.f;
  half4 h0r[2][4];                                         // PRE: the residual of phase 0, this lane's 32 channels of its token
  if constexpr (PRE) {
    ob1 = a.o1_bias[tid];
    const half_t* hrow = a.h0 + (size_t)(m_blk + l31) * C + wave * XO_D + 4 * hi;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) h0r[j][g] = *reinterpret_cast<const half4*>(hrow + j * 32 + 8 * g);
  }
  // scores^T[key][q] = K . Q^T: lane (l31 = key of the tile, hi) holds, for step (j, s), channels j*32 + 16*s + 4*hi + {0..3, 8..11}
  // of its key - the order in which the q accumulators come out of phase 1 (xattn.hip); keys >= L are clamped and masked later
  half4 kq[3][2][2][2];
  {
    const half_t* kb = a.k + (size_t)b * a.L * C + (size_t)wave * XO_D + 4 * hi;
#pragma unroll
    for (int kt = 0; kt < 3; ++kt) {
      const int key = min(kt * 32 + l31, a.L - 1);
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          kq[kt][j][s][0] = *reinterpret_cast<const half4*>(kb + (size_t)key * C + j * 32 + 16 * s);
          kq[kt][j][s][1] = *reinterpret_cast<const half4*>(kb + (size_t)key * C + j * 32 + 16 * s + 8);
        }
    }
  }
  {
    half_t* dst = PRE ? os : xs;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int idx = tid + NT * i, row = idx / (C / 8), c8 = idx - row * (C / 8);
      *reinterpret_cast<half8*>(dst + row * ROW + c8 * 8) = xv[i];
    }
  }
  sconst[tid] = qb;
  sconst[C + tid] = qc;
  sconst[2 * C + tid] = ob;
  if constexpr (PRE) sconst[3 * C + tid] = ob1;
  __syncthreads();                                         // the tile and the constants are visible

  floatx16 acc[2];
  float ln_s1 = 0.f, ln_s2 = 0.f;
  if constexpr (PRE) {
    // ---- phase 0: h1^T[64 * wave ..][32] = Wo1 . a1^T + b1 + h0 -> the activation tile of phase 1 (fp16, as the tensor would be) ----
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float d1 = 0.f, d2 = 0.f;
    xo_gemm<K16, BATCH, NBUF, false>(wr, a.wo1_t, 2 * wave, os, ROW, lane, acc, d1, d2);
    xo_prefetch<K16, BATCH, NBUF>(wr, a.wq_t, 2 * wave, lane);   // phase 1's first batches: in flight under the epilogue and the barrier
    half_t* hrow = xs + l31 * ROW + wave * XO_D + 4 * hi;
    float p1 = 0.f, p2 = 0.f;                              // norm2's statistics from here (the fp16 h1 the tile holds), not from the
#pragma unroll                                             // q pass's K loop: 1 300 cycles per wave less (r6_gq_clock.py on the head kernel)
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const floatx4 b1 = *reinterpret_cast<const floatx4*>(sconst + 3 * C + wave * XO_D + j * 32 + 8 * g + 4 * hi);
        half4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (half_t)((float)(half_t)(acc[j][4 * g + e] + b1[e]) + (float)h0r[j][g][e]);
          const float r = (float)o[e];
          p1 += r;
          p2 = fmaf(r, r, p2);
        }
        *reinterpret_cast<half4*>(hrow + j * 32 + 8 * g) = o;
      }
    lnp[(wave * 2 + hi) * 32 + l31] = float2{p1, p2};
    __syncthreads();                                       // h1 of every channel block is in the tile; nobody reads a1 any more
#pragma unroll
    for (int i = 0; i < 2 * NW; ++i) {                     // fixed order; both half-waves of a token hold the full sums
      const float2 v = lnp[i * 32 + l31];
      ln_s1 += v.x;
      ln_s2 += v.y;
    }
  }

  // ---- phase 1: q^T of head `wave`, LayerNorm statistics of the token rows on the side ----
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  if constexpr (PRE) {
    float d1 = 0.f, d2 = 0.f;
    xo_gemm<K16, BATCH, NBUF, false>(wr, a.wq_t, 2 * wave, xs, ROW, lane, acc, d1, d2);
  } else {
    xo_gemm<K16, BATCH, NBUF, true>(wr, a.wq_t, 2 * wave, xs, ROW, lane, acc, ln_s1, ln_s2);
  }

  // ---- V^T fragments of this head (O^T[d][q] = V^T . P^T): lane (l31 = channel of the tile, hi), step (kt, s2): keys
  // kt*32 + s2*16 + 4*hi + {0..3, 8..11}; columns [L, ldv) are zero by contract, columns >= ldv do not exist.  Five heads: requested
  // here, in flight under the LayerNorm fold and the scores; ten heads (168 VGPRs per wave): behind the softmax ----
  half4 vq[3][2][2][2];
  auto load_v = [&]() {
    const half4 z4 = {0, 0, 0, 0};
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const half_t* vb = a.vt + ((size_t)b * C + (size_t)wave * XO_D + ct * 32 + l31) * a.ldv + 4 * hi;
#pragma unroll
      for (int kt = 0; kt < 3; ++kt)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          const int c0 = kt * 32 + s2 * 16 + 4 * hi;
          vq[kt][s2][ct][0] = (c0 < a.ldv) ? *reinterpret_cast<const half4*>(vb + kt * 32 + s2 * 16) : z4;
          vq[kt][s2][ct][1] = (c0 + 8 < a.ldv) ? *reinterpret_cast<const half4*>(vb + kt * 32 + s2 * 16 + 8) : z4;
        }
    }
  };
  if constexpr (NW <= 5) load_v();

  // ---- q of this lane's token: LayerNorm fold, fp16, MFMA B-operand order (xattn.hip) ----
  // acc[j][r]: channel n = j*32 + (r&3) + 8*(r>>2) + 4*hi of token (lane & 31) -> k-slot e of step (j, s) is r = 8*s + e
  half8 qf[2][2];
  {
    const float inv_k = 1.0f / (float)C;
    const float s1 = PRE ? ln_s1 : xo_xor32_sumf(ln_s1), s2 = PRE ? ln_s2 : xo_xor32_sumf(ln_s2);
    const float mean = s1 * inv_k;
    const float var = fmaxf(s2 * inv_k - mean * mean, 0.f);
    const float ln_a = rsqrtf(var + a.ln_eps);
    const float ln_b = -ln_a * mean;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int nl = wave * XO_D + j * 32 + 8 * q4 + 4 * hi;
        const floatx4 bb = *reinterpret_cast<const floatx4*>(sconst + nl);
        const floatx4 cs = *reinterpret_cast<const floatx4*>(sconst + C + nl);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * q4 + e;
          qf[j][r >> 3][r & 7] = (half_t)fmaf(acc[j][r], ln_a, fmaf(ln_b, cs[e], bb[e]));
        }
      }
  }

  // ---- phase 2: scores^T[key][q], exact softmax over the keys ----
  floatx16 sacc[3];
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const half4 k0 = kq[kt][j][s][0], k1 = kq[kt][j][s][1];
        const half8 kf = {k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]};
        sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[j][s], sacc[kt], 0, 0, 0);
      }
  }
  // the first batches of to_out's weights: five heads have the registers to keep them in flight under the softmax and P . V
  if constexpr (NW <= 5) xo_prefetch<K16, BATCH, NBUF>(wr, a.wo_t, 2 * wave, lane);
  float mx = -3.0e38f;
#pragma unroll
  for (int kt = 0; kt < 3; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (key >= a.L) sacc[kt][r] = -3.0e38f;
      mx = fmaxf(mx, sacc[kt][r]);
    }
  mx = xo_xor32_maxf(mx);                                  // the other half-wave holds the other keys of each tile
  const float mnew = mx * a.scale_log2;
  float ps4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kt = 0; kt < 3; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __builtin_amdgcn_exp2f(fmaf(sacc[kt][r], a.scale_log2, -mnew));
      sacc[kt][r] = p;
      ps4[r & 3] += p;
    }
  const float inv = 1.0f / xo_xor32_sumf((ps4[0] + ps4[1]) + (ps4[2] + ps4[3]));
  if constexpr (NW > 5) load_v();

  floatx16 oacc[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) oacc[ct][r] = 0.f;
#pragma unroll
  for (int kt = 0; kt < 3; ++kt)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      half8 pf;
#pragma unroll
      for (int e = 0; e < 8; ++e) pf[e] = (half_t)sacc[kt][s2 * 8 + e];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const half4 v0 = vq[kt][s2][ct][0], v1 = vq[kt][s2][ct][1];
        const half8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[ct], 0, 0, 0);
      }
    }
  // a2 of head `wave`, fp16 like the tensor xattn.hip stores: oacc[ct][r] = channel ct*32 + (r&3) + 8*(r>>2) + 4*hi of token l31
  {
    half_t* orow = os + l31 * ROW + wave * XO_D;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const half4 o = {(half_t)(oacc[ct][4 * g] * inv), (half_t)(oacc[ct][4 * g + 1] * inv), (half_t)(oacc[ct][4 * g + 2] * inv),
                         (half_t)(oacc[ct][4 * g + 3] * inv)};
        *reinterpret_cast<half4*>(orow + ct * 32 + 8 * g + 4 * hi) = o;
      }
  }
  __syncthreads();                                         // every head's a2 is in the tile

  // ---- phase 3: h2^T[64 * wave ..][32] = Wo . a2^T, + bias + residual ----
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float e1 = 0.f, e2 = 0.f;
  if constexpr (NW > 5) xo_prefetch<K16, BATCH, NBUF>(wr, a.wo_t, 2 * wave, lane);
  xo_gemm<K16, BATCH, NBUF, false>(wr, a.wo_t, 2 * wave, os, ROW, lane, acc, e1, e2);
  {
    half_t* orow = a.out + (size_t)(m_blk + l31) * C + wave * XO_D;
    const half_t* rrow = xs + l31 * ROW + wave * XO_D;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int nl = j * 32 + 8 * g + 4 * hi;
        const floatx4 bo = *reinterpret_cast<const floatx4*>(sconst + 2 * C + wave * XO_D + nl);
        const half4 rr = *reinterpret_cast<const half4*>(rrow + nl);
        // the two-launch path rounds to_out's result + bias to fp16 in registers only after the residual add (tile_epilogue adds the
        // residual to the fp16-rounded value): round the same way
        half4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)((float)(half_t)(acc[j][4 * g + e] + bo[e]) + (float)rr[e]);
        *reinterpret_cast<half4*>(orow + nl) = o;
      }
  }
}

template <int NW, bool PRE>
void launch_nw(const XOArgs& a, hipStream_t s) {
  constexpr size_t lds = xo_lds_bytes(NW * 64);
  static_assert(lds <= 160 * 1024, "LDS");
  auto k = xattn_out_kernel<NW, PRE>;
  static DynLdsOnce once;
  once.set(k, lds);
  hipLaunchKernelGGL(k, dim3(a.M / XO_TOK), dim3(NW * 64), lds, s, a);
}

// ---------------------------------------------------------------------------------------------
// The tail of a SpatialTransformer as ONE launch (unet.py:591 FeedForward.net.2 + residual, then :561-563 proj_out + residual):
//     h3  = h2 + ff.net.2(g) + b          g = the GEGLU product [M][4C]
//     out = x  + proj_out(h3) + b_p       x = the transformer's input
// Same workgroup shape as above (32 tokens x C / 64 waves, wave w = output-channel block w of both GEMMs, weights global -> VGPR
// in fragment order): phase A walks K = 4C over the LDS tile of g, its result - fp16 like the tensor the separate launch stores -
// is the LDS tile phase B multiplies; h3 never goes to HBM.  The output leaves through an LDS tile: coalesced 16-B stores, and
// - when the consumer is a GroupNorm (the next resnet's norm1, unet.py:472) - its (sum, sumsq) partial per (sample, group) of
// this 32-token tile, in the format of the conv epilogues (entry = tile index inside the sample; fixed-order sums, no atomics).
// ---------------------------------------------------------------------------------------------
struct FPArgs {
  const half_t* g;
  const half8* w1_t;
  const float* b1;
  const half_t* res1;
  const half8* w2_t;
  const float* b2;
  const half_t* res2;
  half_t* out;
  float* gn_partial;
  int M, S, gn_G;
};

template <int NW>
constexpr size_t fp_lds_bytes() {
  return (size_t)XO_TOK * (4 * NW * 64 + 8) * 2 + (size_t)XO_TOK * (NW * 64 + 8) * 2 + (size_t)4 * NW * 64 * sizeof(float);
}

template <int NW>
__global__ __launch_bounds__(NW * 64) void ffn_proj_kernel(FPArgs a) {
  constexpr int C = NW * 64, K1 = 4 * C, ROWG = K1 + 8, ROW = C + 8, NT = NW * 64;
  constexpr int BATCH = 4, NBUF = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* gs = reinterpret_cast<half_t*>(smem);            // [32][ROWG]  g rows; behind phase A: the output tile [32][ROW]
  half_t* ts = gs + XO_TOK * ROWG;                         // [32][ROW]   h3
  float* sconst = reinterpret_cast<float*>(ts + XO_TOK * ROW);   // [C] b1 | [C] b2 | [C] column sums | [C] column sums of squares

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int m_blk = blockIdx.x * XO_TOK;

  // ---- everything this workgroup reads before phase A is requested up front: weight batches, the g tile (16 chunks per thread),
  // both residual slices of this lane, the constants ----
  XoW<BATCH, NBUF> wr;
  xo_prefetch<K1 / 16, BATCH, NBUF>(wr, a.w1_t, 2 * wave, lane);
  constexpr int GCH = XO_TOK * (K1 / 8) / NT;              // = 16
  half8 gv[GCH];
#pragma unroll
  for (int i = 0; i < GCH; ++i) {
    const int idx = tid + NT * i, row = idx / (K1 / 8), c8 = idx - row * (K1 / 8);
    gv[i] = *reinterpret_cast<const half8*>(a.g + (size_t)(m_blk + row) * K1 + c8 * 8);
  }
  const float cb1 = a.b1[tid], cb2 = a.b2[tid];
  half4 r1[2][4], r2[2][4];
  {
    const size_t off = (size_t)(m_blk + l31) * C + wave * XO_D + 4 * hi;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        r1[j][q] = *reinterpret_cast<const half4*>(a.res1 + off + j * 32 + 8 * q);
        r2[j][q] = *reinterpret_cast<const half4*>(a.res2 + off + j * 32 + 8 * q);
      }
  }
#pragma unroll
  for (int i = 0; i < GCH; ++i) {
    const int idx = tid + NT * i, row = idx / (K1 / 8), c8 = idx - row * (K1 / 8);
    *reinterpret_cast<half8*>(gs + row * ROWG + c8 * 8) = gv[i];
  }
  sconst[tid] = cb1;
  sconst[C + tid] = cb2;
  __syncthreads();                                         // the g tile and the constants are visible

  // ---- phase A: h3^T[64 * wave ..][32] = W1 . g^T + b1 + h2 -> LDS tile (fp16, as the tensor would be) ----
  floatx16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float d1 = 0.f, d2 = 0.f;
  xo_gemm<K1 / 16, BATCH, NBUF, false>(wr, a.w1_t, 2 * wave, gs, ROWG, lane, acc, d1, d2);
  xo_prefetch<C / 16, BATCH, NBUF>(wr, a.w2_t, 2 * wave, lane);   // phase B's first batches: in flight under the epilogue and the barrier
  {
    half_t* trow = ts + l31 * ROW + wave * XO_D + 4 * hi;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const floatx4 bb = *reinterpret_cast<const floatx4*>(sconst + wave * XO_D + j * 32 + 8 * q + 4 * hi);
        half4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)((float)(half_t)(acc[j][4 * q + e] + bb[e]) + (float)r1[j][q][e]);
        *reinterpret_cast<half4*>(trow + j * 32 + 8 * q) = o;
      }
  }
  __syncthreads();                                         // h3 of every channel block is in the tile; nobody reads g any more

  // ---- phase B: out^T[64 * wave ..][32] = Wp . h3^T + b_p + x -> LDS tile (over the g tile) ----
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  xo_gemm<C / 16, BATCH, NBUF, false>(wr, a.w2_t, 2 * wave, ts, ROW, lane, acc, d1, d2);
  half_t* ot = gs;                                         // [32][ROW]
  {
    half_t* orow = ot + l31 * ROW + wave * XO_D + 4 * hi;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const floatx4 bb = *reinterpret_cast<const floatx4*>(sconst + C + wave * XO_D + j * 32 + 8 * q + 4 * hi);
        half4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (half_t)((float)(half_t)(acc[j][4 * q + e] + bb[e]) + (float)r2[j][q][e]);
        *reinterpret_cast<half4*>(orow + j * 32 + 8 * q) = o;
      }
  }
  __syncthreads();                                         // the output tile is complete
  // coalesced stores: NT == C threads, 4 chunks of 16 B each
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int idx = tid + NT * i, row = idx / (C / 8), c8 = idx - row * (C / 8);
    out_store(reinterpret_cast<half8*>(a.out + (size_t)(m_blk + row) * C + c8 * 8), *reinterpret_cast<const half8*>(ot + row * ROW + c8 * 8));
  }
  if (a.gn_partial == nullptr) return;                     // (block-uniform)
  // GroupNorm statistics of the stored (fp16-rounded) values: thread = channel, fixed-order sums over the 32 tokens, then
  // thread = group folds its channels
  {
    float s = 0.f, q = 0.f;
#pragma unroll 8
    for (int r = 0; r < XO_TOK; ++r) {
      const float f = (float)ot[r * ROW + tid];
      s += f;
      q = fmaf(f, f, q);
    }
    sconst[2 * C + tid] = s;
    sconst[3 * C + tid] = q;
  }
  __syncthreads();
  if (tid < a.gn_G) {
    const int cpg = C / a.gn_G;
    float s = 0.f, q = 0.f;
    for (int c = tid * cpg; c < (tid + 1) * cpg; ++c) {
      s += sconst[2 * C + c];
      q += sconst[3 * C + c];
    }
    const int b = m_blk / a.S, tile = (m_blk - b * a.S) / XO_TOK;
    float* dst = a.gn_partial + (((size_t)b * a.gn_G + tid) * kGnMaxSlabs + tile) * 2;
    dst[0] = s;
    dst[1] = q;
  }
}

// ---------------------------------------------------------------------------------------------
// The HEAD of a SpatialTransformer as ONE launch (round 6; unet.py:553-556 norm -> proj_in, then the first block's :583-586 norm1 ->
// :74-84 to_q | to_k | to_v):
//     xn      = GroupNorm(x)                         statistics from the producer's epilogue partials (GnHook), affine applied here
//     h       = proj_in(xn) + b                      stored: it is the residual stream of the transformer block
//     q|k|v   = [Wq | Wk | Wv] . LayerNorm(h)        LayerNorm folded (UNet::fold_layernorm), q pre-scaled, V^T in attention8's key order
// Same workgroup shape as xattn_out_kernel (32 tokens x C / 64 waves, wave w = 64-channel output block w of all four GEMM passes,
// weights global -> VGPR in fragment order); the GroupNorm'd tile and h live in LDS, every wave stages ITS 64-channel block of an
// output in its own LDS region (no workgroup barrier: only that wave reads it) and writes whole 128-byte row pieces / 8-token V^T
// chunks.  Three launches (GroupNorm apply, proj_in, fused q|k|v: 10 + 11.6 + 20.5 us in sequence) become one.  C = 320 only.
// gn_entries == 0: x arrives normalised already (the fall-back when the producer left no statistics).
// ---------------------------------------------------------------------------------------------
struct GQArgs {
  const half_t* x;
  const float* gn_partial;
  const float* gn_gamma;
  const float* gn_beta;
  const half8* wp_t;
  const float* p_bias;
  half_t* h;
  const half8* wqkv_t;
  const float* qkv_bias;
  const float* qkv_colsum;
  half_t* qk;
  half_t* vt;
  int M, S, ldT, vt_perm, gn_entries, gn_G;
  float gn_eps, ln_eps, q_scale;
  long long* clk;   // phase clock (GnProjQkvDesc::clk): [workgroup][wave][16] cycle stamps, or nullptr
};

constexpr int GQ_WST = 5120;   // per-wave staging: [32 tokens][72] halves row-major (q / k) or [64 channels][40] halves transposed (V^T)
template <int NW, int TB>
constexpr size_t gq_xs_bytes() {   // the input tile [32 TB][C + 8] fp16, later the waves' staging regions
  return std::max((size_t)32 * TB * (NW * 64 + 8) * 2, (size_t)NW * GQ_WST);
}
template <int NW, int TB>
constexpr size_t gq_lds_bytes() {
  return gq_xs_bytes<NW, TB>() + (size_t)32 * TB * (NW * 64 + 8) * 2 + (size_t)9 * NW * 64 * sizeof(float) + 128 * sizeof(float) +
         (size_t)2 * NW * 32 * TB * sizeof(float2);
}

// TB: blocks of 32 tokens per workgroup.  2 where the grid still fills the chip (from two prompts per GPU): every weight fragment a wave
// fetches from L2 then feeds two MFMAs per row block - at eight prompts per GPU the 32-token form re-reads 800 KB of weights per 32
// tokens (1.6 GB per launch) and two resident workgroups per CU did not speed it up; the 64-token form runs 107 us against 125.
template <int NW, int TB>
__global__ __launch_bounds__(NW * 64) void gn_proj_qkv_kernel(GQArgs a) {
  constexpr int C = NW * 64, ROW = C + 8, K16 = C / 16, NT = NW * 64, TOK = 32 * TB;
  constexpr int BATCH = 4, NBUF = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* xs = reinterpret_cast<half_t*>(smem);                      // [32][ROW] GroupNorm'd input rows
  // the waves' staging regions [NW][GQ_WST] lie over the xs tile, which is dead once proj_in has consumed it (behind the barrier
  // that publishes the h tile): 61 KB per workgroup instead of 82, TWO workgroups per CU where the grid has them (eight prompts per GPU)
  char* wst_all = smem;
  constexpr size_t XS_BYTES = gq_xs_bytes<NW, TB>();
  half_t* hs = reinterpret_cast<half_t*>(smem + XS_BYTES);           // [TOK][ROW] proj_in output rows (LayerNorm source)
  float* sconst = reinterpret_cast<float*>(hs + TOK * ROW);       // gsc[C] | gsh[C] | pb[C] | qb[3C] | qc[3C]
  float* gstat = sconst + 9 * C;                                     // mean[64] | rstd[64]
  float2* lnp = reinterpret_cast<float2*>(gstat + 128);              // [2 NW][TOK] LayerNorm partial (sum, sumsq) of h per token

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int m_blk = blockIdx.x * TOK;
  const int b = m_blk / a.S;                                         // S % TOK == 0: one sample per workgroup
  half_t* wst = reinterpret_cast<half_t*>(wst_all + wave * GQ_WST);
  auto tick = [&](int i) {
    if (a.clk && lane == 0) a.clk[((size_t)blockIdx.x * NW + wave) * 16 + i] = (long long)__builtin_readcyclecounter();
  };
  tick(0);

  // ---- everything that can be requested up front, in the order it is needed (a wave's loads return in order): the producer's
  //      GroupNorm partials (8 lanes per group, each a contiguous run of 16 entries as 8 independent float4 loads, then a fixed
  //      shuffle tree - a first version walked the entries in a dependent loop: 10 200 of the wave's 32 600 cycles,
  //      tools/r6_gq_clock.py), the input tile, the first weight batches of proj_in, the per-column constants ----
  const bool gn = a.gn_entries > 0;                                  // kernel-uniform
  constexpr int per = 16;                                            // 8 lanes x 16 entries: launch_gn_proj_qkv takes <= 128 entries
  floatx4 pv[per / 2];
  const int fg = tid >> 3, fj = tid & 7;
  if (gn && tid < 256 && fg < a.gn_G) {
    const floatx4* src = reinterpret_cast<const floatx4*>(a.gn_partial + (((size_t)b * a.gn_G + fg) * kGnMaxSlabs + fj * per) * 2);
#pragma unroll
    for (int k = 0; k < per / 2; ++k) {
      const int e0 = fj * per + 2 * k;                               // first of the two entries of this float4
      pv[k] = (e0 < a.gn_entries) ? src[k] : floatx4{0.f, 0.f, 0.f, 0.f};
    }
  }
  half8 xv[4 * TB];
#pragma unroll
  for (int i = 0; i < 4 * TB; ++i) {
    const int idx = tid + NT * i, row = idx / (C / 8), c8 = idx - row * (C / 8);
    xv[i] = *reinterpret_cast<const half8*>(a.x + (size_t)(m_blk + row) * C + c8 * 8);
  }
  XoW<BATCH, NBUF> wr;
  xo_prefetch<K16, BATCH, NBUF>(wr, a.wp_t, 2 * wave, lane);
  const float pb = a.p_bias[tid];
  const float gam = gn ? a.gn_gamma[tid] : 1.f, bet = gn ? a.gn_beta[tid] : 0.f;
  float qb[3], qc[3];
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    qb[p] = a.qkv_bias[p * C + tid];
    qc[p] = a.qkv_colsum[p * C + tid];
  }
  const int cpg = C / a.gn_G;
  if (gn) {
    if (tid < 256) {
      const int g = fg, j = fj;
      float s = 0.f, q = 0.f;
      if (g < a.gn_G) {
#pragma unroll
        for (int k = 0; k < per / 2; ++k) {
          const int e0 = j * per + 2 * k;
          floatx4 v = (e0 < a.gn_entries) ? pv[k] : floatx4{0.f, 0.f, 0.f, 0.f};
          if (e0 + 1 >= a.gn_entries) {
            v[2] = 0.f;
            v[3] = 0.f;
          }
          s += v[0] + v[2];
          q += v[1] + v[3];
        }
      }
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) {
        s += __shfl_xor(s, o);
        q += __shfl_xor(q, o);
      }
      if (j == 0 && g < a.gn_G) {
        const float inv_n = 1.0f / ((float)cpg * (float)a.S);
        const float mean = s * inv_n;
        const float var = fmaxf(q * inv_n - mean * mean, 0.f);
        gstat[g] = mean;
        gstat[64 + g] = rsqrtf(var + a.gn_eps);
      }
    }
    __syncthreads();
    tick(1);
    const int g = tid / cpg;
    const float sc = gstat[64 + g] * gam;
    sconst[tid] = sc;
    sconst[C + tid] = bet - gstat[g] * sc;
  }
  sconst[2 * C + tid] = pb;
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    sconst[3 * C + p * C + tid] = qb[p];
    sconst[6 * C + p * C + tid] = qc[p];
  }
  __syncthreads();                                                   // scale / shift and the constants are visible
  tick(2);
#pragma unroll
  for (int i = 0; i < 4 * TB; ++i) {
    const int idx = tid + NT * i, row = idx / (C / 8), c8 = idx - row * (C / 8);
    half8 v = xv[i];
    if (gn) {
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] * sconst[c8 * 8 + e] + sconst[C + c8 * 8 + e]);
    }
    *reinterpret_cast<half8*>(xs + row * ROW + c8 * 8) = v;
  }
  __syncthreads();                                                   // the normalised tile is in LDS
  tick(3);

  // ---- proj_in: h^T[64 * wave ..][32] = Wp . xn^T + b -> LDS tile (fp16, as the tensor the separate launch stores) ----
  floatx16 acc[TB][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int tb = 0; tb < TB; ++tb)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tb][j][r] = 0.f;
  };
  zero_acc();
  xo_gemm_tb<K16, BATCH, NBUF, TB>(wr, a.wp_t, 2 * wave, xs, ROW, lane, acc);
  tick(4);
  xo_prefetch<K16, BATCH, NBUF>(wr, a.wqkv_t, 2 * wave, lane);       // q pass: first batches in flight under the epilogue and the barrier
#pragma unroll
  for (int tb = 0; tb < TB; ++tb) {
    half_t* hrow = hs + (tb * 32 + l31) * ROW + wave * XO_D + 4 * hi;
    float p1 = 0.f, p2 = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const floatx4 b1 = *reinterpret_cast<const floatx4*>(sconst + 2 * C + wave * XO_D + j * 32 + 8 * g + 4 * hi);
        half4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          o[e] = (half_t)(acc[tb][j][4 * g + e] + b1[e]);
          const float r = (float)o[e];                               // norm1 sees the fp16 tensor, as behind the separate launch
          p1 += r;
          p2 = fmaf(r, r, p2);
        }
        *reinterpret_cast<half4*>(hrow + j * 32 + 8 * g) = o;
      }
    lnp[(wave * 2 + hi) * TOK + tb * 32 + l31] = float2{p1, p2};     // this lane's 32 of the token's 320 channels
  }
  __syncthreads();                                                   // h of every channel block is in the tile
  tick(5);
#pragma unroll
  for (int i = 0; i < 4 * TB; ++i) {                                 // the residual stream leaves in whole rows
    const int idx = tid + NT * i, row = idx / (C / 8), c8 = idx - row * (C / 8);
    *reinterpret_cast<half8*>(a.h + (size_t)(m_blk + row) * C + c8 * 8) = *reinterpret_cast<const half8*>(hs + row * ROW + c8 * 8);
  }

  // ---- q | k | v: three passes over the h tile; LayerNorm statistics of the token rows from the fragments of the first ----
  float la[TB], lb[TB];                                              // LayerNorm of token tb * 32 + l31: out = acc * la + lb * colsum + bias
#pragma unroll
  for (int tb = 0; tb < TB; ++tb) {
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int i = 0; i < 2 * NW; ++i) {                               // fixed order, every lane of the token the same sum
      const float2 v = lnp[i * TOK + tb * 32 + l31];
      t1 += v.x;
      t2 += v.y;
    }
    const float inv_k = 1.0f / (float)C;
    const float mean = t1 * inv_k;
    la[tb] = rsqrtf(fmaxf(t2 * inv_k - mean * mean, 0.f) + a.ln_eps);
    lb[tb] = -la[tb] * mean;
  }
  const int sp0 = m_blk - b * a.S;
#pragma unroll
  for (int p = 0; p < 3; ++p) {
    zero_acc();
    xo_gemm_tb<K16, BATCH, NBUF, TB>(wr, a.wqkv_t, p * (C / 32) + 2 * wave, hs, ROW, lane, acc);
    if (p < 2) xo_prefetch<K16, BATCH, NBUF>(wr, a.wqkv_t, (p + 1) * (C / 32) + 2 * wave, lane);
    tick(6 + 2 * p);
    const float qs = p == 0 ? a.q_scale : 1.f;
#pragma unroll
    for (int tb = 0; tb < TB; ++tb) {
      // acc[tb][j][4 g + e]: channel j*32 + 8 g + 4 hi + e of the wave's block, token tb * 32 + l31
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int nl = j * 32 + 8 * g + 4 * hi;
          const floatx4 b4 = *reinterpret_cast<const floatx4*>(sconst + 3 * C + p * C + wave * XO_D + nl);
          const floatx4 c4 = *reinterpret_cast<const floatx4*>(sconst + 6 * C + p * C + wave * XO_D + nl);
          half4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = (half_t)(fmaf(acc[tb][j][4 * g + e], la[tb], fmaf(lb[tb], c4[e], b4[e])) * qs);
          if (p < 2) {
            *reinterpret_cast<half4*>(wst + l31 * 72 + nl) = o;
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) wst[(nl + e) * 40 + l31] = o[e];
          }
        }
      // the wave's own staging region back out (DS operations of one wave execute in order: no barrier)
      if (p < 2) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int id = lane + 64 * it, tok = id >> 3, c = id & 7;
          const half8 v = *reinterpret_cast<const half8*>(wst + tok * 72 + c * 8);
          *reinterpret_cast<half8*>(a.qk + (size_t)(m_blk + tb * 32 + tok) * (2 * C) + p * C + wave * XO_D + c * 8) = v;
        }
      } else {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int id = lane + 64 * it, ch = id >> 2, c = id & 3;
          half8 v;
          if (a.vt_perm) {   // chunk c = tokens 16 j + 4 o + {0..3} and 16 j + 8 + 4 o + {0..3}  (j = c >> 1, o = c & 1): AttnDesc::vt_perm
            const half_t* src = wst + ch * 40 + (c >> 1) * 16 + (c & 1) * 4;
            const half4 lo = *reinterpret_cast<const half4*>(src), up = *reinterpret_cast<const half4*>(src + 8);
            v = half8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
          } else {
            v = *reinterpret_cast<const half8*>(wst + ch * 40 + c * 8);
          }
          *reinterpret_cast<half8*>(a.vt + ((size_t)b * C + wave * XO_D + ch) * a.ldT + sp0 + tb * 32 + c * 8) = v;
        }
      }
    }
    tick(7 + 2 * p);
  }
}

}  // namespace

bool xattn_out_ok(int C, int heads, int S, int L) {
  return (heads == 5 || heads == 10) && C == heads * XO_D && S >= XO_TOK && S % XO_TOK == 0 && L >= 1 && L <= XO_KEYS;
}

void launch_xattn_out_retile(const half_t* w, half_t* wt, int C, hipStream_t s) {
  SD_REQUIRE(C % 32 == 0, kInvalidArgument, "xattn_out retile: C=%d", C);
  const int n = C * C / 8;
  hipLaunchKernelGGL(xo_retile_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, reinterpret_cast<half8*>(wt), C, C);
  SD_HIP(hipGetLastError());
}

void launch_xattn_out(const XAttnOutDesc& d, hipStream_t s) {
  SD_REQUIRE(xattn_out_ok(d.C, d.heads, d.S, d.L), kInvalidArgument, "xattn_out: C=%d heads=%d S=%d L=%d", d.C, d.heads, d.S, d.L);
  SD_REQUIRE(d.M % d.S == 0 && d.ldv % 8 == 0 && d.ldv >= d.L && d.ldv <= XO_KEYS, kInvalidArgument, "xattn_out: M=%d ldv=%d", d.M, d.ldv);
  if (d.impl == kAttnSplitEinsumV2 && d.S >= 512)   // the reference would silently drop the tail (attention.py:86)
    SD_REQUIRE(d.S % 512 == 0, kInvalidArgument, "SPLIT_EINSUM_V2 needs S_q %% 512 == 0 (got %d)", d.S);
  const bool pre = d.wo1_t != nullptr;
  SD_REQUIRE(!pre || (d.heads == 5 && d.h0 && d.o1_bias), kInvalidArgument, "xattn_out: the self-attention's to_out in front needs five heads, h0 and its bias");
  XOArgs a{d.x, reinterpret_cast<const half8*>(d.wq_t), d.q_bias, d.q_colsum, d.k, d.vt, reinterpret_cast<const half8*>(d.wo_t), d.o_bias,
           d.out, d.M, d.C, d.S, d.L, d.ldv, d.ln_eps, 1.4426950408889634f / sqrtf((float)XO_D), d.h0, reinterpret_cast<const half8*>(d.wo1_t),
           d.o1_bias};
  if (d.heads == 5 && pre) launch_nw<5, true>(a, s);
  else if (d.heads == 5) launch_nw<5, false>(a, s);
  else launch_nw<10, false>(a, s);
  SD_HIP(hipGetLastError());
}

bool ffn_proj_ok(int C, int K1, int M, int S) {
  return C == 320 && K1 == 4 * C && M >= XO_TOK && M % S == 0 && S % XO_TOK == 0;
}

int launch_ffn_proj(const FfnProjDesc& d, hipStream_t s) {
  SD_REQUIRE(ffn_proj_ok(d.C, d.K1, d.M, d.S) && d.g && d.w1_t && d.b1 && d.res1 && d.w2_t && d.b2 && d.res2 && d.out, kInvalidArgument,
             "ffn_proj: C=%d K1=%d M=%d S=%d", d.C, d.K1, d.M, d.S);
  const int tiles = d.S / XO_TOK;
  const bool stats = d.gn_partial != nullptr && d.gn_groups >= 1 && d.gn_groups <= 64 && d.C % d.gn_groups == 0 && tiles <= 128;
  FPArgs a{d.g, reinterpret_cast<const half8*>(d.w1_t), d.b1, d.res1, reinterpret_cast<const half8*>(d.w2_t), d.b2, d.res2, d.out,
           stats ? d.gn_partial : nullptr, d.M, d.S, d.gn_groups};
  constexpr size_t lds = fp_lds_bytes<5>();
  static_assert(lds <= 160 * 1024, "LDS");
  auto k = ffn_proj_kernel<5>;
  static DynLdsOnce once;
  once.set(k, lds);
  hipLaunchKernelGGL(k, dim3(d.M / XO_TOK), dim3(5 * 64), lds, s, a);
  SD_HIP(hipGetLastError());
  return stats ? tiles : 0;
}

bool gn_proj_qkv_ok(int C, int heads, int S, int M, int ldT, int G) {
  // (any head count: the kernel's column blocks are not heads - SD1.5's eight heads of 40 take the same launch)
  return C == 320 && heads >= 1 && C % heads == 0 && S >= XO_TOK && S % XO_TOK == 0 && M % S == 0 && ldT % 8 == 0 && ldT >= S && G >= 1 && G <= 32 && C % G == 0;
}

void launch_gn_proj_qkv(const GnProjQkvDesc& d, hipStream_t s) {
  SD_REQUIRE(gn_proj_qkv_ok(d.C, d.C / XO_D, d.S, d.M, d.ldT, d.gn_groups) && d.x && d.wp_t && d.p_bias && d.h && d.wqkv_t && d.qkv_bias &&
                 d.qkv_colsum && d.qk && d.vt && d.gn_entries >= 0 && d.gn_entries <= kGnMaxSlabs &&
                 (d.gn_entries == 0 || (d.gn_partial && d.gn_gamma && d.gn_beta)),
             kInvalidArgument, "gn_proj_qkv: C=%d S=%d M=%d ldT=%d entries=%d", d.C, d.S, d.M, d.ldT, d.gn_entries);
  GQArgs a{d.x, d.gn_partial, d.gn_gamma, d.gn_beta, reinterpret_cast<const half8*>(d.wp_t), d.p_bias, d.h,
           reinterpret_cast<const half8*>(d.wqkv_t), d.qkv_bias, d.qkv_colsum, d.qk, d.vt, d.M, d.S, d.ldT, d.vt_perm ? 1 : 0, d.gn_entries,
           d.gn_groups, d.gn_eps, d.ln_eps, d.q_scale, d.clk};
  // 64-token workgroups where they still give every CU one (from two prompts per GPU, M = 16 384): same box, 32 vs 64 tokens: one prompt
  // 4.303 vs 4.326 ms (worse: 128 workgroups), two 6.71 vs 6.69, four 10.74 vs 10.70, eight 18.40 vs 18.31.  SD_GQ_TOK = 32 / 64 (with
  // SD_TUNE): A/B
  static const int tok_env = tune_env_int("SD_GQ_TOK", 0);
  const int want = d.tok ? d.tok : tok_env;
  const bool two = want ? want == 64 : (d.M / 64 >= 256);
  if (two && d.S % 64 == 0) {
    constexpr size_t lds = gq_lds_bytes<5, 2>();
    auto k = gn_proj_qkv_kernel<5, 2>;
    static DynLdsOnce once;
    once.set(k, lds);
    hipLaunchKernelGGL(k, dim3(d.M / 64), dim3(5 * 64), lds, s, a);
  } else {
    constexpr size_t lds = gq_lds_bytes<5, 1>();
    auto k = gn_proj_qkv_kernel<5, 1>;
    static DynLdsOnce once;
    once.set(k, lds);
    hipLaunchKernelGGL(k, dim3(d.M / XO_TOK), dim3(5 * 64), lds, s, a);
  }
  SD_HIP(hipGetLastError());
}

void launch_xattn_out_retile_nk(const half_t* w, half_t* wt, int N, int K, hipStream_t s) {
  SD_REQUIRE(N % 32 == 0 && K % 16 == 0, kInvalidArgument, "fragment-major retile: N=%d K=%d", N, K);
  const int n = N * K / 8;
  hipLaunchKernelGGL(xo_retile_kernel, dim3((n + 255) / 256), dim3(256), 0, s, w, reinterpret_cast<half8*>(wt), N, K);
  SD_HIP(hipGetLastError());
}

}  // namespace sd
