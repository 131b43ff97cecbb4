// Cross-attention front half as ONE launch (CrossAttention.forward with a text context, unet.py:87-118, inside
// BasicTransformerBlock.forward unet.py:586-591):
//     q   = to_q(norm2(h1))                      LayerNormANE (layer_norm.py:51-80) folded into the projection
//     a2  = softmax(q k^T / sqrt(d)) v           per head, k / v = to_k / to_v of the prompt (hoisted: they change with the
//                                                prompt, not with the step)
// One workgroup = 128 query tokens x one head.  Phase 1 is the LayerNorm-folded GEMM of igemm.hip restricted to the 64
// output channels of this head (operands HBM -> LDS by LDS-DMA through a ring, row statistics of the A rows by
// v_dot2_f32_f16 between the MFMAs); the finished q tile never leaves the registers: the transposed MFMA layout
// (rows = channel, cols = token) puts the 64 channels of one query into one lane, which is the B operand of the score MFMA
// K . Q^T up to a permutation of the head channels - and a dot product does not care about the order of its terms, so K is
// gathered from LDS with the same permutation.  Phase 2: the <= 96 keys of the prompt (77) fit ONE key tile set: exact
// softmax over keys in registers (lane = query; attention.py:24-72 "bkhq" orientation), no running maximum, P feeds
// P . V straight from registers.  Replaces a GEMM launch + an attention launch and the q round trip through HBM; the
// three schedules of attention.py differ in how they walk LONG key axes and coincide for a single key tile, so this kernel
// serves all three (the 512-query chunks of SPLIT_EINSUM_V2 are a multiple of its 128-query workgroups).  A sample whose token
// count is not a multiple of 128 (SDXL's 24x24 level: 576) ends in a ragged tile whose surplus rows are computed and dropped.
#include "kernels.h"

namespace sd {
namespace {

constexpr int BK = 64;        // K step of the projection (halves)
constexpr int XBM = 128;      // query tokens per workgroup
constexpr int XD = 64;        // head dim
constexpr int XKEYS = 96;     // key capacity (3 MFMA tiles)
constexpr int KROW = XD + 4;  // LDS row strides in halves: 136 B / 200 B keep the 8-byte fragment gathers conflict-free
constexpr int VROW = XKEYS + 4;

struct XAttnArgs {
  const half_t* x;
  const half_t* wq;
  const float* bias;
  const float* colsum;
  const half_t* k;
  const half_t* vt;
  half_t* out;
  int M, C, S, L, ldv, heads;
  float ln_eps, scale_log2;
};

__device__ __forceinline__ float xor32_sumf(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return __uint_as_float(r0) + __uint_as_float(r1);
}
__device__ __forceinline__ float xor32_maxf(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return fmaxf(__uint_as_float(r0), __uint_as_float(r1));
}

constexpr size_t xattn_lds_bytes(int nst) {
  return (size_t)nst * (XBM + XD) * BK * 2 + (size_t)XKEYS * KROW * 2 + (size_t)XD * VROW * 2 + 2 * XD * sizeof(float);
}

template <int NST>
__global__ __launch_bounds__(256, xattn_lds_bytes(NST) <= 80 * 1024 ? 2 : 1) void xattn_fused_kernel(XAttnArgs a) {
  constexpr int XR = XBM / 32, WR = XD / 32;   // 1-KiB DMA pieces per wave per K step (activations, weights)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* Xs = reinterpret_cast<half_t*>(smem);          // [NST][XBM][BK]   swizzled 128-B rows
  half_t* Ws = Xs + NST * XBM * BK;                      // [NST][XD][BK]
  half_t* Ks = Ws + NST * XD * BK;                       // [XKEYS][KROW]
  half_t* Vs = Ks + XKEYS * KROW;                        // [XD][VROW]
  float* sconst = reinterpret_cast<float*>(Vs + XD * VROW);   // [XD] bias | [XD] colsum of this head

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;

  // XCD-aware order (block b runs on XCD b % 8): each XCD walks a contiguous run of (m-tile, head) pairs, heads fastest, so
  // the heads of one m-tile - which share its 128 x C activation rows - meet in one L2
  const int tps = (a.S + XBM - 1) / XBM;                 // query tiles per sample; the last one may be ragged (SDXL: 576 = 4.5 x 128)
  const int nwg = (a.M / a.S) * tps * a.heads;
  int bid = blockIdx.x;
  {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int mt = bid / a.heads, h = bid - mt * a.heads;
  const int b = mt / tps;                                // one sample per workgroup (its K / V)
  const int m_blk = b * a.S + (mt - b * tps) * XBM;
  const int rows_valid = min(XBM, a.S - (mt - b * tps) * XBM);   // rows past it belong to the next sample: computed, never stored
  const int nk = a.C / BK;

  // ---- prompt K / V^T of this head: requested first (oldest VMEM ops), written to LDS after the projection loop ----
  const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
  half8 kreg[3], vreg[3];
  {
    const half_t* kb = a.k + (size_t)b * a.L * a.C + (size_t)h * XD;
    const half_t* vb = a.vt + ((size_t)b * a.C + (size_t)h * XD) * a.ldv;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int idx = tid + i * 256;                     // 96 rows x 8 chunks
      const int row = idx >> 3, ch = idx & 7;
      kreg[i] = (row < a.L) ? *reinterpret_cast<const half8*>(kb + (size_t)row * a.C + ch * 8) : zero8;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int idx = tid + i * 256;                     // 64 rows x 12 chunks of 8 keys
      const int row = idx / 12, ch = idx - row * 12;
      // contract: columns [L, ldv) of vt are zero (the UNet zero-fills them at create time)
      vreg[i] = (ch * 8 < a.ldv) ? *reinterpret_cast<const half8*>(vb + (size_t)row * a.ldv + ch * 8) : zero8;
    }
  }
  if (tid < XD) {
    sconst[tid] = a.bias[h * XD + tid];
    sconst[XD + tid] = a.colsum[h * XD + tid];
  }

  // ---- phase 1: q^T[64][128] = Wq_h . x^T, LayerNorm statistics of the x rows on the side ----
  const int pchunk = tid & 7, lrow = tid >> 3;           // this thread's 16-B slot / row inside a 1-KiB piece
  const int chunk = pchunk ^ ((lrow >> 1) & 7);          // swizzle on the SOURCE address (the DMA writes lane-linear)
  const half_t* xp[XR];
  const half_t* wp[WR];
#pragma unroll
  for (int i = 0; i < XR; ++i) xp[i] = a.x + (size_t)min(m_blk + lrow + 32 * i, a.M - 1) * a.C + chunk * 8;   // clamped past the tensor
#pragma unroll
  for (int i = 0; i < WR; ++i) wp[i] = a.wq + (size_t)(h * XD + lrow + 32 * i) * a.C + chunk * 8;
  auto load_tile = [&](int stage) {
    char* xs = reinterpret_cast<char*>(Xs + stage * XBM * BK) + wave * 1024;
    char* ws = reinterpret_cast<char*>(Ws + stage * XD * BK) + wave * 1024;
#pragma unroll
    for (int i = 0; i < XR; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)xp[i],
                                       (__attribute__((address_space(3))) void*)(xs + i * 4096), 16, 0, 0);
      xp[i] += BK;
    }
#pragma unroll
    for (int i = 0; i < WR; ++i) {
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wp[i],
                                       (__attribute__((address_space(3))) void*)(ws + i * 4096), 16, 0, 0);
      wp[i] += BK;
    }
  };

  floatx16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float ln_s1 = 0.f, ln_s2 = 0.f;
  const int fsw = (l31 >> 1) & 7;

  constexpr int PER_TILE = XR + WR;
  constexpr int DEPTH = NST - 2;                         // tiles still in flight while tile kt is computed
#pragma unroll
  for (int p = 0; p < NST - 1; ++p)
    if (p < nk) load_tile(p);
  for (int kt = 0; kt < nk; ++kt) {
    const int ahead = nk - 1 - kt;
    const int fly = ahead < DEPTH ? ahead : DEPTH;
    // tile kt has landed for every wave (loads of one wave return in order) and every wave has finished reading the
    // stage the next DMA overwrites; counted wait + raw barrier in one statement (a __syncthreads would drain vmcnt(0))
    if (DEPTH >= 3 && fly == 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((DEPTH >= 3 ? 3 : 0) * PER_TILE) : "memory");
    else if (DEPTH >= 2 && fly == 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((DEPTH >= 2 ? 2 : 0) * PER_TILE) : "memory");
    else if (DEPTH >= 1 && fly == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((DEPTH >= 1 ? 1 : 0) * PER_TILE) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    const int st = kt % NST;
    const half_t* xs = Xs + st * XBM * BK + (wave * 32 + l31) * BK;
    const half_t* ws = Ws + st * XD * BK + l31 * BK;
    half8 xf[4], wf[4][2];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int koff = ((kk * 2 + hi) ^ fsw) * 8;
      xf[kk] = *reinterpret_cast<const half8*>(xs + koff);
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[kk][j] = *reinterpret_cast<const half8*>(ws + j * 32 * BK + koff);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kt + NST - 1 < nk) load_tile((kt + NST - 1) % NST);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
      for (int j = 0; j < 2; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk][j], xf[kk], acc[j], 0, 0, 0);
      const half2v one2 = {(half_t)1.f, (half_t)1.f};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const half2v p2 = {xf[kk][2 * e], xf[kk][2 * e + 1]};
        ln_s2 = __builtin_amdgcn_fdot2(p2, p2, ln_s2, false);
        ln_s1 = __builtin_amdgcn_fdot2(p2, one2, ln_s1, false);
      }
    }
  }

  // ---- K / V^T of the prompt into LDS (their loads landed long ago) ----
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = tid + i * 256;
    const int row = idx >> 3, ch = idx & 7;
    half_t* dst = Ks + row * KROW + ch * 8;              // 136-B rows are only 8-B aligned
    *reinterpret_cast<half4*>(dst) = half4{kreg[i][0], kreg[i][1], kreg[i][2], kreg[i][3]};
    *reinterpret_cast<half4*>(dst + 4) = half4{kreg[i][4], kreg[i][5], kreg[i][6], kreg[i][7]};
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int idx = tid + i * 256;
    const int row = idx / 12, ch = idx - row * 12;
    half_t* dst = Vs + row * VROW + ch * 8;
    *reinterpret_cast<half4*>(dst) = half4{vreg[i][0], vreg[i][1], vreg[i][2], vreg[i][3]};
    *reinterpret_cast<half4*>(dst + 4) = half4{vreg[i][4], vreg[i][5], vreg[i][6], vreg[i][7]};
  }
  __syncthreads();                                       // K / V^T and sconst visible

  // ---- q of this lane's token: LayerNorm fold (igemm.hip LNF epilogue), fp16, MFMA B-operand order ----
  // acc[j][r]: channel n = j*32 + (r&3) + 8*(r>>2) + 4*hi of token (lane & 31) -> k-slot e of step (j, s) is r = 8*s + e
  half8 qf[2][2];
  {
    const float inv_k = 1.0f / (float)a.C;
    const float s1 = xor32_sumf(ln_s1), s2 = xor32_sumf(ln_s2);   // the two k halves of the wave
    const float mean = s1 * inv_k;
    const float var = fmaxf(s2 * inv_k - mean * mean, 0.f);
    const float ln_a = rsqrtf(var + a.ln_eps);
    const float ln_b = -ln_a * mean;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {
        const int nl = j * 32 + 8 * q4 + 4 * hi;
        const floatx4 bb = *reinterpret_cast<const floatx4*>(sconst + nl);
        const floatx4 cs = *reinterpret_cast<const floatx4*>(sconst + XD + nl);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * q4 + e;
          qf[j][r >> 3][r & 7] = (half_t)fmaf(acc[j][r], ln_a, fmaf(ln_b, cs[e], bb[e]));
        }
      }
  }

  // ---- phase 2: scores^T[key][q] = K . Q^T over the permuted head channels, exact softmax over keys ----
  floatx16 sacc[3];
#pragma unroll
  for (int kt = 0; kt < 3; ++kt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) sacc[kt][r] = 0.f;
    const half_t* kr = Ks + (kt * 32 + l31) * KROW + 4 * hi;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const half4 k0 = *reinterpret_cast<const half4*>(kr + j * 32 + 16 * s);
        const half4 k1 = *reinterpret_cast<const half4*>(kr + j * 32 + 16 * s + 8);
        const half8 kf = {k0[0], k0[1], k0[2], k0[3], k1[0], k1[1], k1[2], k1[3]};
        sacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[j][s], sacc[kt], 0, 0, 0);
      }
  }
  float mx = -3.0e38f;
#pragma unroll
  for (int kt = 0; kt < 3; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      if (key >= a.L) sacc[kt][r] = -3.0e38f;
      mx = fmaxf(mx, sacc[kt][r]);
    }
  mx = xor32_maxf(mx);                                   // the other half-wave holds the other keys of each tile
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
  const float inv = 1.0f / xor32_sumf((ps4[0] + ps4[1]) + (ps4[2] + ps4[3]));

  // O^T[d][q] = V^T . P^T: k-slot (hi, e) of step (kt, s2) holds key kt*32 + s2*16 + (e&3) + 8*(e>>2) + 4*hi
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
        const half_t* vp = Vs + (ct * 32 + l31) * VROW + kt * 32 + s2 * 16 + 4 * hi;
        const half4 v0 = *reinterpret_cast<const half4*>(vp);
        const half4 v1 = *reinterpret_cast<const half4*>(vp + 8);
        const half8 vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oacc[ct], 0, 0, 0);
      }
    }

  // ---- normalise + store: oacc[ct][r] = channel ct*32 + (r&3) + 8*(r>>2) + 4*hi of token (lane & 31) ----
  if (wave * 32 + l31 >= rows_valid) return;             // ragged last tile of a sample
  half_t* orow = a.out + (size_t)(m_blk + wave * 32 + l31) * a.C + (size_t)h * XD;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const half4 o = {(half_t)(oacc[ct][4 * g] * inv), (half_t)(oacc[ct][4 * g + 1] * inv),
                       (half_t)(oacc[ct][4 * g + 2] * inv), (half_t)(oacc[ct][4 * g + 3] * inv)};
      *reinterpret_cast<half4*>(orow + ct * 32 + 8 * g + 4 * hi) = o;
    }
}

template <int NST>
void launch_nst(const XAttnArgs& a, hipStream_t s) {
  constexpr size_t lds = xattn_lds_bytes(NST);
  static_assert(lds <= 160 * 1024, "LDS");
  auto k = xattn_fused_kernel<NST>;
  static DynLdsOnce once;
  once.set(k, lds);
  hipLaunchKernelGGL(k, dim3((a.M / a.S) * ((a.S + XBM - 1) / XBM) * a.heads), dim3(256), lds, s, a);
}

}  // namespace

bool xattn_fused_ok(int C, int heads, int S, int L) {
  static const bool off = tune_env_set("SD_NO_XATTN_FUSED");   // A/B switch
  return !off && heads >= 1 && C == heads * XD && C % BK == 0 && S >= 1 && L >= 1 && L <= XKEYS;
}

void launch_xattn_fused(const XAttnDesc& d, hipStream_t s) {
  SD_REQUIRE(xattn_fused_ok(d.C, d.heads, d.S, d.L), kInvalidArgument, "xattn_fused: C=%d heads=%d S=%d L=%d", d.C, d.heads, d.S,
             d.L);
  SD_REQUIRE(d.M % d.S == 0 && d.ldv % 8 == 0 && d.ldv >= d.L && d.ldv <= XKEYS, kInvalidArgument, "xattn_fused: M=%d ldv=%d", d.M,
             d.ldv);
  if (d.impl == kAttnSplitEinsumV2 && d.S >= 512)   // the reference would silently drop the tail (attention.py:86)
    SD_REQUIRE(d.S % 512 == 0, kInvalidArgument, "SPLIT_EINSUM_V2 needs S_q %% 512 == 0 (got %d)", d.S);
  XAttnArgs a{d.x, d.wq, d.bias, d.colsum, d.k, d.vt, d.out, d.M, d.C, d.S, d.L, d.ldv, d.heads, d.ln_eps,
              1.4426950408889634f / sqrtf((float)XD)};
  // ring depth: 2 stages keep two workgroups on a CU (75 KB each), 3 / 4 stages keep more of the weight panel in flight for
  // the deep-K levels; SD_XATTN_NST overrides (tuning)
  static const int forced = tune_env_int("SD_XATTN_NST", 0);
  const int nst = forced ? forced : (d.nst ? d.nst : (d.C <= 640 ? 2 : 3));
  if (nst >= 5) launch_nst<5>(a, s);
  else if (nst == 4) launch_nst<4>(a, s);
  else if (nst == 3) launch_nst<3>(a, s);
  else launch_nst<2>(a, s);
  SD_HIP(hipGetLastError());
}

}  // namespace sd
