// Large-M 1x1 GEMM with the WEIGHTS going global -> VGPR (unet.py:594-617 GEGLU / ff.net.2, :62-118 to_out).
//
// LAB_NOTES Finding 5: the tiled kernels of igemm.hip stop at 430-850 TFLOP/s because BOTH operands of a 128 x 128 x 64 step go
// through the L2 -> LDS fill of one CU (32 KB for 2.1 MFLOP = 64 FLOP per byte), while hipBLASLt reaches 1.0-1.26 PFLOP/s on the
// same shapes.  xattn_out.hip showed the way round it inside this library: weight fragments straight into registers in MFMA order
// from a pre-tiled copy, no LDS fill and no ring barrier on the weight side.  This kernel is that idea as a general GEMM:
//   * workgroup = NW waves, tile BM x (32 NW): wave w owns 32 output columns over all BM rows (BM / 32 accumulator blocks), so
//     nobody shares a weight fragment and the weights of a workgroup cross L2 -> CU exactly once;
//   * weights: per 64-deep K stage a wave requests its four 1-KB fragments (fully coalesced 16 B per lane) PB - 1 stages ahead
//     into a register ring - plain loads, counted by the compiler, nothing to synchronise;
//   * activations: the only operand in LDS.  Every thread carries 16-byte chunks of the BM x 64 stage through registers (requested
//     two stages ahead, written one stage ahead: the split issue-early / write-late form) into a two-slot ring with the bank
//     swizzle of igemm.hip; ONE barrier per stage; 256 FLOP per byte filled at BM = 128;
//   * LayerNorm fold: the loader threads see every activation chunk exactly once on its way to LDS and keep (sum, sumsq) of it -
//     the row statistics cost eight v_dot2 per stage and thread, no extra pass;
//   * epilogue through an LDS tile, whole rows out (bias, LayerNorm fold, residual, or value * gelu(gate) for GEGLU rows in the
//     lane order of wsgemm.hip).
// One workgroup of 8 waves per CU (2 per SIMD), like the 256 x 256 8-phase template of the CDNA guide; plain HIP, no inline asm.
#include "kernels.h"

#include <type_traits>

namespace sd {

namespace {

constexpr int BV_BK = 64;

struct BvArgs {
  const half_t* x;        // [M][K]
  const half_t* wt;       // pre-tiled: [N / 32 strips][K / 16][64 lanes][8 halves]  (launch_bvgemm_retile)
  const float* bias;      // [N] device row order, or null
  const float* colsum;    // [N] LayerNorm fold, or null
  const half_t* res;      // [M][ldo] or null (plain mode)
  half_t* out;            // [M][ldo]
  int M, N, K, nk, ldo;
  int mtiles, ntiles;
  float ln_eps;
  // fused q|k|v (unet.py:74-84 to_q / to_k / to_v as ONE GEMM): column tiles at or beyond n_trans leave token-transposed to
  // out_t [B][N - n_trans][ldT] (attention's V^T, optionally in attention8's key order), the others to out with row length ldo =
  // n_trans; columns below q_cols are multiplied by q_scale on the fp32 accumulator (ConvDesc::q_scale).  out_t == null: off.
  half_t* out_t;
  int n_trans, ldT, HoWo, vt_perm, q_cols;
  float q_scale;
};

__device__ __forceinline__ float bv_gelu_erf(float x) {   // igemm.hip gelu_erf (Abramowitz-Stegun 7.1.26)
  const float z = x * 0.70710678118654752f;
  const float az = fabsf(z);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, az, 1.0f));
  float p = 1.061405429f;
  p = fmaf(p, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __builtin_amdgcn_exp2f(-az * az * 1.4426950408889634f);
  const float erf_abs = fmaf(-p * t, e, 1.0f);
  return 0.5f * x * (1.0f + copysignf(erf_abs, z));
}

// MFMA row n of strip `strip` (the accumulator register of lane half hi = (n >> 2) & 1 that holds it is r = 4 (n >> 3) + (n & 3))
// -> row of the device weight matrix.  Plain: a lane ends up with the 16 consecutive output channels 32 strip + 16 hi + r.
// GEGLU: wsgemm.hip's order - registers 0-3 | 8-11 the values, 4-7 | 12-15 the gates of the 8 consecutive channels
// 16 strip + 8 hi + {0..3 | 4..7}; device rows are 32 values | 32 gates per 64 (UNet::upload_conv_weight).
__host__ __device__ inline int bv_row(int strip, int n, bool geglu) {
  const int hi = (n >> 2) & 1, grp = n >> 3, e = n & 3;
  if (!geglu) return strip * 32 + 16 * hi + 4 * grp + e;
  const int ch = strip * 16 + 8 * hi + 4 * (grp >> 1) + e;
  return (ch >> 5) * 64 + (grp & 1) * 32 + (ch & 31);
}

__global__ __launch_bounds__(256) void bvgemm_retile_kernel(const half_t* __restrict__ w, half_t* __restrict__ wt, int N, int K, int geglu) {
  const int ks16 = K / 16;
  const size_t total = (size_t)(N / 32) * ks16 * 64;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    const size_t r = idx >> 6;
    const int ks = (int)(r % ks16), strip = (int)(r / ks16);
    const int row = bv_row(strip, lane & 31, geglu != 0);
    *reinterpret_cast<half8*>(wt + idx * 8) = *reinterpret_cast<const half8*>(w + (size_t)row * K + ks * 16 + (lane >> 5) * 8);
  }
}

template <int BM, int NW, int TN, bool GEGLU>
struct BvLds {
  static constexpr int OCOLS = (GEGLU ? 16 : 32) * NW * TN;   // output columns of the workgroup
  static constexpr int OROW = OCOLS + 8;                      // staged row stride in halves
  static constexpr int SLOT = BM * BV_BK * 2;                 // one activation stage
  static constexpr int STAGE = BM * OROW * 2;                 // the epilogue's tile (re-uses the ring)
  static constexpr int TROW = BM + 8;                         // transposed staging (V^T tiles of the fused q|k|v): [OCOLS][TROW]
  static constexpr int TSTAGE = GEGLU ? 0 : OCOLS * TROW * 2;
  static constexpr int STG = STAGE > TSTAGE ? STAGE : TSTAGE;
  static constexpr int MAIN = (2 * SLOT > STG) ? 2 * SLOT : STG;
  static constexpr int STAT_OFF = MAIN;                       // [BM][2] floats
  static constexpr int BYTES = STAT_OFF + BM * 2 * 4;
};

// TN: 32-column blocks per wave (1: wave tile BM x 32; 2: BM x 64 - every activation fragment read from LDS meets two weight
// fragments, half the LDS traffic per MFMA, twice the accumulators); PB: weight stages in the register ring (requests run PB - 1
// stages ahead).  Occupancy: 8 waves per CU in every configuration (one workgroup of 8, or two of 4).
template <int BM, int NW, int TN, int PB, bool GEGLU, bool LNF>
__global__ __launch_bounds__(NW * 64, (BM == 64 && TN == 1 && NW == 8) ? 4 : (BM == 32 ? 1 : 2)) void bvgemm_kernel(BvArgs a) {
  using L = BvLds<BM, NW, TN, GEGLU>;
  constexpr int NT = NW * 64;
  constexpr int TM = BM / 32;                                  // accumulator row blocks per wave
  constexpr int CPT = BM * 8 / NT;                             // 16-byte activation chunks per thread and stage
  constexpr int NBF = 4 * TN;                                  // weight fragments per wave and stage
  static_assert(BM * 8 % NT == 0 && CPT >= 1, "whole chunks per thread");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  int bid = blockIdx.x;
  {   // XCD-contiguous walk: the n-tiles of a row block run on ONE XCD, its L2 serves the activations to all of them
    const int nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  const int nt = bid % a.ntiles, mt = bid / a.ntiles;
  const int m0 = mt * BM;
  const int strip0 = (nt * NW + wave) * TN;                    // the wave's first 32-row weight strip
  const int nk = a.nk, ks16 = a.K >> 4;

  // ---- loader coordinates: chunk j of this thread = (row, 16-byte column) of every stage ----
  const half_t* asrc[CPT];
  int adst[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) {
    const int c = tid + j * NT;
    const int row = c >> 3, ch = c & 7;
    const int mr = min(m0 + row, a.M - 1);                     // rows past the end re-read the last row (never stored)
    asrc[j] = a.x + (size_t)mr * a.K + ch * 8;
    adst[j] = row * 128 + ((ch ^ ((row >> 1) & 7)) * 16);
  }
  const half_t* wsrc = a.wt + ((size_t)strip0 * ks16 * 64 + lane) * 8;
  const size_t wstrip = (size_t)ks16 * 512;                    // halves between two strips

  constexpr int AD = (TN >= 2 || CPT >= 8) ? 1 : 2;   // activation register sets: 2 = requested two stages ahead; 1 (the 256-VGPR wave tile, 8-chunk loaders): one
  half8 areg[AD][CPT], breg[PB][NBF];
  float ls1[CPT], ls2[CPT];
#pragma unroll
  for (int j = 0; j < CPT; ++j) ls1[j] = ls2[j] = 0.f;
  auto load_a = [&](half8 (&dst)[CPT], int s) __attribute__((always_inline)) {
    const int sc = min(s, nk - 1);                             // (past the end: a redundant re-read, never used)
#pragma unroll
    for (int j = 0; j < CPT; ++j) dst[j] = *reinterpret_cast<const half8*>(asrc[j] + sc * BV_BK);
  };
  auto load_b = [&](half8 (&dst)[NBF], int s) __attribute__((always_inline)) {
    const int sc = min(s, nk - 1);
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) dst[j * 4 + q] = *reinterpret_cast<const half8*>(wsrc + j * wstrip + (size_t)(sc * 4 + q) * 512);
  };
  auto write_a = [&](const half8 (&src)[CPT], int slot, bool live) __attribute__((always_inline)) {   // live (block-uniform): a stage of the K range, not the clamped tail
    char* base = smem + slot * L::SLOT;
    const half2v one2 = {(half_t)1.f, (half_t)1.f};
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      *reinterpret_cast<half8*>(base + adst[j]) = src[j];
      if (LNF && live) {                                       // the chunk's share of its row's statistics, on the way through
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const half2v p2 = {src[j][2 * e], src[j][2 * e + 1]};
          ls2[j] = __builtin_amdgcn_fdot2(p2, p2, ls2[j], false);
          ls1[j] = __builtin_amdgcn_fdot2(p2, one2, ls1[j], false);
        }
      }
    }
  };

  const int fsw = (l31 >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) foff[q] = l31 * 128 + (((q * 2 + hi) ^ fsw) * 16);

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // fragment reads run one 16-deep step ahead of the MFMAs that use them (left to itself hipcc issues each pair of reads right
  // in front of its MFMAs: the LDS latency then sits between every two MFMAs of a wave)
  auto compute = [&](const half8 (&bq)[NBF], int slot) __attribute__((always_inline)) {
    const char* at = smem + slot * L::SLOT;
    half8 xf[2][TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) xf[0][i] = *reinterpret_cast<const half8*>(at + i * 4096 + foff[0]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (q + 1 < 4) {
#pragma unroll
        for (int i = 0; i < TM; ++i) xf[(q + 1) & 1][i] = *reinterpret_cast<const half8*>(at + i * 4096 + foff[q + 1]);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bq[j * 4 + q], xf[q & 1][i], acc[i][j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  // ---- prologue: the first activation stage(s) and PB - 1 weight stages in flight, stage 0 written ----
  load_a(areg[0], 0);
#pragma unroll
  for (int p = 0; p < PB - 1; ++p) load_b(breg[p], p);
  if constexpr (AD == 2) load_a(areg[1], 1);
  write_a(areg[0], 0, true);
  __syncthreads();

  // stage s: request B(s + PB - 1) and the next activation stage, multiply stage s, barrier; the activation stage s + 1 goes to
  // the other LDS slot in front of (AD == 2: it was requested a stage ago) or behind (AD == 1) the MFMAs.  Ring indices are
  // static: U stages per trip (U = lcm(2, PB)).
  auto stage = [&](int s, auto si_c) __attribute__((always_inline)) {
    constexpr int SI = decltype(si_c)::value;
    constexpr int AI = SI % 2, BI = SI % PB;
    load_b(breg[(BI + PB - 1) % PB], s + PB - 1);
    if constexpr (AD == 2) {
      write_a(areg[(AI + 1) % 2], (AI + 1) % 2, s + 1 < nk);
      load_a(areg[AI], s + 2);   // (this set's stage s went to LDS one stage ago)
      compute(breg[BI], AI);
    } else {
      load_a(areg[0], s + 1);
      compute(breg[BI], AI);
      write_a(areg[0], (AI + 1) % 2, s + 1 < nk);
    }
    __syncthreads();
  };
  constexpr int U = (PB % 2 == 0) ? PB : 2 * PB;
  auto trip = [&](int s, int count) __attribute__((always_inline)) {   // stages s .. s + count - 1 (count <= U, block-uniform)
    [&]<int... I>(std::integer_sequence<int, I...>) {
      ((I < count ? stage(s + I, std::integral_constant<int, I>{}) : void()), ...);
    }(std::make_integer_sequence<int, U>{});
  };
  int s = 0;
  for (; s + U <= nk; s += U) trip(s, U);
  if (s < nk) trip(s, nk - s);

  // ---- LayerNorm statistics: the 8 chunk owners of a row are 8 consecutive lanes ----
  float* stat = reinterpret_cast<float*>(smem + L::STAT_OFF);
  if constexpr (LNF) {
    // (write_a counted stages 0 .. nk - 1 of every chunk; the clamped re-read behind the last stage is written, not counted)
#pragma unroll
    for (int j = 0; j < CPT; ++j) {
      float s1 = ls1[j], s2 = ls2[j];
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) {
        s1 += __shfl_xor(s1, o);
        s2 += __shfl_xor(s2, o);
      }
      const int c = tid + j * NT;
      if ((c & 7) == 0) {
        const float inv_k = 1.0f / (float)a.K;
        const float mean = s1 * inv_k;
        const float var = fmaxf(s2 * inv_k - mean * mean, 0.f);
        const float rstd = rsqrtf(var + a.ln_eps);
        stat[(c >> 3) * 2] = rstd;
        stat[(c >> 3) * 2 + 1] = -rstd * mean;
      }
    }
  }

  // ---- epilogue: per column block the constants of the lane's 16 accumulator rows, tile -> LDS, whole rows -> global ----
  half_t* sg = reinterpret_cast<half_t*>(smem);
  const bool tblock = !GEGLU && a.out_t != nullptr && nt * L::OCOLS >= a.n_trans;   // block-uniform: a V^T tile of the fused q|k|v
  if constexpr (LNF) __syncthreads();                          // statistics visible (the ring is free since the last stage's barrier)
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    float cb[16], cs[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = bv_row(strip0 + j, (r & 3) + 8 * (r >> 2) + 4 * hi, GEGLU);
      cb[r] = a.bias ? a.bias[row] : 0.f;
      cs[r] = LNF ? a.colsum[row] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int row = i * 32 + l31;
      float la = 1.f, lb = 0.f;
      if constexpr (LNF) {
        la = stat[row * 2];
        lb = stat[row * 2 + 1];
      }
      if constexpr (GEGLU) {
        half8 o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int rv = (e < 4) ? e : 8 + (e - 4), rg = rv + 4;
          float v, g;
          if constexpr (LNF) {
            v = fmaf(acc[i][j][rv], la, fmaf(lb, cs[rv], cb[rv]));
            g = fmaf(acc[i][j][rg], la, fmaf(lb, cs[rg], cb[rg]));
          } else {
            v = acc[i][j][rv] + cb[rv];
            g = acc[i][j][rg] + cb[rg];
          }
          o[e] = (half_t)(v * bv_gelu_erf(g));
        }
        *reinterpret_cast<half8*>(sg + row * L::OROW + (wave * TN + j) * 16 + hi * 8) = o;
      } else {
        const bool qcol = (strip0 + j) * 32 < a.q_cols;          // wave-uniform: a query strip of the fused q|k|v
        half8 o[2];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v;
          if constexpr (LNF) v = fmaf(acc[i][j][r], la, fmaf(lb, cs[r], cb[r]));
          else v = acc[i][j][r] + cb[r];
          if (qcol) v *= a.q_scale;
          o[r >> 3][r & 7] = (half_t)v;
        }
        if (tblock) {   // V^T tile: staged [column][token] so that the write-out rows are token-contiguous
          half_t* tg = sg + ((wave * TN + j) * 32 + hi * 16) * L::TROW + row;
#pragma unroll
          for (int r = 0; r < 16; ++r) tg[r * L::TROW] = o[r >> 3][r & 7];
        } else {
          *reinterpret_cast<half8*>(sg + row * L::OROW + (wave * TN + j) * 32 + hi * 16) = o[0];
          *reinterpret_cast<half8*>(sg + row * L::OROW + (wave * TN + j) * 32 + hi * 16 + 8) = o[1];
        }
      }
    }
  }
  __syncthreads();
  constexpr int CPR = L::OCOLS / 8;                            // 16-byte chunks per staged row
  constexpr int ROUNDS = BM * CPR / NT;
  static_assert(BM * CPR % NT == 0, "whole store rounds");
  const size_t col0 = (size_t)nt * L::OCOLS;
  if constexpr (!GEGLU) {
    if (tblock) {   // out_t[b][n - n_trans][s]: 8 tokens of one image per 16-byte store (the row tile lies inside one image)
      const int NV = a.N - a.n_trans, nv0 = (int)col0 - a.n_trans;
      const int b = m0 / a.HoWo, sp0 = m0 - b * a.HoWo;
      for (int id = tid; id < L::OCOLS * (BM / 8); id += NT) {
        const int r = id / (BM / 8), c = id - r * (BM / 8);
        if (m0 + c * 8 < a.M) {
          half8 v;
          if (a.vt_perm) {   // chunk c = tokens 16 j + 4 o + {0..3} and 16 j + 8 + 4 o + {0..3}  (j = c >> 1, o = c & 1): AttnDesc::vt_perm
            const half_t* src = sg + r * L::TROW + (c >> 1) * 16 + (c & 1) * 4;
            const half4 lo = *reinterpret_cast<const half4*>(src), up = *reinterpret_cast<const half4*>(src + 8);
            v = half8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
          } else {
            v = *reinterpret_cast<const half8*>(sg + r * L::TROW + c * 8);
          }
          out_store(reinterpret_cast<half8*>(a.out_t + ((size_t)b * NV + nv0 + r) * a.ldT + sp0 + c * 8), v);
        }
      }
      return;
    }
  }
#pragma unroll 4
  for (int it = 0; it < ROUNDS; ++it) {
    const int id = tid + it * NT;
    const int r = id / CPR, c = id - r * CPR;
    half8 v = *reinterpret_cast<const half8*>(sg + r * L::OROW + c * 8);
    if (m0 + r < a.M) {
      const size_t off = (size_t)(m0 + r) * a.ldo + col0 + c * 8;
      if (!GEGLU && a.res) {
        const half8 rr = *reinterpret_cast<const half8*>(a.res + off);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)rr[e]);
      }
      out_store(reinterpret_cast<half8*>(a.out + off), v);
    }
  }
}

template <int BM, int NW, int TN, int PB, bool GEGLU, bool LNF>
void launch_bv(const BvArgs& a, hipStream_t s) {
  auto k = bvgemm_kernel<BM, NW, TN, PB, GEGLU, LNF>;
  constexpr size_t lds = BvLds<BM, NW, TN, GEGLU>::BYTES;
  static DynLdsOnce once;
  once.set(k, lds);
  hipLaunchKernelGGL(k, dim3(a.mtiles * a.ntiles), dim3(NW * 64), lds, s, a);
}

}  // namespace

// single-source 1x1 GEMM, K a multiple of 64 (>= 256), N a multiple of 128 weight rows, bias / LayerNorm fold / residual or GEGLU;
// no timestep embedding, no fused q|k|v, no GroupNorm statistics of the output, no split-K.
bool bvgemm_shape_ok(const ConvDesc& d) {
  if (d.ksize != 1 || d.stride != 1 || d.up != 1 || d.x1 || d.C0 % 64 != 0 || d.C0 < 256) return false;
  if (d.temb || d.gn_partial || d.gnf_partial || d.n_twins || d.debug) return false;
  if (!(d.out_mode == kOutHalf || d.out_mode == kOutGeglu)) return false;
  if (d.out_mode == kOutGeglu && d.res) return false;
  if (d.out_t) {   // fused q|k|v: whole 128-column tiles on either side of the boundary, row tiles inside one image
    if (d.out_mode != kOutHalf || d.res || d.n_trans <= 0 || d.n_trans % 128 != 0 || d.N % 128 != 0 || (d.Ho * d.Wo) % 128 != 0 ||
        d.ldT % 8 != 0 || d.q_cols % 32 != 0)
      return false;
  }
  if (d.N % 64 != 0) return false;
  return (long)d.B * d.Ho * d.Wo >= 128;
}

size_t bvgemm_tiled_halves(int N, int K) { return (size_t)N * K; }

void launch_bvgemm_retile(const half_t* w, half_t* wt, int N, int K, bool geglu, hipStream_t s) {
  SD_REQUIRE(N % 32 == 0 && K % 16 == 0, kInvalidArgument, "bvgemm retile: N=%d K=%d", N, K);
  const size_t total = (size_t)N * K / 8;
  hipLaunchKernelGGL(bvgemm_retile_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, w, wt, N, K, geglu ? 1 : 0);
  SD_HIP(hipGetLastError());
}

// variant: 1: 64 rows x 8 waves x 32 columns; 2: 128 rows x 8 waves x 32 columns; 3: 128 rows x 4 waves x 64 columns (two
// workgroups per CU); 4: 128 rows x 4 waves x 32 columns (128-column tiles: N % 256 != 0); 5: 32 rows x 2 waves x 32 columns (many
// small workgroups: the small-M experiment - ties the tiled kernels warm, loses cold); 6: 128 rows x 2 waves x 32 columns (64-column
// tiles for N = 320 / 960 at large M); 0 = chosen here (64 / 32 rows x 4 waves x 32 columns were also measured, on SDXL's 1 152-row
// level with cold operands: 19.5 / 20.4 us against 19.1 tiled on 1280->1280, 51.1 / 53.6 against 45.8 on 5120->1280 - removed) from the stand-alone table profiles/r06_bvgemm_bench.txt
int bvgemm_auto_variant(const ConvDesc& d) {
  const long M = (long)d.B * d.Ho * d.Wo;
  if (d.N % 128 != 0) return M >= 4096 ? 6 : 5;
  if (d.N % 256 != 0 || (d.out_t && d.n_trans % 256 != 0)) return 4;
  if (d.out_mode == kOutGeglu) return (d.C0 >= 1280 || M < 16384) ? 2 : 3;
  if (M >= 16384) return 3;
  return 1;
}

void launch_bvgemm(const ConvDesc& d, int variant, hipStream_t s) {
  SD_REQUIRE(bvgemm_shape_ok(d) && d.w_bv, kInvalidArgument, "bvgemm: shape not eligible (C0=%d N=%d mode=%d)", d.C0, d.N, d.out_mode);
  if (variant < 1 || variant > 6) variant = bvgemm_auto_variant(d);
  const int tcols = (variant == 5 || variant == 6) ? 64 : (variant == 4 ? 128 : 256);   // columns per workgroup
  SD_REQUIRE(d.N % tcols == 0, kInvalidArgument, "bvgemm variant %d does not tile N=%d", variant, d.N);
  SD_REQUIRE(!d.out_t || (tcols >= 128 && d.n_trans % tcols == 0), kInvalidArgument,
             "bvgemm variant %d: the q|k / v boundary %d is not a tile boundary", variant, d.n_trans);
  BvArgs a{};
  a.x = d.x0;
  a.wt = d.w_bv;
  a.bias = d.bias;
  a.colsum = d.ln_colsum;
  a.res = d.res;
  a.out = d.out;
  a.M = d.B * d.Ho * d.Wo;
  a.N = d.N;
  a.K = d.C0;
  a.nk = a.K / BV_BK;
  const bool geglu = d.out_mode == kOutGeglu;
  a.ldo = geglu ? d.N / 2 : (d.out_t ? d.n_trans : d.N);
  a.out_t = d.out_t;
  a.n_trans = d.out_t ? d.n_trans : 0x7fffffff;
  a.ldT = d.ldT;
  a.HoWo = d.Ho * d.Wo;
  a.vt_perm = d.out_t ? d.vt_perm : 0;
  a.q_cols = d.out_t ? d.q_cols : 0;
  a.q_scale = d.q_scale;
  a.ntiles = d.N / tcols;
  a.ln_eps = d.ln_eps;
  const int bm = variant == 5 ? 32 : (variant == 1 ? 64 : 128);
  a.mtiles = cdiv(a.M, bm);
  const bool lnf = d.ln_colsum != nullptr;
#define SD_BV(BM_, NW_, TN_, PB_)                                     \
  do {                                                                \
    if (geglu) {                                                      \
      if (lnf) launch_bv<BM_, NW_, TN_, PB_, true, true>(a, s);       \
      else launch_bv<BM_, NW_, TN_, PB_, true, false>(a, s);          \
    } else {                                                          \
      if (lnf) launch_bv<BM_, NW_, TN_, PB_, false, true>(a, s);      \
      else launch_bv<BM_, NW_, TN_, PB_, false, false>(a, s);         \
    }                                                                 \
  } while (0)
  if (variant == 1) SD_BV(64, 8, 1, 4);
  else if (variant == 2) SD_BV(128, 8, 1, 4);
  else if (variant == 3) SD_BV(128, 4, 2, 2);
  else if (variant == 4) SD_BV(128, 4, 1, 4);
  else if (variant == 5) SD_BV(32, 2, 1, 4);
  else SD_BV(128, 2, 1, 2);
#undef SD_BV
  SD_HIP(hipGetLastError());
}

// The library's own rule (launch_conv / UNet::conv_w): where the stand-alone table shows this kernel ahead of the tiled ones -
// 8 192 rows and more, K of at least 640 (K = 320 GEGLU belongs to wsgemm.hip)
bool bvgemm_wanted(const ConvDesc& d) {
  if (!bvgemm_shape_ok(d)) return false;
  const long M = (long)d.B * d.Ho * d.Wo;
  static const bool narrow = tune_env_int("SD_BVGEMM_NARROW", 1) != 0;   // 64-column tiles (N = 320 / 960 ...): A/B
  if (d.N % 128 != 0 && !narrow) return false;
  static const bool qkv = tune_env_int("SD_BVGEMM_QKV", 1) != 0;         // the fused q|k|v epilogue: A/B
  if (d.out_t && !qkv) return false;
  static const int min_m = tune_env_int("SD_BVGEMM_MIN_M", 4096), min_k = tune_env_int("SD_BVGEMM_MIN_K", 640);   // thresholds: A/B
  return M >= min_m && d.C0 >= min_k;
}

}  // namespace sd
