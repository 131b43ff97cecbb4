// K1-K3, head dim 64 - software-pipelined streaming-softmax attention for the UNet's self-attention shapes
// (attention.py:24-168; unet.py:87-118 with S_k = S_q = 4096 / 1024 / 256 at SD2.1-base 512x512).
//
// The general kernels of attention.hip run one wave per SIMD through a serial chain per 64-key tile (K / V^T staged
// HBM -> VGPR -> LDS, barrier, Q.K^T, softmax, P.V): ~2250 cycles per tile for 512 cycles of MFMA (13.7 % MFMA-busy,
// profiles/r03_final_hbm_traffic.json).  This kernel is built the other way round, for d = 64 and S_k % 64 == 0:
//   * 8 (or 4) waves per workgroup, TWO waves per SIMD (MI355X_MICROARCH.md "Two waves per SIMD"), 32 queries per wave;
//   * K and V^T tiles go HBM -> LDS by LDS-DMA (buffer_load ... lds, no VGPR round trip, no ds_write pass) through a ring
//     of D stages behind ONE raw s_barrier per tile and counted vmcnt waits; the bank swizzle lives on the DMA's source
//     address (physical 16-B chunk p of row r holds logical chunk p ^ ((r >> 1) & 7), as in igemm.hip);
//   * the score tile is key-major ([k][q] = K.Q^T, the reference's "bkhq" of attention.py:50): a lane owns ONE query
//     column, so the softmax reductions run over registers (one half-wave exchange per tile) and P feeds the P.V MFMA
//     straight from registers; V^T is produced in the key order those registers have (AttnDesc::vt_perm);
//   * the loop is software-pipelined across tiles: while the VALU exponentiates tile j, the matrix pipe already
//     computes the scores of tile j+1 (second score accumulator), and the row max of tile j+1 is taken in the shadow of
//     P.V of tile j - every phase pairs 4-8 MFMAs with 20-60 independent VALU instructions (sched_group_barrier), so the
//     two co-resident waves of a SIMD fill each other's pipes whatever their relative phase.
// All three reference schedules of these shapes run here (the score tile never leaves registers, so "bhqk" vs "bkhq" is a
// choice of register layout, not of arithmetic): the running max is refreshed lazily - only when a tile exceeds it by more
// than 2^8; P <= 256 is fp16-safe and the softmax is invariant to the stabiliser (a refresh on every tile where any row's
// max moved, the SPLIT_EINSUM kernels' rule, measured 89 vs 70 us on the L0 shape: the refresh is 80 VALU instructions);
// SPLIT_EINSUM_V2's 512-query chunks are two 256-query workgroups.  fp32 running max / sum / accumulators, fp16 operands.
#include "kernels.h"

#include <cstdlib>
#include <type_traits>

namespace sd {
namespace {

constexpr int KT = 64;                       // keys per tile
constexpr int TILE_BYTES = KT * 64 * 2;      // one K (or V^T) tile: 64 rows x 128 B
constexpr int STAGE_BYTES = 2 * TILE_BYTES;  // K | V^T

struct Attn8Args {
  const half_t* q;
  const half_t* k;
  const half_t* vt;
  half_t* out;
  int heads, Sq, Sk;
  int ldq, ldk, ldv, ldo;
  float scale_log2;   // d^-0.5 * log2(e); 1 when q arrives pre-scaled (AttnDesc::q_prescaled)
  int q_tiles;        // query tiles (WAVES * 32 queries) per (sample, head)
  int prescaled;
};

__device__ __forceinline__ void dma16(const __amdgpu_buffer_rsrc_t& rs, char* lds, unsigned voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voffset, soffset, 0, 0);
#endif
}
template <int N>
__device__ __forceinline__ void wait_vmcnt_barrier() {   // counted wait + raw barrier in one statement: no LDS access moves across
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
__device__ __forceinline__ float xor32_max(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return fmaxf(__uint_as_float(r0), __uint_as_float(r1));
}
__device__ __forceinline__ float xor32_sum(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];
  return __uint_as_float(r0) + __uint_as_float(r1);
}
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }   // one v_max3_f32

// EXACT (kept for A/B, SD_ATTN8_EXACT=1): running max refreshed whenever it moved instead of the lazy refresh.
//
// VALU diet.  Measured on the first version of this kernel (profiles/r04_attn8_ablation_v1.txt): the loop was bound by the
// VECTOR ALU's instruction stream - 186 VALU instructions per key tile and wave against 16 MFMAs, two waves per SIMD - not
// by the matrix pipe (37 % busy), the LDS or the DMA.  So everything that can leave the VALU has left it:
//   * the softmax scale is multiplied into Q once (fp16, at kernel entry) and the running max enters the scores as the C
//     operand of the first Q.K^T MFMA (a 16-register block holding -m that only changes when the max moves): the scores
//     come out of the matrix pipe as s*scale - m, ready for v_exp_f32 - no per-score FMA;
//   * the row sums are v_dot2_f32_f16 over the PACKED fp16 probabilities (one instruction per two keys, and the
//     normaliser is the sum of exactly the values P.V multiplies);
//   * V^T arrives in the key order the P registers already have (AttnDesc::vt_perm), so a V^T fragment is one 16-byte
//     LDS read at the SAME per-lane offsets as a K fragment: no half-fragment moves, four address adds per tile;
//   * the row max is a v_max3_f32 chain.
// Left per tile and wave: 32 v_exp_f32, 16 v_cvt_pk_f16_f32, 16 v_dot2, 17 max, ~10 others.
template <int WAVES, int D, bool EXACT>
__global__ __launch_bounds__(WAVES * 64) void attn8_kernel(Attn8Args a) {
  static_assert(WAVES == 4 || WAVES == 8, "4 or 8 waves");
  constexpr int PPT = 16 / WAVES;            // LDS-DMA pieces (1 KiB) per wave per tile: the K pieces first, then the V^T pieces
  constexpr int KP = PPT / 2;                // ... of which K pieces
  static_assert(KP + PPT * (D - 2) <= 63, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) char smem[];   // [D][K tile | V^T tile]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  int bid = blockIdx.x;
  {   // XCD-contiguous walk (attention.hip): the query tiles of one (sample, head) share its K / V^T in one XCD's L2
    const int nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
  }
  const int qtile = bid % a.q_tiles;
  const int bh = bid / a.q_tiles;
  const int b = bh / a.heads, h = bh - b * a.heads;
  const int q0 = (qtile * WAVES + wave) * 32;
  const int nt = a.Sk / KT;

  const half_t* qbase = a.q + (size_t)b * a.Sq * a.ldq + (size_t)h * 64;
  const half_t* kbase = a.k + (size_t)b * a.Sk * a.ldk + (size_t)h * 64;
  const half_t* vbase = a.vt + ((size_t)b * a.heads + h) * 64 * (size_t)a.ldv;

  // ---- Q fragments (B operand of K.Q^T): lane (query l31, k half hi) holds 8 consecutive channels per 16-deep step, carrying
  //      d^-0.5 * log2(e): in the UNet the producing q|k|v GEMM multiplied it into its fp32 accumulator (prescaled: nothing
  //      to do here, q was rounded once); a caller's plain q is multiplied here (one more fp16 rounding per element) ----
  half8 qf[4];
  {
    const int q = q0 + l31;
    const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      qf[kk] = (q < a.Sq) ? *reinterpret_cast<const half8*>(qbase + (size_t)q * a.ldq + kk * 16 + hi * 8) : z;
      if (!a.prescaled) {
#pragma unroll
        for (int e = 0; e < 8; ++e) qf[kk][e] = (half_t)((float)qf[kk][e] * a.scale_log2);
      }
    }
  }

  // ---- LDS-DMA of the K / V^T tiles: loop-invariant per-lane byte offsets, scalar running offsets ----
  const unsigned k_bytes = (unsigned)(((size_t)(a.Sk - 1) * a.ldk + 64) * 2);
  const unsigned v_bytes = (unsigned)(((size_t)63 * a.ldv + a.Sk) * 2);
  unsigned voff[PPT];
#pragma unroll
  for (int i = 0; i < PPT; ++i) {
    const int p = (wave + i * WAVES) & 7;                    // piece = rows 8p .. 8p+7 of the K (i < KP) or V^T tile
    const int r = 8 * p + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);               // logical 16-B chunk this lane fetches (bank swizzle)
    voff[i] = (unsigned)((r * (i < KP ? a.ldk : a.ldv) + c * 8) * 2);
  }
  const int k_step = KT * a.ldk * 2;                         // bytes between consecutive K tiles
  int it_t = 0, it_stage = 0;                                // issue cursor
  auto issue_tile = [&]() {
    const bool live = it_t < nt;                             // wave-uniform; past the end: zero-sized resources (the counted
    const __amdgpu_buffer_rsrc_t rs_k =                      // waits below stay uniform, nothing is fetched)
        __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(kbase), 0, (int)(live ? k_bytes : 0u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(vbase), 0, (int)(live ? v_bytes : 0u), 0x00020000);
    char* st = smem + it_stage * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int p = (wave + i * WAVES) & 7;
      if (i < KP) dma16(rs_k, st + p * 1024, voff[i], it_t * k_step);
      else dma16(rs_v, st + TILE_BYTES + p * 1024, voff[i], it_t * (KT * 2));
    }
    ++it_t;
    it_stage = (it_stage + 1 == D) ? 0 : it_stage + 1;
  };

  // ---- fragment addressing: row l31 (+ 32 per sub-tile / channel tile), logical 16-B chunk 2 s + hi of MFMA step s;
  //      the same offsets serve the K tile (step = 16 channels) and the permuted V^T tile (step = 16 keys) ----
  const int fsw = (l31 >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) foff[s] = l31 * 128 + (((s * 2 + hi) ^ fsw) * 16);
  int aoff[4], noff[4];                                      // ... plus the ring stage of tile j / tile j+1
#pragma unroll
  for (int s = 0; s < 4; ++s) aoff[s] = foff[s];

  floatx16 oacc[2], sA[2], sB[2], negm;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    oacc[0][r] = 0.f;
    oacc[1][r] = 0.f;
    negm[r] = 0.f;
  }
  float mrun = 0.f, ls0 = 0.f, ls1 = 0.f;

  auto read_k = [&](half8 (&kf)[2][4], const int (&off)[4]) {
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) kf[sub][kk] = *reinterpret_cast<const half8*>(smem + off[kk] + sub * 4096);
  };
  auto qk = [&](floatx16 (&s)[2], const half8 (&kf)[2][4]) {   // s = K.Q^T - m: the running max rides in as the C operand
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[sub][0], qf[0], negm, 0, 0, 0);
#pragma unroll
    for (int kk = 1; kk < 4; ++kk)   // kk outer: the two sub-tiles are independent MFMA chains
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) s[sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[sub][kk], qf[kk], s[sub], 0, 0, 0);
  };
  auto raw_max = [&](const floatx16 (&s)[2]) {
    float m0 = max3(s[0][0], s[0][1], s[0][2]), m1 = max3(s[1][0], s[1][1], s[1][2]);
#pragma unroll
    for (int r = 3; r < 15; r += 2) {
      m0 = max3(m0, s[0][r], s[0][r + 1]);
      m1 = max3(m1, s[1][r], s[1][r + 1]);
    }
    return max3(m0, m1, fmaxf(s[0][15], s[1][15]));
  };
  // P of one 32-key sub-tile: exp2 of the (already scaled and stabilised) scores, packed to fp16 in the k-slot order of the
  // P.V MFMA: register r = 8 s2 + e is key 16 s2 + (e & 3) + 8 (e >> 2) + 4 hi - the order the permuted V^T rows are read in
  auto softmax_sub = [&](const floatx16& s, half8 (&pf)[2]) {
    const half2v one2 = {(half_t)1.f, (half_t)1.f};
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
#pragma unroll
      for (int e = 0; e < 8; ++e) pf[s2][e] = (half_t)__builtin_amdgcn_exp2f(s[s2 * 8 + e]);
#pragma unroll
      for (int e = 0; e < 8; e += 4) {
        const half2v p01 = {pf[s2][e], pf[s2][e + 1]}, p23 = {pf[s2][e + 2], pf[s2][e + 3]};
        ls0 = __builtin_amdgcn_fdot2(p01, one2, ls0, false);
        ls1 = __builtin_amdgcn_fdot2(p23, one2, ls1, false);
      }
    }
  };
  auto read_v = [&](half8 (&vf)[2][2], const int (&off)[4], int sub) {   // [s2][ct]
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
        vf[s2][ct] = *reinterpret_cast<const half8*>(smem + off[sub * 2 + s2] + TILE_BYTES + ct * 4096);
  };
  auto pv = [&](const half8 (&vf)[2][2], const half8 (&pf)[2]) {
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) oacc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf[s2][ct], pf[s2], oacc[ct], 0, 0, 0);
  };
  // the running max moves up by `up` (per query; <= 0 only for the very first tile): everything accumulated so far is rescaled,
  // the pending scores `s` (formed against the old max) are shifted, and the C-operand block follows
  auto move_max = [&](floatx16 (&s)[2], float up) {
    const float alpha = __builtin_amdgcn_exp2f(-up);
    mrun += up;
    ls0 *= alpha;
    ls1 *= alpha;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      oacc[0][r] *= alpha;
      oacc[1][r] *= alpha;
      s[0][r] -= up;
      s[1][r] -= up;
      negm[r] = -mrun;
    }
  };

  // ---- prologue: D - 1 tiles in flight, scores of tile 0, exact first running max ----
#pragma unroll
  for (int p = 0; p < D - 1; ++p) {
    asm volatile("" ::: "memory");                           // keep the DMA issue order: the counted waits rely on it
    issue_tile();
  }
  wait_vmcnt_barrier<(PPT - KP) + PPT * (D - 2)>();          // K tile 0 has landed for every wave
  {
    half8 kf[2][4];
    read_k(kf, aoff);
    qk(sA, kf);                                              // negm == 0: plain scaled scores
    // seed the running max: oacc / ls are still zero, so nothing is rescaled - move_max would multiply them by
    // exp2(-up), which is +inf for a first tile far below zero (0 * inf = NaN unless the compiler happens to fold it)
    const float up = xor32_max(raw_max(sA));
    mrun = up;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      sA[0][r] -= up;
      sA[1][r] -= up;
      negm[r] = -up;
    }
  }

  int st_cur = 0;                                            // ring stage of tile j
  // One key tile.  cur: stabilised scores of tile j; nxt: receives those of tile j+1 (NEXT).
  auto step = [&](floatx16 (&cur)[2], floatx16 (&nxt)[2], auto next_c) {
    constexpr bool NEXT = decltype(next_c)::value;
    // K tile j+1 and V^T tile j have landed (loads of one wave return in order: at most V^T(j+1) and the D - 3 younger
    // tiles are still outstanding); every wave is through tile j-1, whose stage the DMA issued next overwrites
    wait_vmcnt_barrier<(PPT - KP) + PPT * (D - 3)>();
    issue_tile();
    const int st_nxt = (st_cur + 1 == D) ? 0 : st_cur + 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) noff[s] = foff[s] + st_nxt * STAGE_BYTES;
    half8 kf[2][4], vf0[2][2], vf1[2][2], pf0[2], pf1[2];
    // ---- phase A: scores of tile j+1 on the matrix pipe | exp of sub-tile 0 of tile j on the VALU ----
    // (fragment reads in their own scheduling region; the exp work leads each group so that it covers the reads' latency)
    if constexpr (NEXT) read_k(kf, noff);
    read_v(vf0, aoff, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NEXT) qk(nxt, kf);
    softmax_sub(cur[0], pf0);
    if constexpr (NEXT) {
#pragma unroll
      for (int g = 0; g < 8; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase B1: P.V of sub-tile 0 | exp of sub-tile 1 ----
    read_v(vf1, aoff, 1);
    __builtin_amdgcn_sched_barrier(0);
    pv(vf0, pf0);
    softmax_sub(cur[1], pf1);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x400, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase B2: P.V of sub-tile 1 | row max of tile j+1 ----
    pv(vf1, pf1);
    float mx = 0.f;
    if constexpr (NEXT) {
      mx = raw_max(nxt);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (NEXT) {
      // how far tile j+1 rises above the running max, per query (the other half-wave holds the other 32 keys)
      const float d = xor32_max(mx);
      constexpr float kThr = EXACT ? 0.f : 8.f;
      if (__any(d > kThr)) move_max(nxt, fmaxf(d, 0.f));     // wave-uniform, rare after the first tiles
    }
    st_cur = st_nxt;
#pragma unroll
    for (int s = 0; s < 4; ++s) aoff[s] = noff[s];
  };
  {
    using Tt = std::true_type;
    using Ff = std::false_type;
    int j = 0;
    for (; j + 2 < nt; j += 2) {
      step(sA, sB, Tt{});
      step(sB, sA, Tt{});
    }
    if (nt - j == 2) {
      step(sA, sB, Tt{});
      step(sB, sA, Ff{});
    } else {
      step(sA, sB, Ff{});
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // the zero-sized tail DMAs too, before the workgroup retires

  // ---- normalise + store: lane (query l31) holds channels ct*32 + 8 g + 4 hi + 0..3 ----
  const float inv = 1.0f / xor32_sum(ls0 + ls1);
  const int q = q0 + l31;
  half_t* orow = a.out + ((size_t)b * a.Sq + q) * a.ldo + (size_t)h * 64;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int g = 0; g < 4; g += 2) {
      // widen the stores: the two half-waves trade 8-byte groups so that each lane ends up with 16 contiguous bytes
      // (lower half: channels 8 g .. 8 g + 7, upper half: 8 g + 8 .. 8 g + 15 of this 32-channel tile)
      unsigned w[2][2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const half2v lo = {(half_t)(oacc[ct][4 * (g + u)] * inv), (half_t)(oacc[ct][4 * (g + u) + 1] * inv)};
        const half2v hh = {(half_t)(oacc[ct][4 * (g + u) + 2] * inv), (half_t)(oacc[ct][4 * (g + u) + 3] * inv)};
        w[u][0] = __builtin_bit_cast(unsigned, lo);
        w[u][1] = __builtin_bit_cast(unsigned, hh);
      }
      unsigned o4[4];
#pragma unroll
      for (int dw = 0; dw < 2; ++dw) {
        const auto r = __builtin_amdgcn_permlane32_swap(w[0][dw], w[1][dw], false, false);
        const unsigned r0 = r[0], r1 = r[1];
        o4[dw] = r0;         // lower half: own group g              | upper half: the lower half's group g + 1
        o4[2 + dw] = r1;     // lower half: the upper half's group g | upper half: own group g + 1
      }
      if (q < a.Sq) {
        typedef unsigned uint4v __attribute__((ext_vector_type(4)));
        const uint4v o = {o4[0], o4[1], o4[2], o4[3]};
        out_store(reinterpret_cast<uint4v*>(orow + ct * 32 + 8 * g + 8 * hi), o);
      }
    }
}

template <int WAVES, int D, bool EXACT>
void launch8(const Attn8Args& a0, int B, hipStream_t s) {
  Attn8Args a = a0;
  a.q_tiles = cdiv(a.Sq, WAVES * 32);
  constexpr size_t lds = (size_t)D * STAGE_BYTES;
  auto k = attn8_kernel<WAVES, D, EXACT>;
  static DynLdsOnce once;
  once.set(k, lds);
  hipLaunchKernelGGL(k, dim3(a.q_tiles * a.heads * B), dim3(WAVES * 64), lds, s, a);
}

}  // namespace

// Build-time decision of the producers of V^T (UNet builder, sd_op_attention): d = 64 and whole 64-key tiles run on this
// kernel, which reads V^T in the permuted key order of AttnDesc::vt_perm.  SD_ATTN8=0 keeps the general kernels (A/B).
bool attention8_shape_ok(int d, int Sq, int Sk) {
  static const bool off = tune_env_int("SD_ATTN8", 1) == 0;
  return !off && d == 64 && Sq >= 1 && Sk >= KT && Sk % KT == 0;
}

bool attention8_ok(const AttnDesc& d) {
  const size_t lim = (size_t)1 << 31;
  return d.vt_perm == 1 && d.d == 64 && d.Sk >= KT && d.Sk % KT == 0 && d.ldq % 8 == 0 && d.ldk % 8 == 0 && d.ldv % 8 == 0 &&
         d.ldo % 8 == 0 && (size_t)d.Sk * d.ldk * 2 < lim && (size_t)64 * d.ldv * 2 < lim;
}

void launch_attention8(const AttnDesc& d, hipStream_t s) {
  Attn8Args a{d.q, d.k, d.vt, d.out, d.heads, d.Sq, d.Sk, d.ldq, d.ldk, d.ldv, d.ldo, 1.4426950408889634f / 8.0f, 0, d.q_prescaled};
  static const bool exact = tune_env_int("SD_ATTN8_EXACT", 0) != 0;
  // 256-query workgroups when they give at least half the CUs one, else 128-query ones: more, smaller workgroups
  const long wg8 = (long)d.B * d.heads * cdiv(d.Sq, 256);
  static const int force = tune_env_int("SD_ATTN8_WAVES", 0);
  const bool eight = force ? force == 8 : wg8 >= 128;
  if (eight) {
    if (exact) launch8<8, 4, true>(a, d.B, s);
    else launch8<8, 4, false>(a, d.B, s);
  } else {
    if (exact) launch8<4, 4, true>(a, d.B, s);
    else launch8<4, 4, false>(a, d.B, s);
  }
  SD_HIP(hipGetLastError());
}

}  // namespace sd
