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

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace sd {
namespace {

#ifndef SD_ATTN8_SK_DEFAULT
#define SD_ATTN8_SK_DEFAULT 1                // the balanced form (SK below) where sk_plan() finds the classic grid unbalanced; 0 = never
#endif
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
  // balanced form (SK): workgroup g runs the units [g * upw, (g + 1) * upw) of the linear order ((sample, head), query tile, key
  // tile); a query tile whose key tiles were shared by several workgroups is finished by the last of them to arrive
  int upw, total_units, max_seg;
  int dbg;            // timing ablations (SD_ATTN8_SK_DBG): 1 = no partial stores, 2 = no arrivals / merge (results are wrong)
  int xcd_local;      // every query tile's segments run on ONE XCD (launch geometry + verified dispatch order): partials through its L2
  float* part;        // [workgroup * max_seg + segment][WAVES][8][64][4] O | [64][2] (m, l)
  unsigned* cnt;      // [(sample, head) * q_tiles][WAVES] arrivals; left at zero
};
constexpr int SK_WAVE_FLOATS = 32 * 64 + 2 * 64;   // one wave's partial result: O (32 x 64 fp32), running max and sum per lane

typedef unsigned uint4v_t __attribute__((ext_vector_type(4)));
typedef unsigned uint2v_t __attribute__((ext_vector_type(2)));
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
//
// SK (round 6): the balanced form for launches whose query tiles do not fill the CUs evenly (S = 4096 at UNet batch 2: 160 query tiles
// of eight waves on 256 CUs).  The (query tile, key tile) units are dealt out evenly - 40 key tiles of a query tile and, where the
// range crosses into the next query tile, a second segment - each segment leaves (m, l, O) of its keys in HBM, and the LAST workgroup
// to arrive at a query tile (one agent-scope counter per tile, no spinning) merges them in segment order: bit-reproducible.
template <int WAVES, int D, bool EXACT, bool SK>
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
  const int nt_all = a.Sk / KT;
  // ---- normalise + store the 32 queries this wave holds of query tile qg_: lane (query l31) has channels ct*32 + 8 g + 4 hi + 0..3 ----
  auto store_out = [&](int qg_, const floatx16 (&o)[2], float inv) __attribute__((always_inline)) {
    const int qtile_ = qg_ % a.q_tiles, bh_ = qg_ / a.q_tiles;
    const int b_ = bh_ / a.heads, h_ = bh_ - b_ * a.heads;
    const int q = (qtile_ * WAVES + wave) * 32 + l31;
    half_t* orow = a.out + ((size_t)b_ * a.Sq + q) * a.ldo + (size_t)h_ * 64;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int g = 0; g < 4; g += 2) {
        // widen the stores: the two half-waves trade 8-byte groups so that each lane ends up with 16 contiguous bytes
        // (lower half: channels 8 g .. 8 g + 7, upper half: 8 g + 8 .. 8 g + 15 of this 32-channel tile)
        unsigned w[2][2];
#pragma unroll
        for (int u2 = 0; u2 < 2; ++u2) {
          const half2v lo = {(half_t)(o[ct][4 * (g + u2)] * inv), (half_t)(o[ct][4 * (g + u2) + 1] * inv)};
          const half2v hh = {(half_t)(o[ct][4 * (g + u2) + 2] * inv), (half_t)(o[ct][4 * (g + u2) + 3] * inv)};
          w[u2][0] = __builtin_bit_cast(unsigned, lo);
          w[u2][1] = __builtin_bit_cast(unsigned, hh);
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
          const uint4v_t ov = {o4[0], o4[1], o4[2], o4[3]};
          out_store(reinterpret_cast<uint4v_t*>(orow + ct * 32 + 8 * g + 8 * hi), ov);
        }
      }
  };
  // The partial results of the balanced form travel with scoped stores and loads instead of a release / acquire fence pair - sc1 (agent
  // scope: written through / read past this XCD's L2), or sc0 alone where the launch keeps a query tile's segments on one XCD: a fence writes back and invalidates the WHOLE L2 of the XCD, once per segment and
  // workgroup, and takes the K / V tiles of every other workgroup with it (measured: 150 us against the classic grid's 70).
  typedef float float4v __attribute__((ext_vector_type(4)));
  typedef float float2v __attribute__((ext_vector_type(2)));
  constexpr int kAgent = 16;                                 // aux bit 4 = sc1
  constexpr int kLocal = 1;                                  // aux bit 0 = sc0: past the CU's vector cache, served by the XCD's L2
  auto part_rsrc = [&](int g, int sg, bool live) {
    float* base = a.part + ((size_t)(g * a.max_seg + sg) * WAVES + wave) * SK_WAVE_FLOATS;
    return __builtin_amdgcn_make_buffer_rsrc(base, 0, live ? SK_WAVE_FLOATS * 4 : 0, 0x00020000);
  };
  int pq0 = -1, pn0 = 0, pq1 = -1, pn1 = 0;                  // query tiles this workgroup left partial results for (at most two)
  unsigned o0 = 0, o1 = 0;                                   // ... and what their arrival counters held
  bool posted0 = false;
  int u = SK ? bid * a.upw : 0;                              // unit cursor of the balanced form
  const int u_end = SK ? min(u + a.upw, a.total_units) : 1;
  int seg = 0;

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
  // ---- fragment addressing: row l31 (+ 32 per sub-tile / channel tile), logical 16-B chunk 2 s + hi of MFMA step s;
  //      the same offsets serve the K tile (step = 16 channels) and the permuted V^T tile (step = 16 keys) ----
  const int fsw = (l31 >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) foff[s] = l31 * 128 + (((s * 2 + hi) ^ fsw) * 16);
  int aoff[4], noff[4];                                      // ... plus the ring stage of tile j / tile j+1
#pragma unroll
  for (int s = 0; s < 4; ++s) aoff[s] = foff[s];
  // The ring outlives a segment of the balanced form: its issue cursor walks the workgroup's units across query-tile boundaries, so
  // the first tiles of the next segment are in flight while the last ones of this segment are consumed.
  int it_stage = 0, st_cur = 0;
  int i_u = u, i_kt = 0, i_qg = 0;
  const half_t* i_kbase = a.k;
  const half_t* i_vbase = a.vt;
  auto issue_bases = [&]() {
    const int bh_ = i_qg / a.q_tiles;
    const int b_ = bh_ / a.heads, h_ = bh_ - b_ * a.heads;
    i_kbase = a.k + (size_t)b_ * a.Sk * a.ldk + (size_t)h_ * 64;
    i_vbase = a.vt + ((size_t)b_ * a.heads + h_) * 64 * (size_t)a.ldv;
  };
  if constexpr (SK) {
    i_qg = u / nt_all;
    i_kt = u - i_qg * nt_all;
    if (u < u_end) issue_bases();
  }
  while (u < u_end) {   // (the classic form: one pass)
  int qg, kt0, kt1;
  if constexpr (SK) {
    qg = u / nt_all;
    kt0 = u - qg * nt_all;
    kt1 = min(nt_all, kt0 + (u_end - u));
  } else {
    qg = bid;
    kt0 = 0;
    kt1 = nt_all;
  }
  const int qtile = qg % a.q_tiles;
  const int bh = qg / a.q_tiles;
  const int b = bh / a.heads, h = bh - b * a.heads;
  const int q0 = (qtile * WAVES + wave) * 32;
  const int nt = kt1 - kt0;

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

  int it_t = kt0;                                            // issue cursor of the classic form
  auto issue_tile = [&]() {
    bool live;                                               // wave-uniform; past the end: zero-sized resources (the counted
    const half_t* kb;                                        // waits below stay uniform, nothing is fetched)
    const half_t* vb;
    int t;
    if constexpr (SK) {
      live = i_u < u_end;
      kb = i_kbase;
      vb = i_vbase;
      t = i_kt;
    } else {
      live = it_t < kt1;
      kb = kbase;
      vb = vbase;
      t = it_t;
    }
    const __amdgpu_buffer_rsrc_t rs_k = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(kb), 0, (int)(live ? k_bytes : 0u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(vb), 0, (int)(live ? v_bytes : 0u), 0x00020000);
    char* st = smem + it_stage * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < PPT; ++i) {
      const int p = (wave + i * WAVES) & 7;
      if (i < KP) dma16(rs_k, st + p * 1024, voff[i], t * k_step);
      else dma16(rs_v, st + TILE_BYTES + p * 1024, voff[i], t * (KT * 2));
    }
    if constexpr (SK) {
      ++i_u;
      if (++i_kt == nt_all) {                                // on into the next query tile (another (sample, head) every q_tiles of them)
        i_kt = 0;
        ++i_qg;
        if (i_u < u_end) issue_bases();
      }
    } else {
      ++it_t;
    }
    it_stage = (it_stage + 1 == D) ? 0 : it_stage + 1;
  };

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
  if (!SK || seg == 0) {   // (later segments of the balanced form: their first tiles were issued by the previous segment's steps, and K
                           //  tile 0 landed for every wave behind the barrier of its last step)
#pragma unroll
    for (int p = 0; p < D - 1; ++p) {
      asm volatile("" ::: "memory");                         // keep the DMA issue order: the counted waits rely on it
      issue_tile();
    }
    wait_vmcnt_barrier<(PPT - KP) + PPT * (D - 2)>();        // K tile 0 has landed for every wave
  }
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

  // (st_cur: ring stage of tile j)
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
      if constexpr (SK) {
        // the arrival of this workgroup's FIRST partial segment is posted here, four steps into a later segment: its stores are
        // acknowledged (a wave's vector memory operations complete in order and the counted waits of steps 2 and 3 left only younger
        // ones outstanding), nothing waits, and the other workgroups of that query tile find it long before they finish
        if (j == 2 && pq0 >= 0 && !posted0) {
          if (lane == 0) o0 = atomicAdd(a.cnt + pq0 * WAVES + wave, 1u);
          posted0 = true;
        }
      }
    }
    if (nt - j == 2) {
      step(sA, sB, Tt{});
      step(sB, sA, Ff{});
    } else {
      step(sA, sB, Ff{});
    }
  }
  if constexpr (!SK) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the zero-sized tail DMAs too, before the workgroup retires

  if constexpr (SK) {
    const int g_first = (qg * nt_all) / a.upw, g_last = (qg * nt_all + nt_all - 1) / a.upw;
    const int nseg = g_last - g_first + 1;                   // workgroups that share this query tile's keys
    if (nseg > 1) {
      // this segment's (m, l, O) leave for HBM; nobody waits for them here - the arrival is posted after the workgroup's last segment
      const __amdgpu_buffer_rsrc_t rs = part_rsrc(bid, seg, true);
      const float2v ml = {mrun, xor32_sum(ls0 + ls1)};
      auto put = [&](auto aux_c) __attribute__((always_inline)) {
        constexpr int AUX = decltype(aux_c)::value;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float4v v = {oacc[ct][4 * g], oacc[ct][4 * g + 1], oacc[ct][4 * g + 2], oacc[ct][4 * g + 3]};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4v_t, v), rs, ((ct * 4 + g) * 64 + lane) * 16, 0, AUX);
          }
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(uint2v_t, ml), rs, 32 * 64 * 4 + lane * 8, 0, AUX);
      };
      if (a.dbg & 1) {
      } else if (a.xcd_local) put(std::integral_constant<int, kLocal>{});
      else put(std::integral_constant<int, kAgent>{});
      if (pq0 < 0) {
        pq0 = qg;
        pn0 = nseg;
      } else {
        pq1 = qg;
        pn1 = nseg;
      }
    } else {
      store_out(qg, oacc, 1.0f / xor32_sum(ls0 + ls1));
    }
    u += nt;
    ++seg;
  } else {
    store_out(qg, oacc, 1.0f / xor32_sum(ls0 + ls1));
    break;
  }
  }   // segments

  if constexpr (SK) {
    // ---- arrivals, one counter per (query tile, wave): a wave's 32 queries are merged by the wave of the LAST segment to arrive ----
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the partial stores are acknowledged (and the zero-sized tail DMAs done)
    if (a.dbg & 2) return;
    if (lane == 0) {
      if (pq0 >= 0 && !posted0) o0 = atomicAdd(a.cnt + pq0 * WAVES + wave, 1u);
      if (pq1 >= 0) o1 = atomicAdd(a.cnt + pq1 * WAVES + wave, 1u);
    }
    o0 = __builtin_amdgcn_readfirstlane(o0);
    o1 = __builtin_amdgcn_readfirstlane(o1);
    auto merge = [&](int qg_, int nseg) __attribute__((always_inline)) {
      // segment order whoever merges (streaming-softmax update: the stabiliser follows the segments' own): bit-reproducible
      const int g_first = (qg_ * nt_all) / a.upw;
      floatx16 o[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o[0][r] = 0.f;
        o[1][r] = 0.f;
      }
      float mx = -3.0e38f, lsum = 0.f;
      auto fold = [&](const float4v (&v)[8], const float2v& ml) {
        const float mn = fmaxf(mx, ml[0]);
        const float wo = __builtin_amdgcn_exp2f(mx - mn), wi = __builtin_amdgcn_exp2f(ml[0] - mn);
        mx = mn;
        lsum = fmaf(ml[1], wi, lsum * wo);
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int e = 0; e < 4; ++e) o[t >> 2][4 * (t & 3) + e] = fmaf(v[t][e], wi, o[t >> 2][4 * (t & 3) + e] * wo);
      };
      auto fetch_aux = [&](int i, float4v (&v)[8], float2v& ml, bool live, auto aux_c) __attribute__((always_inline)) {
        constexpr int AUX = decltype(aux_c)::value;
        const int g = g_first + (live ? i : 0);
        const __amdgpu_buffer_rsrc_t rs = part_rsrc(g, qg_ - (g * a.upw) / nt_all, live);
#pragma unroll
        for (int t = 0; t < 8; ++t) v[t] = __builtin_bit_cast(float4v, __builtin_amdgcn_raw_buffer_load_b128(rs, (t * 64 + lane) * 16, 0, AUX));
        ml = __builtin_bit_cast(float2v, __builtin_amdgcn_raw_buffer_load_b64(rs, 32 * 64 * 4 + lane * 8, 0, AUX));
      };
      auto fetch = [&](int i, float4v (&v)[8], float2v& ml, bool live) __attribute__((always_inline)) {
        if (a.xcd_local) fetch_aux(i, v, ml, live, std::integral_constant<int, kLocal>{});
        else fetch_aux(i, v, ml, live, std::integral_constant<int, kAgent>{});
      };
      if (nseg <= 3) {   // all of them in flight at once
        float4v v[3][8];
        float2v ml[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) fetch(i, v[i], ml[i], i < nseg);
#pragma unroll
        for (int i = 0; i < 3; ++i)
          if (i < nseg) fold(v[i], ml[i]);
      } else {
        for (int i = 0; i < nseg; ++i) {
          float4v v[8];
          float2v ml;
          fetch(i, v, ml, true);
          fold(v, ml);
        }
      }
      if (lane == 0) a.cnt[qg_ * WAVES + wave] = 0;          // ready for the next launch
      store_out(qg_, o, 1.0f / lsum);
    };
    if (pq0 >= 0 && o0 == (unsigned)(pn0 - 1)) merge(pq0, pn0);
    if (pq1 >= 0 && o1 == (unsigned)(pn1 - 1)) merge(pq1, pn1);
  }
}

template <int WAVES, int D, bool EXACT>
void launch8(const Attn8Args& a0, int B, hipStream_t s) {
  Attn8Args a = a0;
  a.q_tiles = cdiv(a.Sq, WAVES * 32);
  constexpr size_t lds = (size_t)D * STAGE_BYTES;
  auto k = attn8_kernel<WAVES, D, EXACT, false>;
  static DynLdsOnce once;
  once.set(k, lds);
  hipLaunchKernelGGL(k, dim3(a.q_tiles * a.heads * B), dim3(WAVES * 64), lds, s, a);
}

template <int WAVES, int D, bool EXACT>
void launch8_sk(const Attn8Args& a, int grid, hipStream_t s) {
  constexpr size_t lds = (size_t)D * STAGE_BYTES;
  auto k = attn8_kernel<WAVES, D, EXACT, true>;
  static DynLdsOnce once;
  once.set(k, lds);
  hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), lds, s, a);
}

int device_cus() {
  static const int n = [] {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
    return v;
  }();
  return n;
}

// Do consecutive workgroup ids go round-robin over eight XCDs (id % 8), as the locality walks of this library assume?  Probed once with a
// grid that over-subscribes the chip; the balanced form may send its partial results through the XCD's L2 (sc0) instead of past it (sc1)
// only when this held for every workgroup - an experiment (see below), not the product path.
__global__ void xcc_probe_kernel(unsigned* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20) & 0xf;   // HW_REG_XCC_ID[3:0]
}
bool xcd_round_robin() {
  static const bool ok = [] {
    // OFF unless SD_ATTN8_SK_LOCAL=1 (with SD_TUNE): measured equal to the agent-scope path (62.7 vs 62.4 us), and the agent-scope path
    // does not depend on where the hardware places a workgroup (the probe below runs alone, not beside other streams' kernels)
    if (tune_env_int("SD_ATTN8_SK_LOCAL", 0) == 0) return false;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(nullptr, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return false;
    const int n = 8192;
    unsigned* d = nullptr;
    if (hipMalloc(&d, n * sizeof(unsigned)) != hipSuccess) return false;
    std::vector<unsigned> h(n, 0xffu);
    bool good = hipMemset(d, 0xff, n * sizeof(unsigned)) == hipSuccess;
    if (good) {
      hipLaunchKernelGGL(xcc_probe_kernel, dim3(n), dim3(256), 0, nullptr, d);
      good = hipDeviceSynchronize() == hipSuccess && hipMemcpy(h.data(), d, n * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess;
    }
    (void)hipFree(d);
    for (int i = 0; i < n && good; ++i) good = h[i] == (unsigned)(i & 7);
    return good;
  }();
  return ok;
}

// The balanced form's launch geometry for a problem (nothing when the classic grid is the better one).
struct SkPlan {
  bool on = false;
  bool xcd_local = false;   // each XCD's share of the workgroups (the kernel's XCD-contiguous walk) is whole query tiles
  int waves = 8, q_tiles = 0, upw = 0, total = 0, max_seg = 0, grid = 0;
  size_t part_bytes = 0;
  int n_cnt = 0;
};
SkPlan sk_plan(const AttnDesc& d) {
  // SD_ATTN8_SK (with SD_TUNE): 0 = never, 1 = where the classic grid leaves CUs idle (default), 2 = wherever it can run
  static const int env_mode = tune_env_int("SD_ATTN8_SK", SD_ATTN8_SK_DEFAULT);
  static const int force_waves = tune_env_int("SD_ATTN8_WAVES", 0);
  static const int min_upw = tune_env_int("SD_ATTN8_SK_MIN_UPW", 32);
  SkPlan p;
  const int mode = d.sk_force ? 2 : env_mode;
  if (mode == 0 || d.Sq % 32 != 0) return p;
  const int ncu = device_cus();
  const long wg8 = (long)d.B * d.heads * cdiv(d.Sq, 256);
  p.waves = force_waves ? (force_waves == 8 ? 8 : 4) : (wg8 >= 128 ? 8 : 4);
  if (d.Sq % (p.waves * 32) != 0) return p;                  // whole query tiles only: every wave of a segment has 32 live queries
  p.q_tiles = d.Sq / (p.waves * 32);
  const long classic = (long)d.B * d.heads * p.q_tiles;
  const int nt = d.Sk / KT;
  const long total = classic * nt;
  if (total >= ((long)1 << 30)) return p;
  // the library's rule: where the classic grid's last round leaves more than a fifth of the CUs idle (160 or 320 eight-wave workgroups on
  // 256 CUs - UNet batch 2 and 4 at 64x64 - run 69 -> 62 and 138 -> 109 us; 480 workgroups, 0.94 full, 158 -> 170: the merge costs ~ 12 us)
  if (mode == 1 && (double)classic / (double)(((classic + ncu - 1) / ncu) * ncu) >= 0.8) return p;
  p.total = (int)total;
  static const int wgs_per_cu = std::max(1, tune_env_int("SD_ATTN8_SK_WGS", 1));
  const long want = (long)ncu * wgs_per_cu;
  p.upw = (int)((total + want - 1) / want);
  if (d.sk_upw > 0) p.upw = d.sk_upw;                        // (operator tests: any split)
  else if (!d.sk_force && p.upw < min_upw) return p;
  p.grid = cdiv(p.total, p.upw);
  p.max_seg = (p.upw + nt - 2) / nt + 1;
  p.part_bytes = (size_t)p.grid * p.max_seg * p.waves * SK_WAVE_FLOATS * sizeof(float);
  // the kernel's walk hands XCD x the logical workgroups [x * grid / 8, (x + 1) * grid / 8): whole query tiles when that many units are
  p.xcd_local = p.grid % 8 == 0 && p.total == p.grid * p.upw && ((long)(p.grid / 8) * p.upw) % nt == 0 && xcd_round_robin();
  p.n_cnt = (int)classic * p.waves;                          // one arrival counter per (query tile, wave)
  p.on = true;
  return p;
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

bool attention8_sk_scratch(const AttnDesc& d, size_t* part_bytes, int* n_counters) {
  const SkPlan p = sk_plan(d);
  if (part_bytes) *part_bytes = p.on ? p.part_bytes : 0;
  if (n_counters) *n_counters = p.on ? p.n_cnt : 0;
  return p.on;
}

void launch_attention8(const AttnDesc& d, hipStream_t s) {
  Attn8Args a{d.q, d.k, d.vt, d.out, d.heads, d.Sq, d.Sk, d.ldq, d.ldk, d.ldv, d.ldo, 1.4426950408889634f / 8.0f, 0, d.q_prescaled};
  static const bool exact = tune_env_int("SD_ATTN8_EXACT", 0) != 0;
  if (d.sk_part && d.sk_cnt) {   // the caller offers scratch: the balanced form where it pays
    const SkPlan p = sk_plan(d);
    if (p.on && p.part_bytes <= d.sk_part_bytes && p.n_cnt <= d.sk_cnt_n) {
      a.q_tiles = p.q_tiles;
      a.upw = p.upw;
      a.total_units = p.total;
      a.max_seg = p.max_seg;
      a.xcd_local = p.xcd_local ? 1 : 0;
      static const int dbg = tune_env_int("SD_ATTN8_SK_DBG", 0);
      a.dbg = dbg;
      static const bool verbose = tune_env_int("SD_ATTN8_SK_VERBOSE", 0) != 0;
      if (verbose) fprintf(stderr, "attention8 balanced: grid %d x %d waves, %d units each, max_seg %d, xcd_local %d, dbg %d\n", p.grid, p.waves, p.upw, p.max_seg, a.xcd_local, dbg);
      a.part = d.sk_part;
      a.cnt = d.sk_cnt;
      if (p.waves == 8) {
        if (exact) launch8_sk<8, 4, true>(a, p.grid, s);
        else launch8_sk<8, 4, false>(a, p.grid, s);
      } else {
        if (exact) launch8_sk<4, 4, true>(a, p.grid, s);
        else launch8_sk<4, 4, false>(a, p.grid, s);
      }
      SD_HIP(hipGetLastError());
      return;
    }
  }
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
