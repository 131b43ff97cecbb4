// K1-K3 - streaming-softmax multi-head attention on gfx950 MFMA, three schedules mirroring
// python_coreml_stable_diffusion/attention.py (selected at run time like unet.py:51-59):
//
//   ORIGINAL         attention.py:147-168  heads batched in the grid, score tile oriented [q][k]
//                    (the reference's "bhqk"), softmax over k = ACROSS LANES: row max by DPP
//                    wavefront shuffles, P re-laid-out through a wave-private LDS tile for P.V.
//   SPLIT_EINSUM     attention.py:24-72    per-head key-major score tile [k][q] (the reference's
//                    "bkhq", softmax over dim=1): computed as K.Q^T so every lane owns one query
//                    column and reduces over keys IN REGISTERS - no cross-lane traffic except one
//                    half-wave exchange; P feeds the P.V MFMA straight from registers.
//   SPLIT_EINSUM_V2  attention.py:77-144   SPLIT_EINSUM with the query axis cut into 512-query
//                    chunks (CHUNK_SIZE attention.py:75): one 8-wave workgroup x 64 queries per
//                    wave per chunk, or 2 / 4 smaller workgroups per chunk when the launch has too
//                    few chunks to fill the chip.  Falls back to SPLIT_EINSUM when S_q < 512
//                    (attention.py:88-92) and rejects S_q % 512 != 0 instead of silently dropping
//                    the tail (attention.py:86).
//
// All three: scores never touch HBM (the reference materialises attn_weights, 335 MB at
// S=4096), scale d^-0.5 applied to the scores (attention.py:49,123,159) folded with log2(e) so
// the exponent is the native v_exp_f32 (base 2; cf. the reference's own exp2 softmax,
// attention.py:11-22), fp32 running max / sum / accumulators, fp16 operands.
// Data: q [B][Sq][ldq], k [B][Sk][ldk] token-major (head h = columns h*d..), vt [B][C][ldv]
// channel-major (written by the V projection GEMM's transposed epilogue) so that both MFMA
// operands of P.V are key-contiguous.  K / V^T tiles of 64 keys are staged HBM -> VGPR -> LDS
// (padded, bank-conflict-free rows) with a register double buffer, one barrier per tile.
#include "kernels.h"
#include <type_traits>

namespace sd {
namespace {

constexpr int KT = 64;   // keys per tile

struct AttnArgs {
  const half_t* q;
  const half_t* k;
  const half_t* vt;
  half_t* out;
  int heads, d, Sq, Sk;
  int ldq, ldk, ldv, ldo;
  float scale_log2;   // d^-0.5 * log2(e)
  int q_tiles;        // query tiles per (sample, head): set by launch_one
};

template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
// gfx950 cross-row exchanges: v_permlane16_swap trades the odd 16-lane rows of one operand with
// the even rows of the other, v_permlane32_swap the two wave halves; fed the same value twice,
// the two results are {own, partner} for the xor-16 / xor-32 butterfly step - one VALU op
// instead of a ds_bpermute round trip through the LDS crossbar.
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
__device__ __forceinline__ float xor32_max(float v) {
  float x, y;
  swap32(v, x, y);
  return fmaxf(x, y);
}
__device__ __forceinline__ float xor32_sum(float v) {
  float x, y;
  swap32(v, x, y);
  return x + y;
}
// reductions across the 32 lanes that share (lane >> 5): 4 DPP steps inside each 16-lane row
// (quad_perm xor1, xor2, row_half_mirror, row_mirror) + one permlane16 swap across the two rows
__device__ __forceinline__ void half_wave_max16(float (&v)[16]) {
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], dpp_mov<0xB1>(v[r]));
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], dpp_mov<0x4E>(v[r]));
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], dpp_mov<0x141>(v[r]));
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], dpp_mov<0x140>(v[r]));
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float x, y;
    swap16(v[r], x, y);
    v[r] = fmaxf(x, y);
  }
}
__device__ __forceinline__ float half_wave_sum(float v) {
  v += dpp_mov<0xB1>(v);
  v += dpp_mov<0x4E>(v);
  v += dpp_mov<0x141>(v);
  v += dpp_mov<0x140>(v);
  float x, y;
  swap16(v, x, y);
  return x + y;
}

// MODE 0: ORIGINAL ([q][k] tiles), MODE 1: SPLIT_EINSUM ([k][q] tiles).  QT = 32-query tiles per wave.
template <int DK16, int DC32, int MODE, int WAVES, int QT>
__global__ __launch_bounds__(WAVES * 64) void attn_kernel(AttnArgs a) {
  constexpr int NT = WAVES * 64;
  constexpr int DKP = DK16 * 16;                  // head dim padded for the QK^T k-loop
  constexpr int DCP = DC32 * 32;                  // head dim padded for the PV output tiles
  constexpr int KROW = DKP + 8;                   // halves; (DKP*2+16) B = odd multiple of 16 B
  constexpr int VROW = (MODE == 0) ? KT + 8 : KT + 4;   // 144 B (b128 reads) / 136 B (b64 reads)
  constexpr int PROW = KT + 8;
  constexpr int KCH = DKP / 8;                    // 16-B chunks per K row
  constexpr int K_ITEMS = (KT * KCH + NT - 1) / NT;
  constexpr int V_ITEMS = (DCP * (KT / 8) + NT - 1) / NT;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  half_t* Ks = reinterpret_cast<half_t*>(smem);                    // [2][KT][KROW]
  half_t* Vs = Ks + 2 * KT * KROW;                                 // [2][DCP][VROW]
  half_t* Ps = Vs + 2 * DCP * VROW;                                // MODE 0: [WAVES][QT*32][PROW]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  // 1-D grid of (sample, head, query tile) with the query tile fastest, XCD-aware: the dispatcher deals consecutive
  // workgroup ids round-robin over the 8 XCDs, so id -> (id % 8) * (n / 8) + id / 8 hands every XCD a CONTIGUOUS run of
  // that order - the query tiles of one (sample, head) share its K / V in ONE XCD's L2 instead of pulling all keys and
  // values of every head through the fabric into all eight
  int bid = blockIdx.x;
  {
    const int nwg = gridDim.x;
    const int q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    bid = ((xcd < r8) ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;   // bijective for any nwg
  }
  const int qtile = bid % a.q_tiles;
  const int bh = bid / a.q_tiles;
  const int b = bh / a.heads, h = bh - b * a.heads;
  const int q_wave0 = (qtile * WAVES + wave) * QT * 32;

  const half_t* qbase = a.q + (size_t)b * a.Sq * a.ldq + (size_t)h * a.d;
  const half_t* kbase = a.k + (size_t)b * a.Sk * a.ldk + (size_t)h * a.d;
  const half_t* vbase = a.vt + ((size_t)b * a.heads * a.d + (size_t)h * a.d) * a.ldv;
  const half8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};

  // ---- Q fragments stay in registers for the whole kernel ----
  half8 qf[QT][DK16];
#pragma unroll
  for (int t = 0; t < QT; ++t) {
    const int q = q_wave0 + t * 32 + l31;
#pragma unroll
    for (int kk = 0; kk < DK16; ++kk) {
      const int c = kk * 16 + hi * 8;
      qf[t][kk] = (q < a.Sq && c < a.d) ? *reinterpret_cast<const half8*>(qbase + (size_t)q * a.ldq + c) : zero8;
    }
  }

  half8 kreg[K_ITEMS], vreg[V_ITEMS];
  auto load_tiles = [&](int kt) {
    const int key0 = kt * KT;
#pragma unroll
    for (int i = 0; i < K_ITEMS; ++i) {
      const int idx = tid + i * NT;
      const int row = idx / KCH, ch = idx - row * KCH;
      const int key = key0 + row, c = ch * 8;
      kreg[i] = (idx < KT * KCH && key < a.Sk && c < a.d)
                    ? *reinterpret_cast<const half8*>(kbase + (size_t)key * a.ldk + c)
                    : zero8;
    }
#pragma unroll
    for (int i = 0; i < V_ITEMS; ++i) {
      const int idx = tid + i * NT;
      const int row = idx >> 3, ch = idx & 7;
      const int key = key0 + ch * 8;
      // contract: columns [Sk, ldv) of vt are zero (the UNet zero-fills them once at create time)
      vreg[i] = (idx < DCP * 8 && row < a.d && key < a.ldv)
                    ? *reinterpret_cast<const half8*>(vbase + (size_t)row * a.ldv + key)
                    : zero8;
    }
  };
  auto store_tiles = [&](int buf) {
    half_t* ks = Ks + buf * KT * KROW;
    half_t* vs = Vs + buf * DCP * VROW;
#pragma unroll
    for (int i = 0; i < K_ITEMS; ++i) {
      const int idx = tid + i * NT;
      const int row = idx / KCH, ch = idx - row * KCH;
      if (idx < KT * KCH) *reinterpret_cast<half8*>(ks + row * KROW + ch * 8) = kreg[i];
    }
#pragma unroll
    for (int i = 0; i < V_ITEMS; ++i) {
      const int idx = tid + i * NT;
      const int row = idx >> 3, ch = idx & 7;
      if (idx < DCP * 8) {
        if constexpr (MODE == 0) {
          *reinterpret_cast<half8*>(vs + row * VROW + ch * 8) = vreg[i];
        } else {   // 136-B rows are only 8-B aligned
          half4 lo = {vreg[i][0], vreg[i][1], vreg[i][2], vreg[i][3]};
          half4 hi4 = {vreg[i][4], vreg[i][5], vreg[i][6], vreg[i][7]};
          *reinterpret_cast<half4*>(vs + row * VROW + ch * 8) = lo;
          *reinterpret_cast<half4*>(vs + row * VROW + ch * 8 + 4) = hi4;
        }
      }
    }
  };

  floatx16 oacc[QT][DC32];
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int ct = 0; ct < DC32; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[t][ct][r] = 0.f;

  // running softmax state.  MODE 1: one query per lane.  MODE 0: 16 query rows per lane.
  constexpr int NS = (MODE == 0) ? 16 : 1;
  float mrun[QT][NS], lrun[QT][NS];
#pragma unroll
  for (int t = 0; t < QT; ++t)
#pragma unroll
    for (int r = 0; r < NS; ++r) {
      mrun[t][r] = -1e30f;
      lrun[t][r] = 0.f;
    }

  const int ntiles = (a.Sk + KT - 1) / KT;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  // One key tile.  LAST is a compile-time flag: only the final tile can be ragged (needs the
  // -inf masking) and has nothing to prefetch, so the steady-state loop carries neither the
  // per-score compare/select nor the conditional loads (the compiler if-converts a runtime flag).
  auto tile_step = [&](const int kt, auto first_c, auto last_c) {
    constexpr bool FIRST = decltype(first_c)::value;   // ORIGINAL: the first tile seeds the running max exactly
    constexpr bool LAST = decltype(last_c)::value;
    const int buf = kt & 1;
    constexpr bool more = !LAST;
    if constexpr (more) load_tiles(kt + 1);
    const half_t* ks = Ks + buf * KT * KROW;
    const half_t* vs = Vs + buf * DCP * VROW;
    const bool tail = LAST && (kt + 1) * KT > a.Sk;

    // ---------------- scores: two 32-key sub-tiles ----------------
    floatx16 sacc[QT][2];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) sacc[t][sub][r] = 0.f;
    // With one wave per SIMD nothing hides an LDS round trip, and the compiler's just-in-time
    // ds_read -> s_waitcnt -> MFMA chains expose one per fragment: request every K fragment of the
    // tile up front (d <= 64: 32 VGPRs), and the V^T fragments right behind the S MFMAs so that
    // their latency disappears under the softmax VALU work.
    constexpr bool KPRE = DK16 <= 4;
    // Key row that lane l31 of sub-tile `sub` multiplies.  ORIGINAL interleaves the two sub-tiles
    // (keys 2*l31 and 2*l31+1) so that a lane's two probabilities of a query row are ADJACENT keys:
    // one packed 32-bit P write per row instead of two 16-bit ones, P and V^T stay in key order.
    auto krow = [&](int sub) { return MODE == 0 ? 2 * l31 + sub : sub * 32 + l31; };
    constexpr bool VPRE = DC32 <= 2 && DK16 <= 4;
    half8 kfr[KPRE ? DK16 : 1][2];
    if constexpr (KPRE) {
#pragma unroll
      for (int kk = 0; kk < DK16; ++kk)
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
          kfr[kk][sub] = *reinterpret_cast<const half8*>(ks + krow(sub) * KROW + kk * 16 + hi * 8);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int kk = 0; kk < DK16; ++kk)   // kk outer: the two sub-tiles are independent MFMA chains
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
        half8 kf;
        if constexpr (KPRE) kf = kfr[kk][sub];
        else kf = *reinterpret_cast<const half8*>(ks + krow(sub) * KROW + kk * 16 + hi * 8);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
          if constexpr (MODE == 0)   // rows = q, cols = key
            sacc[t][sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(qf[t][kk], kf, sacc[t][sub], 0, 0, 0);
          else                       // rows = key, cols = q
            sacc[t][sub] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[t][kk], sacc[t][sub], 0, 0, 0);
        }
      }

    half8 vfr[VPRE ? 4 : 1][VPRE ? DC32 : 1];     // [16-key step][32-channel tile]
    if constexpr (VPRE) {
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int ct = 0; ct < DC32; ++ct) {
          if constexpr (MODE == 1) {   // keys of step s4 in the P-register order (two 4-key groups, 8 apart)
            const half_t* vp = vs + (ct * 32 + l31) * VROW + s4 * 16 + 4 * hi;
            const half4 v0 = *reinterpret_cast<const half4*>(vp);
            const half4 v1 = *reinterpret_cast<const half4*>(vp + 8);
            vfr[s4][ct] = half8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          } else {
            vfr[s4][ct] = *reinterpret_cast<const half8*>(vs + (ct * 32 + l31) * VROW + s4 * 16 + hi * 8);
          }
        }
      __builtin_amdgcn_sched_barrier(0);
    }

    if constexpr (MODE == 1) {
      // ======== SPLIT_EINSUM: lane = query column, registers = keys ========
      // VALU diet (the loop is VALU-, not MFMA-bound): the max runs on RAW scores (scale > 0 commutes
      // with max), scale and -max fold into ONE fma before the exp2, and the accumulator rescale is
      // skipped - exactly, alpha == 1 - on the (common) tiles where no lane's running max moved.
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        float mx = -3.0e38f;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (tail) {
              const int key = kt * KT + sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
              if (key >= a.Sk) sacc[t][sub][r] = -3.0e38f;
            }
            mx = fmaxf(mx, sacc[t][sub][r]);
          }
        mx = xor32_max(mx);                            // the other half-wave holds the other 32 keys
        const float mold = mrun[t][0];
        const float mnew = fmaxf(mold, mx * a.scale_log2);
        float ps4[4] = {0.f, 0.f, 0.f, 0.f};           // four short add chains instead of one of 32
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float p = __builtin_amdgcn_exp2f(fmaf(sacc[t][sub][r], a.scale_log2, -mnew));
            sacc[t][sub][r] = p;
            ps4[r & 3] += p;
          }
        const float psum = (ps4[0] + ps4[1]) + (ps4[2] + ps4[3]);
        if (__any(mnew > mold)) {                      // wave-uniform
          const float alpha = __builtin_amdgcn_exp2f(mold - mnew);
          lrun[t][0] *= alpha;
#pragma unroll
          for (int ct = 0; ct < DC32; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) oacc[t][ct][r] *= alpha;
          mrun[t][0] = mnew;
        }
        lrun[t][0] += psum;                            // per-lane partial (own keys); merged at the end
      }
      // P (registers) is already the B operand of O^T += V^T . P^T: k-slot (hi, e) of MFMA step s2
      // holds key  sub*32 + s2*16 + (e&3) + 8*(e>>2) + 4*hi ; V^T is gathered with the same map.
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
          half8 pf[QT];
#pragma unroll
          for (int t = 0; t < QT; ++t)
#pragma unroll
            for (int e = 0; e < 8; ++e) pf[t][e] = (half_t)sacc[t][sub][s2 * 8 + e];
#pragma unroll
          for (int ct = 0; ct < DC32; ++ct) {
            half8 vf;
            if constexpr (VPRE) {
              vf = vfr[sub * 2 + s2][ct];
            } else {
              const half_t* vp = vs + (ct * 32 + l31) * VROW + sub * 32 + s2 * 16 + 4 * hi;
              const half4 v0 = *reinterpret_cast<const half4*>(vp);
              const half4 v1 = *reinterpret_cast<const half4*>(vp + 8);
              vf = half8{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            }
#pragma unroll
            for (int t = 0; t < QT; ++t)
              oacc[t][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[t], oacc[t][ct], 0, 0, 0);
          }
        }
    } else {
      // ======== ORIGINAL: lane = key column, registers = 16 query rows; softmax across lanes ========
      half_t* ps = Ps + wave * (QT * 32) * PROW;
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        // Lazy running max.  The first tile seeds the running max exactly.  Every later tile forms
        // its exponent arguments against the OLD running max (in place of the scores) and takes the
        // exact row max - 16 rows x 5 cross-lane butterfly steps, the bulk of this schedule's VALU
        // work - only when some argument exceeds 2^kLazyBits; then the maxima move up by exactly
        // that excess and O, l are rescaled.  Softmax is invariant to the stabiliser; P stays
        // <= 2^kLazyBits (fp16-safe) and never underflows earlier than with the exact max of the
        // tiles seen so far would allow by more than that factor.
        constexpr float kLazyBits = 8.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (tail) {
            if (kt * KT + 2 * l31 >= a.Sk) sacc[t][0][r] = -3.0e38f;
            if (kt * KT + 2 * l31 + 1 >= a.Sk) sacc[t][1][r] = -3.0e38f;
          }
        }
        if constexpr (FIRST) {
          float mnew[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) mnew[r] = fmaxf(sacc[t][0][r], sacc[t][1][r]);
          // row max across the 32 lanes, butterfly step-major over the 16 rows: consecutive DPP ops
          // are independent, so none of them waits on the VALU->DPP hazard of its predecessor
          half_wave_max16(mnew);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            mrun[t][r] = mnew[r] * a.scale_log2;
            sacc[t][0][r] = fmaf(sacc[t][0][r], a.scale_log2, -mrun[t][r]);
            sacc[t][1][r] = fmaf(sacc[t][1][r], a.scale_log2, -mrun[t][r]);
          }
        } else {
          float amax = -3.0e38f;
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            sacc[t][0][r] = fmaf(sacc[t][0][r], a.scale_log2, -mrun[t][r]);
            sacc[t][1][r] = fmaf(sacc[t][1][r], a.scale_log2, -mrun[t][r]);
            amax = fmaxf(amax, fmaxf(sacc[t][0][r], sacc[t][1][r]));
          }
          if (__any(amax > kLazyBits)) {               // wave-uniform, rare after the first tiles
            float up[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) up[r] = fmaxf(sacc[t][0][r], sacc[t][1][r]);
            half_wave_max16(up);                       // how far each row's max moved above the old one
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float d = fmaxf(up[r], 0.f);
              const float alpha = __builtin_amdgcn_exp2f(-d);
              mrun[t][r] += d;
              lrun[t][r] *= alpha;
#pragma unroll
              for (int ct = 0; ct < DC32; ++ct) oacc[t][ct][r] *= alpha;
              sacc[t][0][r] -= d;
              sacc[t][1][r] -= d;
            }
          }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float p0 = __builtin_amdgcn_exp2f(sacc[t][0][r]);
          const float p1 = __builtin_amdgcn_exp2f(sacc[t][1][r]);
          lrun[t][r] += p0 + p1;                       // per-lane partial, reduced at the end
          const int qrow = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const half2v pp = {(half_t)p0, (half_t)p1};  // keys 2*l31, 2*l31+1
          *reinterpret_cast<half2v*>(ps + qrow * PROW + 2 * l31) = pp;
        }
      }
      __builtin_amdgcn_wave_barrier();   // P tile is wave-private; LDS ops of one wave complete in order
#pragma unroll
      for (int s4 = 0; s4 < KT / 16; ++s4) {
        half8 pf[QT];
#pragma unroll
        for (int t = 0; t < QT; ++t)
          pf[t] = *reinterpret_cast<const half8*>(ps + (t * 32 + l31) * PROW + s4 * 16 + hi * 8);
#pragma unroll
        for (int ct = 0; ct < DC32; ++ct) {
          half8 vf;
          if constexpr (VPRE) vf = vfr[s4][ct];
          else vf = *reinterpret_cast<const half8*>(vs + (ct * 32 + l31) * VROW + s4 * 16 + hi * 8);
#pragma unroll
          for (int t = 0; t < QT; ++t)
            oacc[t][ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pf[t], vf, oacc[t][ct], 0, 0, 0);
        }
      }
      __builtin_amdgcn_wave_barrier();
    }

    if constexpr (more) {
      store_tiles(buf ^ 1);
      __syncthreads();
    }
  };
  if constexpr (MODE == 0) {
    if (ntiles == 1) {
      tile_step(0, std::true_type{}, std::true_type{});
    } else {
      tile_step(0, std::true_type{}, std::false_type{});
      for (int kt = 1; kt + 1 < ntiles; ++kt) tile_step(kt, std::false_type{}, std::false_type{});
      tile_step(ntiles - 1, std::false_type{}, std::true_type{});
    }
  } else {
    for (int kt = 0; kt + 1 < ntiles; ++kt) tile_step(kt, std::false_type{}, std::false_type{});
    tile_step(ntiles - 1, std::false_type{}, std::true_type{});
  }

  // ---------------- normalise + store ----------------
  half_t* obase = a.out + (size_t)b * a.Sq * a.ldo + (size_t)h * a.d;
  if constexpr (MODE == 1) {
#pragma unroll
    for (int t = 0; t < QT; ++t) {
      const float ltot = xor32_sum(lrun[t][0]);
      const float inv = 1.0f / ltot;
      const int q = q_wave0 + t * 32 + l31;
      if (q < a.Sq) {
#pragma unroll
        for (int ct = 0; ct < DC32; ++ct)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int c = ct * 32 + 8 * g + 4 * hi;
            if (c < a.d) {
              half4 o = {(half_t)(oacc[t][ct][4 * g] * inv), (half_t)(oacc[t][ct][4 * g + 1] * inv),
                         (half_t)(oacc[t][ct][4 * g + 2] * inv), (half_t)(oacc[t][ct][4 * g + 3] * inv)};
              *reinterpret_cast<half4*>(obase + (size_t)q * a.ldo + c) = o;
            }
          }
      }
    }
  } else {
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float inv = 1.0f / half_wave_sum(lrun[t][r]);
        const int q = q_wave0 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
        if (q < a.Sq) {
#pragma unroll
          for (int ct = 0; ct < DC32; ++ct) {
            const int c = ct * 32 + l31;
            if (c < a.d) obase[(size_t)q * a.ldo + c] = (half_t)(oacc[t][ct][r] * inv);
          }
        }
      }
  }
}

template <int DK16, int DC32, int MODE, int WAVES, int QT>
void launch_one(const AttnArgs& a, int B, hipStream_t s) {
  constexpr int DKP = DK16 * 16, DCP = DC32 * 32;
  constexpr int VROW = (MODE == 0) ? KT + 8 : KT + 4;
  size_t lds = (size_t)2 * KT * (DKP + 8) * 2 + (size_t)2 * DCP * VROW * 2;
  if (MODE == 0) lds += (size_t)WAVES * QT * 32 * (KT + 8) * 2;
  auto k = attn_kernel<DK16, DC32, MODE, WAVES, QT>;
  static DynLdsOnce once;   // per instantiation, per device
  once.set(k, lds);
  AttnArgs aa = a;
  aa.q_tiles = cdiv(a.Sq, WAVES * QT * 32);
  dim3 grid(aa.q_tiles * a.heads * B);
  hipLaunchKernelGGL(k, grid, dim3(WAVES * 64), lds, s, aa);
}

template <int DK16, int DC32>
void launch_d(const AttnArgs& a, int B, int impl, hipStream_t s) {
  if (impl == kAttnOriginal) {
    launch_one<DK16, DC32, 0, 4, 1>(a, B, s);
  } else if (impl == kAttnSplitEinsum) {
    launch_one<DK16, DC32, 1, 4, 1>(a, B, s);
  } else {
    // SPLIT_EINSUM_V2: the query axis is cut into 512-query chunks (attention.py:75-86).  A chunk is
    // one 8-wave workgroup (64 queries per wave) when the launch has enough chunks to fill the 256
    // CUs; with few chunks (CFG batch 2 at 64x64: 80) each chunk is split over two 256-query or four
    // 128-query workgroups instead - same chunk arithmetic, more workgroups in flight.
    const long chunks = (long)B * a.heads * cdiv(a.Sq, 512);
    if constexpr (DC32 <= 2) {
      if (chunks >= 192) {
        launch_one<DK16, DC32, 1, 8, 2>(a, B, s);
        return;
      }
    }
    if (chunks * 2 >= 192)
      launch_one<DK16, DC32, 1, 8, 1>(a, B, s);   // also: d > 64 (register budget of two query tiles per wave)
    else
      launch_one<DK16, DC32, 1, 4, 1>(a, B, s);
  }
}

}  // namespace

bool attention_supported(int d) { return d > 0 && d % 8 == 0 && d <= 160; }

void launch_attention(const AttnDesc& d, hipStream_t s) {
  SD_REQUIRE(attention_supported(d.d), kUnsupported, "attention: head dim %d unsupported (need d %% 8 == 0, d <= 160)",
             d.d);
  SD_REQUIRE(d.ldq % 8 == 0 && d.ldk % 8 == 0 && d.ldv % 8 == 0 && d.ldo % 4 == 0, kInvalidArgument,
             "attention: leading dims must be multiples of 8 (ldq %d ldk %d ldv %d ldo %d)", d.ldq, d.ldk, d.ldv,
             d.ldo);
  SD_REQUIRE(d.ldv >= d.Sk, kInvalidArgument, "attention: ldv %d < Sk %d", d.ldv, d.Sk);
  int impl = d.impl;
  if (impl == kAttnSplitEinsumV2) {
    if (d.Sq < 512)
      impl = kAttnSplitEinsum;                        // attention.py:88-92
    else
      SD_REQUIRE(d.Sq % 512 == 0, kInvalidArgument,
                 "SPLIT_EINSUM_V2 needs S_q %% 512 == 0 (got %d); the reference would silently drop the tail "
                 "(attention.py:86)", d.Sq);
  }
  if (d.vt_perm) {   // d = 64, whole key tiles: the software-pipelined two-waves-per-SIMD kernel (attention8.hip)
    SD_REQUIRE(attention8_ok(d), kInvalidArgument, "attention: permuted V^T but the shape is not attention8's (d %d Sq %d Sk %d)", d.d,
               d.Sq, d.Sk);
    AttnDesc d8 = d;
    d8.impl = impl;
    launch_attention8(d8, s);
    return;
  }
  AttnArgs a{d.q, d.k, d.vt, d.out, d.heads, d.d, d.Sq, d.Sk, d.ldq, d.ldk, d.ldv, d.ldo,
             1.4426950408889634f / sqrtf((float)d.d)};
  const int dk16 = cdiv(d.d, 16), dc32 = cdiv(d.d, 32);
  if (dk16 == 1 && dc32 == 1) launch_d<1, 1>(a, d.B, impl, s);
  else if (dk16 == 2 && dc32 == 1) launch_d<2, 1>(a, d.B, impl, s);
  else if (dk16 == 3 && dc32 == 2) launch_d<3, 2>(a, d.B, impl, s);
  else if (dk16 == 4 && dc32 == 2) launch_d<4, 2>(a, d.B, impl, s);
  else if (dk16 == 5 && dc32 == 3) launch_d<5, 3>(a, d.B, impl, s);
  else if (dk16 == 6 && dc32 == 3) launch_d<6, 3>(a, d.B, impl, s);
  else if (dk16 == 7 && dc32 == 4) launch_d<7, 4>(a, d.B, impl, s);
  else if (dk16 == 8 && dc32 == 4) launch_d<8, 4>(a, d.B, impl, s);
  else if (dk16 == 9 && dc32 == 5) launch_d<9, 5>(a, d.B, impl, s);
  else if (dk16 == 10 && dc32 == 5) launch_d<10, 5>(a, d.B, impl, s);
  else fail(kUnsupported, "attention: no kernel for head dim %d", d.d);
  SD_HIP(hipGetLastError());
}

}  // namespace sd
