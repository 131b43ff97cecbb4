// Weight-STATIONARY 1x1 GEMM for the widest-M, shortest-K layers of the step: the GEGLU projection of the 320-channel level
// (unet.py:594-617 ff.net.0.proj with norm3 folded in: K = 320, N = 2560, M = 8192 at CFG batch 2, 65 536 at eight prompts).
//
// The tiled kernels of igemm.hip give every 64 x 128 output tile its own workgroup: 2 560 workgroups that each pull 120 KB of
// operands through L2 -> LDS for five K steps and then run the erf-GELU epilogue with the matrix pipe idle - 35 us for 13.4 GFLOP
// (0.15 of the MFMA roof; hipBLASLt takes 36 us for the plain GEMM of the shape, profiles/r06_gemm_ceiling.txt); the LDS fill
// rate of a CU, not the MFMA pipe, is the bound (LAB_NOTES Finding 5).  Here the WEIGHTS never touch LDS:
//   * a workgroup of NW waves owns NW x 32 weight rows for its whole life: wave w holds its 32 rows x K as MFMA fragments in
//     registers (K / 16 x 4 VGPRs = 80 at K = 320), fetched once, fragment-major from a pre-tiled copy (one coalesced 1-KB load
//     per fragment, like xattn_out.hip / wstream.hip);
//   * the workgroup walks row tiles of 64 tokens: the 64 x K activation tile (40 KB) arrives by LDS-DMA into a two-tile ring,
//     one tile ahead; every wave reads the whole tile as B fragments (bank swizzle on the DMA source address, as igemm.hip):
//     262 FLOP per byte filled instead of 43;
//   * the epilogue of tile t (LayerNorm fold, bias, erf-GELU gate, fp16) is VALU work issued BETWEEN the MFMAs of tile t + 1
//     (second accumulator set), its result goes through a double-buffered LDS staging tile and leaves as whole 256-B rows one
//     iteration later: one s_barrier per tile, nothing waits for a store;
//   * the row statistics of the LayerNorm fold (sum, sumsq of the fp16 inputs) are taken once per tile - 8 lanes per row - and
//     shared through LDS instead of once per wave.
// grid (N / (32 NW), workers): a column group x a strided subset of the row tiles.  Arithmetic is that of the tiled LNF / GEGLU
// kernels (fp16 operands, fp32 accumulate, out = gelu_erf(gate) * value on fp32, one rounding); summation order of the
// statistics differs, so results agree to rounding, not bit for bit.
#include "kernels.h"

#include <type_traits>

namespace sd {

namespace {

constexpr int WSG_K = 320, WSG_KS = WSG_K / 16, WSG_KC = WSG_K / 64;   // K steps of 16, 64-wide K chunks (8 KB of LDS each)
constexpr int WSG_BM = 64;
constexpr int WSG_TILE_BYTES = WSG_KC * 8192;                         // one 64-row activation tile
constexpr int WSG_PIECES = WSG_KC * 8;                                // 1-KB LDS-DMA pieces per tile

struct WsgArgs {
  const half_t* x;        // [M][K]
  const half_t* wt;       // pre-tiled: [N / 32][K / 16][64 lanes][8 halves]  (launch_wsgemm_retile)
  const float* bias;      // [N] in the DEVICE row order of the weights (GEGLU: 32 value | 32 gate interleaved)
  const float* colsum;    // [N] LayerNorm fold (same order) or null
  half_t* out;            // [M][N / 2]
  int M, N, tiles;
  float ln_eps;
  long long* prof;        // ABL == 5: s_memtime stamps of workgroup (0, 0), wave 0: [iteration][phase]
  // QKV mode (fused q|k|v, unet.py:74-84 as ONE GEMM): column groups at or beyond n_trans leave token-transposed to out_t
  // [B][N - n_trans][ldT] (V^T, vt_perm: attention8's key order), the others to out with row length n_trans; columns below q_cols
  // are multiplied by q_scale on the fp32 accumulator
  half_t* out_t;
  int n_trans, ldT, HoWo, vt_perm, q_cols;
  float q_scale;
};

__device__ __forceinline__ float wsg_gelu_erf(float x) {   // igemm.hip gelu_erf (Abramowitz-Stegun 7.1.26)
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

__device__ __forceinline__ void wsg_dma16(const __amdgpu_buffer_rsrc_t& rs, char* lds, unsigned voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voffset, soffset, 0, 0);
#endif
}

// MFMA row n of a strip (n = lane & 31 of the weight operand; the accumulator register r of the lane half `hi` that holds it is
// r = 4 (n >> 3) + (n & 3), hi = (n >> 2) & 1) -> row of the device weight matrix.  GEGLU: a strip is 16 output channels; registers
// 0-3 | 8-11 of a lane are the VALUES of channels 8 hi + 0..3 | 4..7 of the strip, registers 4-7 | 12-15 their GATES, so a lane
// ends up with eight consecutive output channels (one 16-byte store) and never needs another lane's gate.
__host__ __device__ inline int wsg_geglu_row(int strip, int n) {
  const int hi = (n >> 2) & 1, grp = n >> 3, e = n & 3;
  const int ch = strip * 16 + 8 * hi + 4 * (grp >> 1) + e;   // output channel
  return (ch >> 5) * 64 + (grp & 1) * 32 + (ch & 31);        // device layout: 32 values | 32 gates per 64 rows (upload_conv_weight)
}

// plain rows (QKV mode): a lane ends up with the 16 consecutive output channels 32 strip + 16 hi + r
__host__ __device__ inline int wsg_row(int strip, int n, bool geglu) {
  if (geglu) return wsg_geglu_row(strip, n);
  return strip * 32 + 16 * ((n >> 2) & 1) + 4 * (n >> 3) + (n & 3);
}

__global__ __launch_bounds__(256) void wsgemm_retile_kernel(const half_t* __restrict__ w, half_t* __restrict__ wt, int N, int geglu) {
  const size_t total = (size_t)(N / 32) * WSG_KS * 64;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    const size_t r = idx >> 6;
    const int ks = (int)(r % WSG_KS), strip = (int)(r / WSG_KS);
    const int row = wsg_row(strip, lane & 31, geglu != 0);
    *reinterpret_cast<half8*>(wt + idx * 8) = *reinterpret_cast<const half8*>(w + (size_t)row * WSG_K + ks * 16 + (lane >> 5) * 8);
  }
}

template <int NW, bool QKV>
struct WsgLds {
  static constexpr int OCOLS = NW * (QKV ? 32 : 16);           // output columns of the workgroup (GEGLU: half its weight rows)
  static constexpr int OROW = OCOLS + 8;                       // staged row stride in halves (+16 B: conflict-free 16-B writes)
  static constexpr int TROW = WSG_BM + 8;                      // transposed staging of a V^T tile: [OCOLS][TROW]
  static constexpr int STAGE_BYTES = (QKV && OCOLS * TROW > WSG_BM * OROW ? OCOLS * TROW : WSG_BM * OROW) * 2;
  static constexpr int A_OFF = 0;                              // [2][tile]
  static constexpr int ST_OFF = 2 * WSG_TILE_BYTES;            // [2] staged tiles
  static constexpr int STAT_OFF = ST_OFF + 2 * STAGE_BYTES;    // [2][64][2] floats: (rstd, -mean * rstd)
  static constexpr int CONST_OFF = STAT_OFF + 2 * WSG_BM * 2 * 4;   // [NW][bias | colsum][hi][16] floats: the epilogue constants of a wave
  static constexpr int BYTES = CONST_OFF + NW * 2 * 2 * 16 * 4;
};

// NW waves (NW * 32 weight rows per workgroup: NW * 16 GEGLU outputs, or NW * 32 plain columns in QKV mode); LNF: LayerNorm folded
// in (a.colsum).
// ABL (ablation builds, tools/r6_wsgemm_bench.py; results are garbage when != 0): 1 no MFMAs, 2 no erf-GELU (value * gate),
// 3 no LayerNorm statistics pass, 4 the activation tile is fetched once (no DMA in the loop), 5 timestamps, 6 no global stores
template <int NW, bool LNF, int ABL = 0, bool QKV = false>
__global__ __launch_bounds__(NW * 64) void wsgemm_geglu_kernel(WsgArgs a) {
  using L = WsgLds<NW, QKV>;
  constexpr int NT = NW * 64;
  constexpr int PPW = (WSG_PIECES + NW - 1) / NW;              // DMA pieces per wave and tile (dead ones use a zero-sized resource)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int strip = blockIdx.x * NW + wave;
  const int worker = blockIdx.y, workers = gridDim.y;
  const int T = worker < a.tiles ? (a.tiles - worker + workers - 1) / workers : 0;   // my row tiles: worker, worker + workers, ...
  if (T == 0) return;

  // ---- the wave's weights: K / 16 fragments, resident for the kernel's life ----
  half8 bf[WSG_KS];
  {
    const half_t* wp = a.wt + ((size_t)strip * WSG_KS * 64 + lane) * 8;
#pragma unroll
    for (int s = 0; s < WSG_KS; ++s) bf[s] = *reinterpret_cast<const half8*>(wp + (size_t)s * 512);
  }
  // epilogue constants of the wave's 32 weight rows (bias, LayerNorm column sum), indexed [hi][accumulator register]: kept in LDS
  // (32 VGPRs otherwise) and read back four registers at a time, just before they are used
  float* const ctab = reinterpret_cast<float*>(smem + L::CONST_OFF) + wave * 64;
  if (lane < 32) {
    const int h = lane >> 4, r = lane & 15;
    const int row = wsg_row(strip, (r & 3) + 8 * (r >> 2) + 4 * h, !QKV);
    ctab[h * 16 + r] = a.bias ? a.bias[row] : 0.f;
    ctab[32 + h * 16 + r] = LNF ? a.colsum[row] : 0.f;
  }
  // ---- LDS-DMA of the activation tiles ----
  const unsigned x_bytes = (unsigned)((size_t)a.M * WSG_K * 2);
  unsigned voff[PPW];
  int pdst[PPW];
  bool plive[PPW];
#pragma unroll
  for (int j = 0; j < PPW; ++j) {
    const int p = wave + j * NW;                               // piece: K chunk p >> 3, rows 8 (p & 7) .. + 7
    plive[j] = p < WSG_PIECES;
    const int kc = (p >> 3) % WSG_KC, pr = p & 7;
    const int r = 8 * pr + (lane >> 3);
    const int c = (lane & 7) ^ ((r >> 1) & 7);                 // logical 16-B chunk this lane fetches (bank swizzle)
    voff[j] = (unsigned)((r * WSG_K + kc * 64 + c * 8) * 2);
    pdst[j] = kc * 8192 + pr * 1024;
  }
  auto issue_tile = [&](int t) {                               // t-th tile of this workgroup -> ring slot t & 1 (t >= T: nothing)
    const bool live_t = t < T && (ABL != 4 || t < 2);
    const int m0 = (worker + t * workers) * WSG_BM;
    char* dst = smem + L::A_OFF + (t & 1) * WSG_TILE_BYTES;
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
      const __amdgpu_buffer_rsrc_t rs =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.x), 0, (int)((live_t && plive[j]) ? x_bytes : 0u), 0x00020000);
      wsg_dma16(rs, dst + pdst[j], voff[j], live_t ? m0 * WSG_K * 2 : 0);
    }
  };

  // fragment addressing inside a tile: row l31 (+ 32 for the second sub-tile), K step s: chunk s >> 2, logical 16-B chunk 2 (s & 3) + hi
  const int fsw = (l31 >> 1) & 7;
  int foff[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) foff[q] = l31 * 128 + (((q * 2 + hi) ^ fsw) * 16);

  floatx16 accA[2], accB[2];

  // LayerNorm statistics of tile t: 8 lanes per row, 5 of its 40 16-byte chunks each; (rstd, -mean * rstd) -> LDS slot t & 1
  auto stats = [&](int t) {
    if constexpr (LNF && ABL != 3) {
      const char* at = smem + L::A_OFF + (t & 1) * WSG_TILE_BYTES;
      float* st = reinterpret_cast<float*>(smem + L::STAT_OFF) + (t & 1) * WSG_BM * 2;
      const half2v one2 = {(half_t)1.f, (half_t)1.f};
      for (int slot = tid; slot < WSG_BM * 8; slot += NT) {
        const int r = slot >> 3, sub = slot & 7;
        const int phys = sub ^ ((r >> 1) & 7);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < WSG_KC; ++j) {
          const half8 v = *reinterpret_cast<const half8*>(at + j * 8192 + r * 128 + phys * 16);
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const half2v p2 = {v[2 * e], v[2 * e + 1]};
            s2 = __builtin_amdgcn_fdot2(p2, p2, s2, false);
            s1 = __builtin_amdgcn_fdot2(p2, one2, s1, false);
          }
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
          s1 += __shfl_xor(s1, o);
          s2 += __shfl_xor(s2, o);
        }
        if (sub == 0) {
          const float mean = s1 * (1.0f / WSG_K);
          const float var = fmaxf(s2 * (1.0f / WSG_K) - mean * mean, 0.f);
          const float rstd = rsqrtf(var + a.ln_eps);
          st[r * 2] = rstd;
          st[r * 2 + 1] = -rstd * mean;
        }
      }
    }
  };
  // One pipeline step, ONE basic block: the MFMAs of tile tc (into accC) with the epilogue of tile te (from accE: LayerNorm fold,
  // bias, value * gelu_erf(gate), fp16 -> staging slot te & 1) as the VALU work between them.  hipcc left to itself emits the 40
  // MFMAs first (each behind the wait for its one fragment read) and the ~400 VALU instructions of the epilogue afterwards, with
  // the matrix pipe idle - and a sched_group_barrier pattern over the whole block did not change that; a sched_barrier per K step
  // does: the fragment reads run two K steps ahead (register ring of three).
  auto step = [&](floatx16 (&accC)[2], int tc, const floatx16 (&accE)[2], int te, auto mm_c, auto ep_c) {
    constexpr bool MM = decltype(mm_c)::value, EP = decltype(ep_c)::value;
    const char* at = smem + L::A_OFF + (tc & 1) * WSG_TILE_BYTES;
    const float* st = reinterpret_cast<const float*>(smem + L::STAT_OFF) + (te & 1) * WSG_BM * 2;
    half_t* sg = reinterpret_cast<half_t*>(smem + L::ST_OFF + (te & 1) * L::STAGE_BYTES);
    half8 xf[3][2];
    auto rd = [&](int s) {
#pragma unroll
      for (int i = 0; i < 2; ++i) xf[s % 3][i] = *reinterpret_cast<const half8*>(at + (s >> 2) * 8192 + i * 4096 + foff[s & 3]);
    };
    float la[2] = {1.f, 1.f}, lb[2] = {0.f, 0.f};
    half8 o[2];
    floatx4 kb_v, kb_g, ks_v, ks_g;
    const bool tblock = QKV && blockIdx.x * L::OCOLS >= a.n_trans;   // block-uniform: a V^T column group
    const float qs = (QKV && strip * 32 < a.q_cols) ? a.q_scale : 1.f;   // wave-uniform: a query strip
    auto unit = [&](int u) {   // GEGLU: one output (value register rv, gate register rg); QKV: the two plain values of those registers
      const int h4 = u >> 3, i = (u >> 2) & 1, q = u & 3;   // h4: register octet (0-7 | 8-15), i: sub-tile
      const int rv = 8 * h4 + q, rg = rv + 4, e = 4 * h4 + q;
      float v, g;
      if constexpr (LNF) {
        v = fmaf(accE[i][rv], la[i], fmaf(lb[i], ks_v[q], kb_v[q]));
        g = fmaf(accE[i][rg], la[i], fmaf(lb[i], ks_g[q], kb_g[q]));
      } else {
        v = accE[i][rv] + kb_v[q];
        g = accE[i][rg] + kb_g[q];
      }
      if constexpr (QKV) {   // (the octet of registers 8 h4 .. 8 h4 + 7 = channels 8 h4 .. + 7 of the lane's 16; flushed per octet)
        o[i][q] = (half_t)(v * qs);
        o[i][4 + q] = (half_t)(g * qs);
      } else {
        o[i][e] = (half_t)(ABL == 2 ? v * g : v * wsg_gelu_erf(g));
      }
    };
    auto flush_qkv = [&](int h4) {   // QKV mode: the finished octet of both sub-tiles -> staging (V^T groups: [column][token])
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = i * 32 + l31;
        if (tblock) {
          half_t* tg = sg + (wave * 32 + hi * 16 + 8 * h4) * L::TROW + row;
#pragma unroll
          for (int r = 0; r < 8; ++r) tg[r * L::TROW] = o[i][r];
        } else {
          *reinterpret_cast<half8*>(sg + row * L::OROW + wave * 32 + hi * 16 + 8 * h4) = o[i];
        }
      }
    };
    auto consts = [&](int h4) {
      kb_v = *reinterpret_cast<const floatx4*>(ctab + hi * 16 + 8 * h4);
      kb_g = *reinterpret_cast<const floatx4*>(ctab + hi * 16 + 8 * h4 + 4);
      if constexpr (LNF) {
        ks_v = *reinterpret_cast<const floatx4*>(ctab + 32 + hi * 16 + 8 * h4);
        ks_g = *reinterpret_cast<const floatx4*>(ctab + 32 + hi * 16 + 8 * h4 + 4);
      }
    };
    if constexpr (MM) {
      rd(0);
      rd(1);
    }
    if constexpr (EP) {
      if constexpr (LNF) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const float2 ab = *reinterpret_cast<const float2*>(st + (i * 32 + l31) * 2);
          la[i] = ab.x;
          lb[i] = ab.y;
        }
      }
      consts(0);
    }
    const floatx16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < WSG_KS; ++s) {
      if constexpr (MM) {
        if (s + 2 < WSG_KS) rd(s + 2);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if constexpr (ABL == 1) {   // keep the operands alive, issue no MFMA
            asm volatile("" ::"v"(bf[s]), "v"(xf[s % 3][i]));
            if (s == 0) accC[i] = zero16;
          } else {
            accC[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[s], xf[s % 3][i], s == 0 ? zero16 : accC[i], 0, 0, 0);
          }
        }
      }
      if constexpr (EP) {   // 16 units over K steps 2 .. 17 (the first reads and the last MFMAs keep their slots free)
        if (s >= 2 && s < 18) {
          if (s == 10) {
            if constexpr (QKV) flush_qkv(0);
            consts(1);
          }
          unit(s - 2);
        }
      }
      // nothing moves across a K step: two fragment reads (two steps ahead), two MFMAs and one epilogue unit (~21 VALU
      // instructions, issued while the matrix pipe runs the MFMAs) per scheduling region
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (EP) {
      if constexpr (QKV) {
        flush_qkv(1);
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<half8*>(sg + (i * 32 + l31) * L::OROW + wave * 16 + hi * 8) = o[i];
      }
    }
  };
  // staged tile t -> global, whole rows: OCOLS / 8 16-byte chunks per row (V^T groups: 8 tokens per chunk, OCOLS channel rows)
  auto store_tile = [&](int t) {
    const half_t* sg = reinterpret_cast<const half_t*>(smem + L::ST_OFF + (t & 1) * L::STAGE_BYTES);
    const int m0 = (worker + t * workers) * WSG_BM;
    constexpr int CPR = L::OCOLS / 8;
    if constexpr (QKV) {
      if (blockIdx.x * L::OCOLS >= a.n_trans) {   // out_t[b][n - n_trans][s]: the 64-token tile lies inside one image
        const int NV = a.N - a.n_trans, nv0 = blockIdx.x * L::OCOLS - a.n_trans;
        const int b = m0 / a.HoWo, sp0 = m0 - b * a.HoWo;
#pragma unroll
        for (int it = 0; it < (L::OCOLS * 8) / NT; ++it) {
          const int id = tid + it * NT;
          const int r = id >> 3, c = id & 7;
          half8 v;
          if (a.vt_perm) {   // chunk c = tokens 16 j + 4 o + {0..3} and 16 j + 8 + 4 o + {0..3}  (j = c >> 1, o = c & 1)
            const half_t* src = sg + r * L::TROW + (c >> 1) * 16 + (c & 1) * 4;
            const half4 lo = *reinterpret_cast<const half4*>(src), up = *reinterpret_cast<const half4*>(src + 8);
            v = half8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
          } else {
            v = *reinterpret_cast<const half8*>(sg + r * L::TROW + c * 8);
          }
          if (m0 + c * 8 < a.M) *reinterpret_cast<half8*>(a.out_t + ((size_t)b * NV + nv0 + r) * a.ldT + sp0 + c * 8) = v;
        }
        return;
      }
    }
    const int NO = QKV ? a.n_trans : (a.N >> 1);
#pragma unroll
    for (int it = 0; it < (WSG_BM * CPR) / NT; ++it) {
      const int id = tid + it * NT;
      const int r = id / CPR, c = id - r * CPR;
      const half8 v = *reinterpret_cast<const half8*>(sg + r * L::OROW + c * 8);
      if (m0 + r < a.M && ABL != 6) out_store(reinterpret_cast<half8*>(a.out + (size_t)(m0 + r) * NO + (size_t)blockIdx.x * L::OCOLS + c * 8), v);
    }
  };
  static_assert((WSG_BM * (L::OCOLS / 8)) % NT == 0 && (L::OCOLS * 8) % NT == 0, "whole store rounds");
  auto top = [&]() {   // my DMA pieces (and older stores) are done; then everybody's: the tile is readable, the other slot is free
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (lgkmcnt: the staging / statistics / table writes)
  };

  int pi = 0;
  auto stamp = [&](int ph) {
    if constexpr (ABL == 5) {
      if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && pi < 8) a.prof[pi * 8 + ph] = (long long)__builtin_readcyclecounter();
    }
  };
  // ---- software pipeline over the workgroup's T tiles: iteration t = [barrier | DMA t+1 | store t-2 | stats t | MFMA t || epilogue t-1]
  using Tt = std::true_type;
  using Ff = std::false_type;
  issue_tile(0);
  top();                                                       // tile 0 landed, the constant tables are visible
  issue_tile(1);
  stats(0);
  step(accA, 0, accB, 0, Tt{}, Ff{});
  int t = 1;
  for (; t + 1 < T; t += 2) {                                  // two tiles per trip: the accumulator sets alternate statically
    stamp(0);
    top();
    stamp(1);
    issue_tile(t + 1);
    if (t >= 2) store_tile(t - 2);
    stamp(2);
    stats(t);
    stamp(3);
    step(accB, t, accA, t - 1, Tt{}, Tt{});
    stamp(4);
    ++pi;
    stamp(0);
    top();
    stamp(1);
    issue_tile(t + 2);
    store_tile(t - 1);
    stamp(2);
    stats(t + 1);
    stamp(3);
    step(accA, t + 1, accB, t, Tt{}, Tt{});
    stamp(4);
    ++pi;
  }
  if (t < T) {                                                 // one tile left: it goes to B, the pending epilogue is A's
    top();
    if (t >= 2) store_tile(t - 2);
    stats(t);
    step(accB, t, accA, t - 1, Tt{}, Tt{});
    top();
    store_tile(t - 1);
    step(accA, 0, accB, t, Ff{}, Tt{});
    ++t;
  } else {                                                     // (T tiles done, the last one sits in A)
    top();
    if (t >= 2) store_tile(t - 2);
    step(accB, 0, accA, t - 1, Ff{}, Tt{});
  }
  top();
  store_tile(T - 1);
}

template <int NW, bool LNF, int ABL = 0, bool QKV = false>
void launch_wsg(const WsgArgs& a, int workers, hipStream_t s) {
  auto k = wsgemm_geglu_kernel<NW, LNF, ABL, QKV>;
  constexpr size_t lds = WsgLds<NW, QKV>::BYTES;
  static DynLdsOnce once;
  once.set(k, lds);
  hipLaunchKernelGGL(k, dim3(a.N / (32 * NW), workers), dim3(NW * 64), lds, s, a);
}

}  // namespace

// K = 320, single source, no residual / timestep embedding, a workgroup walking several row tiles: the GEGLU projection (kOutGeglu,
// N a multiple of 256: eight waves per workgroup; LayerNorm fold optional) or the fused q|k|v projection (out_t set, LayerNorm fold,
// N and the q|k / v boundary multiples of 160: five waves per workgroup, 64-token tiles inside one image).
bool wsgemm_shape_ok(const ConvDesc& d) {
  if (d.ksize != 1 || d.stride != 1 || d.up != 1 || d.x1 || d.C0 != WSG_K) return false;
  if (d.res || d.temb || d.gn_partial || d.gnf_partial || d.n_twins || !d.bias) return false;
  if ((long)d.B * d.Ho * d.Wo < 2048) return false;
  if (d.out_t)
    return d.out_mode == kOutHalf && d.ln_colsum && !d.debug && d.N % 160 == 0 && d.n_trans > 0 && d.n_trans % 160 == 0 &&
           (d.Ho * d.Wo) % WSG_BM == 0 && d.ldT % 8 == 0 && d.q_cols % 32 == 0;
  return d.out_mode == kOutGeglu && d.N % 256 == 0;
}

// The library's own rule: the GEGLU projection always (stand-alone 31.5 -> 27.5 us at M = 8 192, 272 -> 157 at M = 65 536); the
// fused q|k|v only on request (SD_WSGEMM_QKV=1 with SD_TUNE): five waves x 4 tiles per workgroup run 21.2 us where the tiled
// kernel takes 17.0 at M = 8 192 and tie it at M = 65 536 (104 vs 107 us) - the per-tile issue latency of Finding 16 again.
bool wsgemm_wanted(const ConvDesc& d) {
  if (!wsgemm_shape_ok(d)) return false;
  static const bool qkv = tune_env_int("SD_WSGEMM_QKV", 0) != 0;
  return d.out_t == nullptr || qkv;
}

size_t wsgemm_tiled_halves(int N) { return (size_t)N * WSG_K; }

void launch_wsgemm_retile(const half_t* w, half_t* wt, int N, bool geglu, hipStream_t s) {
  SD_REQUIRE(N % 32 == 0, kInvalidArgument, "wsgemm retile: N=%d", N);
  const size_t total = (size_t)N * WSG_K / 8;
  hipLaunchKernelGGL(wsgemm_retile_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 2048)), dim3(256), 0, s, w, wt, N, geglu ? 1 : 0);
  SD_HIP(hipGetLastError());
}

void launch_wsgemm(const ConvDesc& d, hipStream_t s) {
  SD_REQUIRE(wsgemm_shape_ok(d) && d.w_ws, kInvalidArgument, "wsgemm: shape not eligible (C0=%d N=%d mode=%d)", d.C0, d.N, d.out_mode);
  WsgArgs a{};
  a.x = d.x0;
  a.wt = d.w_ws;
  a.bias = d.bias;
  a.colsum = d.ln_colsum;
  a.out = d.out;
  a.M = d.B * d.Ho * d.Wo;
  a.N = d.N;
  a.tiles = cdiv(a.M, WSG_BM);
  a.ln_eps = d.ln_eps;
  const bool qkv = d.out_t != nullptr;
  a.out_t = d.out_t;
  a.n_trans = qkv ? d.n_trans : 0x7fffffff;
  a.ldT = d.ldT;
  a.HoWo = d.Ho * d.Wo;
  a.vt_perm = qkv ? d.vt_perm : 0;
  a.q_cols = qkv ? d.q_cols : 0;
  a.q_scale = d.q_scale;
  const int groups = a.N / (qkv ? 160 : 256);
  // one workgroup per CU (105-125 KB of LDS): as many row workers per column group as the 256 CUs give, never more than the tiles
  int workers = std::max(1, std::min(a.tiles, 256 / groups));
  // equal trip counts beat a ragged last round: the smallest worker count with the same number of rounds
  const int rounds = cdiv(a.tiles, workers);
  workers = cdiv(a.tiles, rounds);
  a.prof = d.prof;
  if (qkv) {
    launch_wsg<5, true, 0, true>(a, workers, s);
  } else if (d.debug) {   // ablation builds of the LayerNorm-folded form (measurement only)
    SD_REQUIRE(d.ln_colsum && d.debug >= 1 && d.debug <= 6 && (d.debug != 5 || d.prof), kInvalidArgument, "wsgemm ablation %d", d.debug);
    switch (d.debug) {
      case 1: launch_wsg<8, true, 1>(a, workers, s); break;
      case 2: launch_wsg<8, true, 2>(a, workers, s); break;
      case 3: launch_wsg<8, true, 3>(a, workers, s); break;
      case 4: launch_wsg<8, true, 4>(a, workers, s); break;
      case 5: launch_wsg<8, true, 5>(a, workers, s); break;
      default: launch_wsg<8, true, 6>(a, workers, s); break;
    }
  } else if (d.ln_colsum) launch_wsg<8, true>(a, workers, s);
  else launch_wsg<8, false>(a, workers, s);
  SD_HIP(hipGetLastError());
}

}  // namespace sd
