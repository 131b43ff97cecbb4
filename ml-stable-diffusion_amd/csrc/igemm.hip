// K4/K5 - implicit-GEMM convolution (3x3 / 1x1, stride 1/2, fused nearest-x2 gather, fused
// channel-concat) on gfx950 MFMA (v_mfma_f32_32x32x16_f16), fp16 in / fp32 accumulate.
//
// Math (reference): nn.Conv2d calls of python_coreml_stable_diffusion/unet.py:74-84 (q/k/v/out
// 1x1), :435-468 (resnet 3x3 + 1x1 shortcut), :496-510 (up/down-sample), :533-551 (proj_in/out),
// :601-617 (GEGLU feed-forward), and the torch.cat([h, skip], dim=1) of :213-216 which we never
// materialise (second K-source).
//
// Mapping: out[m][n] = sum_k X[m][k] W[n][k]; m = (b, oy, ox) output pixel, n = output channel,
// k = tap * Ctot + c.  Both operands are K-contiguous in HBM (NHWC activations, [N][taps][C]
// weights re-laid-out at load), which is exactly the MFMA A/B fragment shape (8 consecutive k
// per lane), so tiles go HBM -> VGPR (16-B coalesced) -> LDS (padded rows, conflict-free
// ds_read_b128) -> MFMA with a register-staged double buffer (one barrier per 64-deep K step).
// The tile is computed TRANSPOSED (rows = n, cols = m) so each lane owns 4 consecutive output
// channels of one pixel: bias / timestep-embedding / residual adds and the fp16 store are 8-byte
// vector accesses with no cross-lane traffic.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>

#include "kernels.h"

namespace sd {

namespace {

constexpr int BK = 64;            // K step (halves)
constexpr int LDS_ROW = BK + 8;   // padded row stride in halves (144 B): 16 rows -> 16 distinct 16-B slots

struct IgemmArgs {
  const half_t* x0;
  const half_t* x1;
  const half_t* w;
  const float* bias;
  const float* temb;
  const half_t* res;
  half_t* out;
  float* partial;
  int C0, C1, Ctot;
  int B, Hi, Wi, Ho, Wo, HoWo;
  int ksize, stride, up, pad;
  int M, N, K;
  int temb_stride;
  int nk_total, nk_per_split, splitk;
  int slab;   // 1: the tile leaves as an fp32 slab of a.partial (split-K, or a forced slab for reduce_twin_kernel), no epilogue
  int out_mode, ldT;
  int debug;   // ablation (microbench only): bits 0-1: 1 = loads+barriers only, 2 = compute only; bit 2: timestamps
  long long* prof;
  const half_t* zeros;   // >= 16 B of zeros: source of padding / out-of-range rows
  int tiles_x, tiles_y;  // halo kernel: 8x16-pixel output tiles per image
  // LayerNorm folded into a 1x1 GEMM (LNF kernels): w already carries gamma, bias carries W.beta,
  // colsum[n] = sum_k w[n][k]; the kernel accumulates the row statistics of its A tile on the fly
  const float* ln_colsum;
  float ln_eps;
  // fused q|k|v projection: output columns >= n_trans go, token-transposed, to out_t [B][N-n_trans][ldT]
  // (the V^T operand of attention); columns below it to out with row length ldo
  int n_trans, ldo;
  half_t* out_t;
  int vt_perm;   // 1: out_t rows leave with the two middle 4-token blocks of every 16 tokens swapped (AttnDesc::vt_perm)
  int res_pre;   // 1: igemm_kernel fetches its residual tile at kernel entry (SD_RES_PREFETCH=0 switches it off, A/B)
  // GroupNorm statistics of the OUTPUT tensor from this kernel's epilogue (the consumer is torch.nn.GroupNorm of
  // unet.py:430-451 / :528-531): per (sample, group, m-tile) partial (sum, sumsq) of the fp16-rounded outputs, written to
  // gn_partial [B][G][kGnMaxSlabs][2] at entry mt * 2 + slot (slot 1: the part of a group that began in the previous n-tile).
  // Every entry < 2 * gn_T is written by exactly one workgroup per launch (no atomics: the replay stays bit-reproducible).
  float* gn_partial;
  int gn_G, gn_cpg, gn_T;   // groups, channels per group, m-tiles per sample
  // Tile order inside an XCD's contiguous run of workgroup ids.  0: m fastest - consecutive workgroups share a WEIGHT panel
  // (right when the weights outweigh the activations: the 8x8 / 16x16 levels at small batch).  1: n fastest - consecutive
  // workgroups share an ACTIVATION panel, so each XCD pulls its rows through the fabric once and the other n-tiles hit its L2
  // (round 2 measured 27 MB of fabric reads for a 320->320 GEMM at M = 8192 with 10.6 MB of operands: every n-tile of a row
  // block ran on a different XCD).  Chosen per launch from the operand sizes (launch_conv).
  int n_fast;
  // GroupNorm folded into a 1x1 GEMM (gemm_pipe_kernel GNF): partial (sum, sumsq) entries of the input's producer, affine, eps
  const float* gnf_partial;
  const float* gnf_gamma;
  const float* gnf_beta;
  float gnf_eps;
  int gnf_G, gnf_entries;
  // fused q|k|v: columns [0, q_cols) leave multiplied by q_scale in fp32 (ConvDesc::q_scale); q_cols = 0: off
  float q_scale;
  int q_cols;
};

constexpr int kGnScratchFloats = 256 * 17;   // per-thread (sum[8], sumsq[8]) of the epilogue's store loop, +1 pad

// Per-tile GroupNorm statistics from the store loop of an epilogue.  Thread t owns the 8-channel chunk (t % (BN/8)) of the
// rows it stored; fs / fq are its sums over those rows.  scratch: kGnScratchFloats + 2 * BN floats of LDS that no thread
// reads or writes any more.  Fixed-order reductions only.
template <int BN>
__device__ __forceinline__ void tile_gn_stats(const IgemmArgs& a, float* scratch, const float (&fs)[8], const float (&fq)[8],
                                              int n_blk, int b, int mt) {
  constexpr int OWC = BN / 8;
  const int tid = threadIdx.x;
  float* chan = scratch + kGnScratchFloats;   // [2][BN] per-channel sum | sumsq of this tile
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    scratch[tid * 17 + e] = fs[e];
    scratch[tid * 17 + 8 + e] = fq[e];
  }
  __syncthreads();
  for (int cc = tid; cc < 2 * BN; cc += 256) {
    const int which = cc / BN, ch = cc - which * BN;
    const int cl = ch >> 3, e = ch & 7;
    float s = 0.f;
#pragma unroll 4
    for (int k = 0; k < 256 / OWC; ++k) s += scratch[(k * OWC + cl) * 17 + which * 8 + e];
    chan[cc] = s;
  }
  __syncthreads();
  const int g0 = n_blk / a.gn_cpg;
  const int g = g0 + tid;
  const int n_end = min(n_blk + BN, a.N);
  if (tid < BN && g < a.gn_G && g * a.gn_cpg < n_end) {
    const int gs = g * a.gn_cpg, ge = gs + a.gn_cpg;
    const int lo = max(gs, n_blk), hi = min(ge, n_end);
    float s = 0.f, q = 0.f;
    for (int c = lo; c < hi; ++c) {
      s += chan[c - n_blk];
      q += chan[BN + c - n_blk];
    }
    float* dst = a.gn_partial + (((size_t)b * a.gn_G + g) * kGnMaxSlabs + (size_t)mt * 2) * 2;
    if (gs < n_blk) {            // the group began in the previous n-tile, which wrote slot 0
      dst[2] = s;
      dst[3] = q;
    } else {
      dst[0] = s;
      dst[1] = q;
      if (ge <= n_end) {         // the group ends inside this tile: nobody else writes slot 1
        dst[2] = 0.f;
        dst[3] = 0.f;
      }
    }
  }
}

// exact-GELU (erf form, unet.py:613-617 via F.gelu) with erf from Abramowitz-Stegun 7.1.26:
// |erf error| < 6.1e-7 in fp32, |gelu error| < 3.7e-7 absolute and < 1.7e-4 relative wherever
// |gelu| > 1e-3 - below the fp16 rounding of the output - at a third of the VALU cost of the
// libm erff: with K only 320-1280 deep the epilogue is a visible share of a GEGLU GEMM
// (measured 610 -> 501 us per UNet step over the 16 GEGLU launches, tools/geglu_bench.py).
__device__ __forceinline__ float gelu_erf(float x) {
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

// v + (value of lane ^ 32): one v_permlane32_swap instead of a ds_bpermute round trip
__device__ __forceinline__ float xor32_sum(float v) {
  const unsigned u = __float_as_uint(v);
  const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned r0 = r[0], r1 = r[1];   // scalars first: bit-casting the vector-element lvalue reads lane 0 twice
  return __uint_as_float(r0) + __uint_as_float(r1);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
#include "gn_body.inc"   // groupnorm_apply_body / groupnorm_fused_body (the stand-alone launches are in norm.hip)

// buffer_load_dwordx4 ... lds (16 bytes per lane straight into LDS; M0 carries the wave-uniform LDS base).  hipcc's HOST pass
// checks the 16-byte form against a target without the gfx950 feature and then silently drops the enclosing kernel's stub,
// so the builtin is only visible to the device pass.
__device__ __forceinline__ void dma16_to_lds(const __amdgpu_buffer_rsrc_t& rs, char* lds, unsigned voffset, int soffset) {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voffset, soffset, 0, 0);
#endif
}

// Tile epilogue shared by the GEMM kernels: acc[i][j] is the 32x32 block (pixel block (wm*TM+i), channel block (wn*TN+j)) of
// a BM x BN tile owned by wave (wm, wn) of a WGM x WGN wave grid, in the transposed MFMA layout
// n = n0 + (r&3) + 8*(r>>2) + 4*hi ; m = m0 + (lane&31).  Split-K slabs, or bias / timestep embedding / LayerNorm fold /
// GEGLU / residual / fused q|k|v write-out staged through LDS (`smem` is free: the caller's K loop is over and every wave
// has passed a barrier after its last fragment read - this function starts with its own barrier for that).
// resv (NRES > 0, use_resv): the residual chunks of the final store loop, fetched by the caller at kernel entry - read here, a
// residual tile costs every workgroup one exposed memory round trip after its K loop.
template <int BM, int BN, int WGM, int WGN, int TM, int TN, bool LNF, int NRES = 0>
__device__ __forceinline__ void tile_epilogue(const IgemmArgs& a, floatx16 (&acc)[TM][TN], const float (&ln_a)[TM],
                                              const float (&ln_b)[TM], char* smem, float* sconst, float const_b, float const_t,
                                              float const_c, int m_blk, int n_blk, int wave, int split, bool temb_uniform,
                                              const half8* resv = nullptr, bool use_resv = false) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wm = wave / WGN, wn = wave % WGN;
  const int frow = lane & 31, hi = lane >> 5;
  // acc[i][j][r]: n = n0 + (r&3) + 8*(r>>2) + 4*hi ; m = m0 + (lane&31)
  if (a.slab) {   // fp32 partial slabs; bias/temb/residual are applied by splitk_reduce_kernel / reduce_twin_kernel
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m_blk + (wm * TM + i) * 32 + frow;
      if (m >= a.M) continue;
      float* prow = a.partial + ((size_t)split * a.M + m) * a.N;
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n_blk + (wn * TN + j) * 32 + 8 * q + 4 * hi;
          if (n < a.N) {
            floatx4 v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            out_store(reinterpret_cast<floatx4*>(prow + n), v);
          }
        }
    }
  } else {
    // Stage the finished tile through LDS (free after the K loop) so that the global stores - and
    // the residual loads - are whole 16-B-per-lane row segments instead of 32 scattered 16-B pieces
    // per instruction (the scattered form cost ~11k cycles per 128x128 tile, prof_conv).
    const bool geglu = a.out_mode == kOutGeglu;
    constexpr int OW = BN;                 // staged tile width in halves (GEGLU uses the first BN/2)
    constexpr int OROW = OW + 8;           // +16 B pad: conflict-free 16-B reads
    constexpr int TROW = BM + 8;           // transposed staging (fused q|k|v: the V^T columns), [BN][TROW]
    half_t* ot = reinterpret_cast<half_t*>(smem);   // [BM][OROW] or [BN][TROW]  (<= the K-loop buffers)
    const bool tblock = n_blk >= a.n_trans;         // block-uniform
    if (tid < BN) {
      sconst[tid] = const_b + const_t;
      if constexpr (LNF) sconst[BN + tid] = const_c;
    }
    __syncthreads();                       // every wave is done with its last fragment reads; sconst is visible
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int ml = (wm * TM + i) * 32 + frow;
      const int m = m_blk + ml;
      const int b = (m < a.M) ? m / a.HoWo : 0;
      if (geglu) {
        if constexpr (TN % 2 == 0) {
#pragma unroll
          for (int j = 0; j < TN; j += 2)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int nl = (wn * TN + j) * 32 + 8 * q + 4 * hi;          // value rows (interleaved W)
              half4 o;
              const floatx4 bv4 = *reinterpret_cast<const floatx4*>(sconst + nl);        // 0 beyond N
              const floatx4 bg4 = *reinterpret_cast<const floatx4*>(sconst + nl + 32);
              floatx4 cv4 = {0.f, 0.f, 0.f, 0.f}, cg4 = {0.f, 0.f, 0.f, 0.f};
              if constexpr (LNF) {
                cv4 = *reinterpret_cast<const floatx4*>(sconst + BN + nl);
                cg4 = *reinterpret_cast<const floatx4*>(sconst + BN + nl + 32);
              }
#pragma unroll
              for (int e = 0; e < 4; ++e) {
                float v, g;
                if constexpr (LNF) {
                  v = fmaf(acc[i][j][4 * q + e], ln_a[i], fmaf(ln_b[i], cv4[e], bv4[e]));
                  g = fmaf(acc[i][j + 1][4 * q + e], ln_a[i], fmaf(ln_b[i], cg4[e], bg4[e]));
                } else {
                  v = acc[i][j][4 * q + e] + bv4[e];
                  g = acc[i][j + 1][4 * q + e] + bg4[e];
                }
                o[e] = (half_t)(v * gelu_erf(g));
              }
              *reinterpret_cast<half4*>(ot + ml * OROW + (wn * TN + j) * 16 + 8 * q + 4 * hi) = o;
            }
        }
      } else {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int nl = (wn * TN + j) * 32 + 8 * q + 4 * hi;
            const int n = n_blk + nl;
            float v[4] = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
            const floatx4 bb = *reinterpret_cast<const floatx4*>(sconst + nl);            // bias (+ temb), 0 beyond N
            if constexpr (LNF) {
              const floatx4 cs = *reinterpret_cast<const floatx4*>(sconst + BN + nl);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], ln_a[i], fmaf(ln_b[i], cs[e], bb[e]));
            } else {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += bb[e];
            }
            if (n < a.q_cols) {   // queries for attention8: softmax scale and log2(e) before the rounding to fp16
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] *= a.q_scale;
            }
            if (a.temb && !temb_uniform && n < a.N) {   // tile straddles samples (HoWo < BM): per-row sample index
              floatx4 tt = *reinterpret_cast<const floatx4*>(a.temb + (size_t)b * a.temb_stride + n);
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += tt[e];
            }
            half4 o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
            if (tblock) {   // V^T columns: staged [n][m] so the write-out rows are token-contiguous
#pragma unroll
              for (int e = 0; e < 4; ++e) ot[(nl + e) * TROW + ml] = o[e];
            } else {
              *reinterpret_cast<half4*>(ot + ml * OROW + nl) = o;
            }
          }
      }
    }
    __syncthreads();
    if (tblock) {   // out_t[b][n - n_trans][s]: 8 consecutive tokens of one image per 16-B store
      const int NV = a.N - a.n_trans;
      for (int idx = tid; idx < BN * (BM / 8); idx += 256) {
        const int r = idx / (BM / 8), c = idx - r * (BM / 8);
        const int nv = n_blk + r - a.n_trans, m = m_blk + c * 8;
        if (nv < NV && m < a.M) {
          const int b = m / a.HoWo, sp = m - b * a.HoWo;
          half8 v;
          if (a.vt_perm) {   // 16-B chunk c of the row = tokens 16 j + 4 o + {0..3} and 16 j + 8 + 4 o + {0..3}  (j = c >> 1, o = c & 1)
            const half_t* src = ot + r * TROW + (c >> 1) * 16 + (c & 1) * 4;
            const half4 lo = *reinterpret_cast<const half4*>(src), up = *reinterpret_cast<const half4*>(src + 8);
            v = half8{lo[0], lo[1], lo[2], lo[3], up[0], up[1], up[2], up[3]};
          } else {
            v = *reinterpret_cast<const half8*>(ot + r * TROW + c * 8);
          }
          out_store(reinterpret_cast<half8*>(a.out_t + ((size_t)b * NV + nv) * a.ldT + sp), v);
        }
      }
      return;
    }
    const int NO = geglu ? (a.N >> 1) : a.ldo;            // output row length
    const int nb0 = geglu ? (n_blk >> 1) : n_blk;         // first output column of this tile
    constexpr int OWC = OW / 8;                           // 16-B chunks per staged row (GEGLU: first half used)
    const int wc = geglu ? OWC / 2 : OWC;
    // GroupNorm statistics of what this tile stores (block-uniform; launch_conv checks: rows of one sample, N % 8 == 0)
    const bool gn = a.gn_partial != nullptr && !geglu;
    float fs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, fq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool stored = false;
    if constexpr (NRES > 0) {
      if (use_resv) {   // (never GEGLU: wc == OWC, BM * OWC == NRES * 256)
#pragma unroll
        for (int it = 0; it < NRES; ++it) {
          const int idx = tid + it * 256;
          const int r = idx / OWC, c = idx - r * OWC;
          const int m = m_blk + r, n = nb0 + c * 8;
          if (m < a.M && n < NO) {
            half8 v = *reinterpret_cast<const half8*>(ot + r * OROW + c * 8);
            half_t* dst = a.out + (size_t)m * NO + n;
            if (n + 8 <= NO) {
              const half8 rr = resv[it];
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)rr[e]);
              out_store(reinterpret_cast<half8*>(dst), v);
              if (gn) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                  const float f = (float)v[e];
                  fs[e] += f;
                  fq[e] = fmaf(f, f, fq[e]);
                }
              }
            } else {
              for (int e = 0; e < NO - n; ++e) dst[e] = (half_t)((float)v[e] + (float)a.res[(size_t)m * NO + n + e]);
            }
          }
        }
        stored = true;
      }
    }
    if (!stored) {
      for (int idx = tid; idx < BM * wc; idx += 256) {
        const int r = idx / wc, c = idx - r * wc;
        const int m = m_blk + r, n = nb0 + c * 8;
        if (m < a.M && n < NO) {
          half8 v = *reinterpret_cast<const half8*>(ot + r * OROW + c * 8);
          half_t* dst = a.out + (size_t)m * NO + n;
          if (n + 8 <= NO) {
            if (a.res) {
              const half8 rr = *reinterpret_cast<const half8*>(a.res + (size_t)m * NO + n);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)rr[e]);
            }
            out_store(reinterpret_cast<half8*>(dst), v);
            if (gn) {
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                const float f = (float)v[e];
                fs[e] += f;
                fq[e] = fmaf(f, f, fq[e]);
              }
            }
          } else {   // ragged last chunk (N % 8 == 4)
            for (int e = 0; e < NO - n; ++e) dst[e] = a.res ? (half_t)((float)v[e] + (float)a.res[(size_t)m * NO + n + e]) : v[e];
          }
        }
      }
    }
    if (gn) {   // scratch behind the staged tile (launch_variant sizes the LDS for it)
      float* scratch = reinterpret_cast<float*>(smem + (((size_t)BM * OROW * sizeof(half_t) + 15) & ~(size_t)15));
      tile_gn_stats<BN>(a, scratch, fs, fq, n_blk, m_blk / a.HoWo, (m_blk % a.HoWo) / BM);
    }
  }
}

// One 256-thread workgroup = 4 wavefronts laid out WGM x WGN over a BM x BN tile.
// GLDS = true: tiles go HBM -> LDS directly (global_load_lds_dwordx4, no VGPR round trip and no
// ds_write pass: the write pass was ~60 % of the LDS-pipe time of the register-staged version).
// The DMA writes lane-linear 1-KiB pieces (8 rows x 128 B), so the bank swizzle lives on the
// per-lane SOURCE address: physical 16-B chunk p of row r holds logical chunk p ^ ((r >> 1) & 7),
// and fragment reads apply the same XOR (conflict-free ds_read_b128, cdna guide rule 21).
// KG = 2 (round 5): IN-WORKGROUP split-K.  The small-M GEMMs of the 16x16 / 8x8 levels (1280 -> 1280 at M = 512: 160 tiles of
// 64 x 64, 20 K steps each) expose their serial K loop - 0.45 us per barrier-separated step - on 160 of 256 CUs; a split-K over
// workgroups costs a dependent reduce launch (measured: it loses).  Here a workgroup is TWO groups of four waves, each with its
// own LDS ring, walking one half of the K range in lockstep (the s_barrier is workgroup-wide; the shorter half pads with one
// barrier-only step); group 1 hands its accumulators over through LDS and retires (s_barrier only waits for surviving
// waves), group 0 adds them and runs the epilogue.  Same loop body, half as many steps, no HBM slabs, no extra launch.
template <int BM, int BN, int WGM, int WGN, bool TRANS_OUT, bool GLDS, int NST, int DBG = 0, bool LNF = false, int KG = 1>
__global__ __launch_bounds__(256 * KG) void igemm_kernel(IgemmArgs a) {
  static_assert(WGM * WGN == 4, "4 waves");
  static_assert(KG == 1 || (KG == 2 && GLDS && NST >= 3 && !LNF && !TRANS_OUT && DBG == 0), "in-workgroup split-K: ring variants of the plain epilogue");
  static_assert(!(LNF && TRANS_OUT), "LayerNorm fold uses the in-lane row layout of the non-transposed tile");
  constexpr int TM = BM / WGM / 32;   // 32x32 MFMA tiles per wave along m
  constexpr int TN = BN / WGN / 32;
  constexpr int XR = BM / 32;         // 16-B chunks each thread stages per K step (X tile)
  constexpr int WR = BN / 32;
  constexpr int ROW = GLDS ? BK : LDS_ROW;   // LDS row stride in halves

  // DBG (compile-time, microbench-only instantiations): bits 0-1: 1 = loads+barriers only, 2 = compute only;
  // bit 2: phase timestamps; bit 3: no LDS reads; bit 4: no MFMAs.  Production kernels have DBG == 0.
  const bool prof = (DBG & 4) && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0;
  long long prof_t[4];
  long long prof_w0 = 0;
  if (prof) { prof_t[0] = clock64(); prof_w0 = wall_clock64(); }
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int kg = KG > 1 ? __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 8) : 0;   // K group of this wave
  half_t* Xs = reinterpret_cast<half_t*>(smem) + (size_t)kg * NST * (BM + BN) * ROW;   // [NST][BM][ROW] of this K group
  half_t* Ws = Xs + NST * BM * ROW;                             // [NST][BN][ROW]
  // behind the ring: per-column epilogue constants [BN] bias (+ timestep-embedding row when the tile lies in
  // one sample) | [BN] LayerNorm colsum.  Loaded once per workgroup next to the first tile's DMA, so the
  // epilogue reads them from LDS instead of issuing dependent global loads per accumulator fragment
  // (those cost the 128x128 tile ~9k cycles, profiles/r01_prof_conv_phases.log).
  float* sconst = reinterpret_cast<float*>(smem + (size_t)KG * NST * (BM + BN) * ROW * sizeof(half_t));

  const int tid = threadIdx.x & 255;   // thread inside its K group
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform (LDS-DMA base -> M0)
  const int wm = wave / WGN, wn = wave % WGN;

  // XCD-aware tile order: the dispatcher round-robins consecutive block ids over the 8 XCDs
  // (MI355X_MICROARCH: block b -> XCD b % 8).  Remap so that each XCD walks a contiguous run of
  // tiles that share the same weight panel (n-tile) -> the panel stays in that XCD's L2.
  int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
  int nwg = nbm * nbn;
  int bid = blockIdx.x;
  {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    bid = base + idx;   // bijective for any nwg
  }
  const int bn_idx = a.n_fast ? bid % nbn : bid / nbm, bm_idx = a.n_fast ? bid / nbn : bid % nbm;   // IgemmArgs::n_fast
  const int m_blk = bm_idx * BM, n_blk = bn_idx * BN;

  const int split = blockIdx.y;
  int kt_begin = split * a.nk_per_split;
  int kt_end = kt_begin + a.nk_per_split;
  if (kt_end > a.nk_total) kt_end = a.nk_total;
  int kg_steps = kt_end - kt_begin;   // barrier-separated steps every K group goes through (KG == 2: the longer half)
  if constexpr (KG == 2) {
    const int half0 = (kt_end - kt_begin + 1) >> 1;
    kg_steps = half0;
    if (kg == 0) kt_end = kt_begin + half0;
    else kt_begin += half0;
  }

  // ---- per-thread staging coordinates (loop invariant) ----
  const int pchunk = tid & 7;       // physical 16-B slot of the 128-B K row this thread fills
  const int lrow = tid >> 3;        // 0..31  (= wave * 8 + lane / 8)
  int x_pix[XR];                    // b*Hi*Wi (or -1 when the row is past M)
  int x_iy0[XR], x_ix0[XR];
#pragma unroll
  for (int i = 0; i < XR; ++i) {
    int m = m_blk + lrow + 32 * i;
    if (m < a.M) {
      int b = m / a.HoWo;
      int rem = m - b * a.HoWo;
      int oy = rem / a.Wo;
      int ox = rem - oy * a.Wo;
      x_pix[i] = b * a.Hi * a.Wi;
      x_iy0[i] = oy * a.stride - a.pad;
      x_ix0[i] = ox * a.stride - a.pad;
    } else {
      x_pix[i] = -1;
      x_iy0[i] = 0;
      x_ix0[i] = 0;
    }
  }
  const int Hup = a.Hi * a.up, Wup = a.Wi * a.up;
  const int upshift = a.up >> 1;    // up in {1,2}
  // logical chunk this thread fetches for its rows: rows lrow+32*i share (row >> 1) & 7 == (lrow >> 1) & 7
  const int chunk = GLDS ? (pchunk ^ ((lrow >> 1) & 7)) : pchunk;

  half8 xr[GLDS ? 1 : XR], wr[GLDS ? 1 : WR];

  // K-tile iterator.  Tiles are visited in order, so every staged row just walks a pointer:
  // per K step the loader is one 64-bit add per row.  The im2col source (3x3 tap shift, zero
  // padding, nearest-x2 upsample, which of the two concat sources) is re-derived only when the
  // tap or the source changes (every Ctot/64 or C0/64 steps) - a wave-uniform, rarely taken branch.
  int it_tap = (kt_begin * BK) / a.Ctot;
  int it_cc = kt_begin * BK - it_tap * a.Ctot;
  bool retarget = true;
  const half_t* const zeros = a.zeros;
  const half_t* xp[XR];             // where this thread's 16 bytes of row i come from for the next tile
  int xstep[XR];                    // halves to advance per K step: BK, or 0 while the row reads zeros
  const half_t* wp[WR];
  int wstep[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    int n = n_blk + lrow + 32 * i;
    wp[i] = (n < a.N) ? a.w + (size_t)n * a.K + (size_t)kt_begin * BK + chunk * 8 : zeros;
    wstep[i] = (n < a.N) ? BK : 0;
  }

  auto load_tile = [&](int stage) {
    if constexpr ((DBG & 3) == 2) return;
    if (retarget) {                   // wave-uniform
      const int ky = (a.ksize == 3) ? it_tap / 3 : 0;
      const int kx = (a.ksize == 3) ? it_tap - ky * 3 : 0;
      int cc = it_cc;
      const half_t* src = a.x0;
      int Csrc = a.C0;
      if (cc >= a.C0) {
        src = a.x1;
        cc -= a.C0;
        Csrc = a.C1;
      }
      src += cc + chunk * 8;
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        int iy = x_iy0[i] + ky, ix = x_ix0[i] + kx;
        bool ok = (x_pix[i] >= 0) && (iy >= 0) && (iy < Hup) && (ix >= 0) && (ix < Wup);
        int pix = x_pix[i] + (iy >> upshift) * a.Wi + (ix >> upshift);
        xp[i] = ok ? src + (size_t)pix * Csrc : zeros;
        xstep[i] = ok ? BK : 0;
      }
    }
    it_cc += BK;                      // advance the iterator for the next call
    if (it_cc >= a.Ctot) {
      it_cc = 0;
      ++it_tap;
    }
    retarget = (it_cc == 0) || (it_cc == a.C0);
    if constexpr (GLDS) {
      char* xs = reinterpret_cast<char*>(Xs + stage * BM * ROW) + wave * 1024;   // wave-uniform piece base
      char* ws = reinterpret_cast<char*>(Ws + stage * BN * ROW) + wave * 1024;
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)xp[i],
                                         (__attribute__((address_space(3))) void*)(xs + i * 4096), 16, 0, 0);
        xp[i] += xstep[i];
      }
#pragma unroll
      for (int i = 0; i < WR; ++i) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)wp[i],
                                         (__attribute__((address_space(3))) void*)(ws + i * 4096), 16, 0, 0);
        wp[i] += wstep[i];
      }
    } else {
#pragma unroll
      for (int i = 0; i < XR; ++i) {
        xr[i] = *reinterpret_cast<const half8*>(xp[i]);
        xp[i] += xstep[i];
      }
#pragma unroll
      for (int i = 0; i < WR; ++i) {
        wr[i] = *reinterpret_cast<const half8*>(wp[i]);
        wp[i] += wstep[i];
      }
    }
  };
  auto store_tile = [&](int buf) {   // register-staged variant only
    if constexpr (!GLDS) {
      half_t* xs = Xs + buf * BM * ROW;
      half_t* ws = Ws + buf * BN * ROW;
#pragma unroll
      for (int i = 0; i < XR; ++i)
        *reinterpret_cast<half8*>(xs + (lrow + 32 * i) * ROW + chunk * 8) = xr[i];
#pragma unroll
      for (int i = 0; i < WR; ++i)
        *reinterpret_cast<half8*>(ws + (lrow + 32 * i) * ROW + chunk * 8) = wr[i];
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31;           // fragment row (m or n within the 32-tile)
  const int fk = (lane >> 5) * 8;       // k offset of this half-wave inside a 16-deep MFMA step
  const int fsw = (frow >> 1) & 7;      // GLDS: swizzle of this lane's fragment rows (tile bases are multiples of 32)

  // K-step body, split so the next tile's address math + DMA issue can sit in the shadow of the MFMAs:
  //   read_frags(buf)  - all 16 ds_read_b128 of the step up front (with one wave per SIMD a
  //                      read->wait->4 MFMA chain exposes the LDS latency four times per step)
  //   [load_tile(next)] - scheduled by the compiler between / behind the MFMAs
  //   mfma_step()      - 16 back-to-back MFMAs
  half8 xf[BK / 16][TM] = {}, wf[BK / 16][TN] = {};
  // LNF: per-lane partial sum / sum of squares of activation row (lane & 31) of each m-tile, over the
  // k chunks this half-wave reads (v_dot2_f32_f16: exact fp16 products, fp32 accumulation)
  float ln_s1[TM] = {}, ln_s2[TM] = {};
  auto read_frags = [&](int buf) {
    if constexpr ((DBG & 3) == 1) return;
    if constexpr ((DBG & 8) != 0) {   // ablation: no LDS reads (fragments = loop-invariant garbage kept live)
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(xf[kk][i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(wf[kk][j]));
      }
      return;
    }
    const half_t* xs = Xs + buf * BM * ROW + (wm * TM * 32 + frow) * ROW;
    const half_t* ws = Ws + buf * BN * ROW + (wn * TN * 32 + frow) * ROW;
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
      const int koff = GLDS ? (((kk * 2 + (lane >> 5)) ^ fsw) * 8) : (kk * 16 + fk);
#pragma unroll
      for (int i = 0; i < TM; ++i) xf[kk][i] = *reinterpret_cast<const half8*>(xs + i * 32 * ROW + koff);
#pragma unroll
      for (int j = 0; j < TN; ++j) wf[kk][j] = *reinterpret_cast<const half8*>(ws + j * 32 * ROW + koff);
    }
    __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead (the scheduler would sink them back to their uses)
  };
  auto mfma_step = [&]() {
    if constexpr ((DBG & 3) == 1) return;
    if constexpr ((DBG & 16) != 0) {  // ablation: no MFMAs (fragments consumed so the reads stay)
#pragma unroll
      for (int kk = 0; kk < BK / 16; ++kk) {
#pragma unroll
        for (int i = 0; i < TM; ++i) asm volatile("" ::"v"(xf[kk][i]));
#pragma unroll
        for (int j = 0; j < TN; ++j) asm volatile("" ::"v"(wf[kk][j]));
      }
      return;
    }
#pragma unroll
    for (int kk = 0; kk < BK / 16; ++kk) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          if constexpr (TRANS_OUT)   // rows = m, cols = n
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xf[kk][i], wf[kk][j], acc[i][j], 0, 0, 0);
          else                       // rows = n, cols = m
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[kk][j], xf[kk][i], acc[i][j], 0, 0, 0);
        }
      if constexpr (LNF) {           // VALU work that fits the issue slots between the MFMAs
        const half2v one2 = {(half_t)1.f, (half_t)1.f};
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const half2v p2 = {xf[kk][i][2 * e], xf[kk][i][2 * e + 1]};
            ln_s2[i] = __builtin_amdgcn_fdot2(p2, p2, ln_s2[i], false);
            ln_s1[i] = __builtin_amdgcn_fdot2(p2, one2, ln_s1[i], false);
          }
      }
    }
  };

  // the tile lies inside one sample -> its timestep-embedding row is a per-column constant too
  const bool temb_uniform = a.temb != nullptr && (a.HoWo % BM) == 0;
  // issued before the first tile's DMA (oldest VMEM ops of the wave, so the counted vmcnt waits of the ring
  // are unaffected); first used after the K loop, where they are written to LDS ahead of the epilogue barrier
  float const_b = 0.f, const_t = 0.f, const_c = 0.f;   // combined only at the store: no early use, no early wait
  const bool use_consts = !TRANS_OUT && !a.slab;
  if (use_consts && tid < BN) {
    const int n = n_blk + tid;
    if (n < a.N) {
      if (a.bias) const_b = a.bias[n];
      if (temb_uniform) const_t = a.temb[(size_t)(m_blk / a.HoWo) * a.temb_stride + n];
      if constexpr (LNF) const_c = a.ln_colsum[n];
    }
  }

  // residual tile of the final store loop, requested now (oldest VMEM ops: the ring's counted waits are unaffected)
  constexpr int RIT = BM * BN / 8 / 256;
  constexpr bool RES_PRE = !TRANS_OUT && GLDS && RIT <= 4 && DBG == 0;
  half8 resv[RES_PRE ? RIT : 1];
  const bool use_resv = RES_PRE && a.res_pre && a.res != nullptr && !a.slab && a.out_mode == kOutHalf && n_blk < a.n_trans && kg == 0;
  if constexpr (RES_PRE) {
    if (use_resv) {
#pragma unroll
      for (int it = 0; it < RIT; ++it) {
        const int idx = tid + it * 256;
        const int r = idx / (BN / 8), c = idx - r * (BN / 8);
        const int m = m_blk + r, n = n_blk + c * 8;
        const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        resv[it] = (m < a.M && n + 8 <= a.ldo) ? *reinterpret_cast<const half8*>(a.res + (size_t)m * a.ldo + n) : z;
      }
    }
  }
  if (prof) prof_t[1] = clock64();
  if constexpr (GLDS && NST >= 3) {
    // NST-stage ring: the DMA of tile kt+NST-1 is issued while tile kt is computed and tiles
    // kt+1 .. kt+NST-2 are still in flight (weight-streaming layers need the bytes in flight: at the
    // 8x8 / 16x16 levels every K step of a two-stage loop is one exposed HBM round trip).
    // Raw s_barrier + COUNTED vmcnt (a __syncthreads() would drain vmcnt(0) and serialise the ring,
    // cdna guide section 5 "Pipelining across barriers"); wait + barrier sit in one asm statement
    // with a memory clobber so no LDS access is scheduled across them.
    constexpr int PER_TILE = XR + WR;    // LDS-DMA instructions per wave per tile
    constexpr int DEPTH = NST - 2;       // tiles that may still be in flight while tile kt is computed
    static_assert(DEPTH * PER_TILE <= 63, "vmcnt range");
#pragma unroll
    for (int p = 0; p < NST - 1; ++p)
      if (kt_begin + p < kt_end) load_tile(p);
    for (int rel = 0; rel < kg_steps; ++rel) {
      const int kt = kt_begin + rel;
      if (KG == 2 && kt >= kt_end) {       // the shorter K half: keep the workgroup-wide barrier count (wave-uniform branch)
        asm volatile("s_barrier" ::: "memory");
        continue;
      }
      const int ahead = kt_end - 1 - kt;   // tiles after this one that are already issued (capped below)
      // wait until at most min(ahead, DEPTH) newer tiles are outstanding == tile kt has landed (loads of one
      // wave return in order); the immediate must be a literal, hence the ladder
      const int fly = ahead < DEPTH ? ahead : DEPTH;
      if (DEPTH >= 6 && fly == 6) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((DEPTH >= 6 ? 6 : 0) * PER_TILE) : "memory");
      else if (DEPTH >= 5 && fly == 5) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((DEPTH >= 5 ? 5 : 0) * PER_TILE) : "memory");
      else if (DEPTH >= 4 && fly == 4) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((DEPTH >= 4 ? 4 : 0) * PER_TILE) : "memory");
      else if (DEPTH >= 3 && fly == 3) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((DEPTH >= 3 ? 3 : 0) * PER_TILE) : "memory");
      else if (DEPTH >= 2 && fly == 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"((DEPTH >= 2 ? 2 : 0) * PER_TILE) : "memory");
      else if (fly == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(PER_TILE) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      read_frags(rel % NST);
      if (kt + NST - 1 < kt_end) load_tile((rel + NST - 1) % NST);
      mfma_step();
    }
  } else if constexpr (GLDS) {
    if (kt_begin < kt_end) load_tile(0);
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const int buf = (kt - kt_begin) & 1;
      // vmcnt(0) + barrier: this tile's DMA has landed for every wave, and every wave is done
      // reading the other stage, which the next DMA may now overwrite
      __syncthreads();
      read_frags(buf);
      if (kt + 1 < kt_end) load_tile(buf ^ 1);
      mfma_step();
    }
  } else {
    if (kt_begin < kt_end) {
      load_tile(0);
      store_tile(0);
    }
    __syncthreads();
    for (int kt = kt_begin; kt < kt_end; ++kt) {
      const int buf = (kt - kt_begin) & 1;
      const bool more = (kt + 1) < kt_end;
      read_frags(buf);
      if (more) load_tile(0);     // HBM/L2 latency hides under this tile's MFMAs
      mfma_step();
      if (more) store_tile(buf ^ 1);
      __syncthreads();
    }
  }

  if (prof) prof_t[2] = clock64();
  if constexpr (KG == 2) {
    // K group 1 hands its accumulators to group 0 through LDS (its own ring: every read of it is behind the barrier) and retires
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    floatx4* hx = reinterpret_cast<floatx4*>(smem + (size_t)NST * (BM + BN) * ROW * sizeof(half_t));   // group 1's ring
    static_assert((size_t)TM * TN * 4 * 256 * sizeof(floatx4) <= (size_t)NST * (BM + BN) * ROW * sizeof(half_t), "hand-over fits one ring");
    if (kg == 1) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            hx[((i * TN + j) * 4 + q) * 256 + tid] = floatx4{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
    }
    __syncthreads();
    if (kg == 1) return;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const floatx4 v = hx[((i * TN + j) * 4 + q) * 256 + tid];
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i][j][4 * q + e] += v[e];
        }
  }
  // ---------------------------------- epilogue ----------------------------------
  const int hi = lane >> 5;
  // LNF: y = rstd*(x.W') - rstd*mean*colsum + bias'  ->  out = acc*ln_a + ln_b*colsum[n] + bias[n]
  float ln_a[TM] = {}, ln_b[TM] = {};
  if constexpr (LNF) {
    const float inv_k = 1.0f / (float)a.K;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const float s1 = xor32_sum(ln_s1[i]), s2 = xor32_sum(ln_s2[i]);   // the two k halves of the wave
      const float mean = s1 * inv_k;
      const float var = fmaxf(s2 * inv_k - mean * mean, 0.f);
      ln_a[i] = rsqrtf(var + a.ln_eps);
      ln_b[i] = -ln_a[i] * mean;
    }
  }
  if constexpr (TRANS_OUT) {
    // acc[i][j][r]: m = m0 + (r&3) + 8*(r>>2) + 4*hi ; n = n0 + (lane&31).  out[b][n][s]
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int n = n_blk + (wn * TN + j) * 32 + frow;
        const int m0 = m_blk + (wm * TM + i) * 32 + 4 * hi;
        if (n >= a.N) continue;
        const float bv = a.bias ? a.bias[n] : 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = m0 + 8 * q;
          if ((a.HoWo & 3) == 0 && (a.ldT & 3) == 0 && m + 3 < a.M) {   // 4 tokens of one image, 8-B aligned
            const int b = m / a.HoWo;
            const int s = m - b * a.HoWo;
            half4 o = {(half_t)(acc[i][j][4 * q] + bv), (half_t)(acc[i][j][4 * q + 1] + bv),
                       (half_t)(acc[i][j][4 * q + 2] + bv), (half_t)(acc[i][j][4 * q + 3] + bv)};
            *reinterpret_cast<half4*>(a.out + ((size_t)b * a.N + n) * a.ldT + s) = o;
            continue;
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            int mm = m + e;
            if (mm < a.M) {
              int b = mm / a.HoWo;
              int s = mm - b * a.HoWo;
              a.out[((size_t)b * a.N + n) * a.ldT + s] = (half_t)(acc[i][j][4 * q + e] + bv);
            }
          }
        }
      }
    return;
  } else {
    tile_epilogue<BM, BN, WGM, WGN, TM, TN, LNF, (RES_PRE ? RIT : 0)>(a, acc, ln_a, ln_b, smem, sconst, const_b, const_t, const_c, m_blk,
                                                                      n_blk, wave, split, temb_uniform, resv, use_resv);
  }
  if (prof) {
    prof_t[3] = clock64();
    a.prof[0] = prof_t[0];
    a.prof[1] = prof_t[1];
    a.prof[2] = prof_t[2];
    a.prof[3] = prof_t[3];
    a.prof[4] = wall_clock64() - prof_w0;
  }
}

// ---------------------------------------------------------------------------------------------
// 3x3 / stride 1 convolution with an LDS-resident input HALO tile (resnet convs, unet.py:435-456).
// The im2col kernel above re-fetches the activation tile once per tap (9x); its K step moves
// 32 KB into LDS for 2 MFLOP, and the measured LDS fill rate (~40 GB/s per CU, tools/prof_conv.py) -
// not the MFMA pipe - bounds it.  Here a workgroup owns an 8x16-pixel output tile: per 64-channel
// chunk it stages the 10x18 halo ONCE (23 KB, reused by all nine taps) plus one 16-KB weight tile
// per tap -> 167 KB instead of 288 KB per nine K steps.  The tap is a shift of the fragment's LDS
// row.  The next chunk's halo streams in as one 1-KB DMA piece per wave per tap.
// (Round 2's first halo kernel - one wave per 32x64 sub-tile, fragments read and waited for before the MFMAs, plan tiles
// 5 / 6 - lost to the K-split kernel below on every shape of every model and was removed in round 4; its ablation is
// profiles/r02_ablate_halo.txt.)
// ---------------------------------------------------------------------------------------------
constexpr int HALO_W = 18, HALO_ROWS = 180, HALO_PIECES = 23, HALO_LDS_ROWS = 184, HALO_PPW = 6;

template <int N>
__device__ __forceinline__ void wait_vmcnt_barrier() {   // counted wait + raw barrier in one statement (no LDS access moves across)
  asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(N) : "memory");
}
constexpr int halo_mod9(int t) { return ((t % 9) + 9) % 9; }
// halo pieces of the NEXT chunk a wave issues in tap step `tap`.  Plain kernel: one per step in steps 0-5.  GNL (GroupNorm in the
// loader): two in step 0, one in steps 1-4 - the last piece must have LANDED one step before the halo is first read (step 8), so
// that its owner can still transform it in place behind the barrier of step 7.
template <bool GNL = false>
constexpr int halo_x_issued(int tap) {
  const int t = halo_mod9(tap);
  return GNL ? (t == 0 ? 2 : (t <= 4 ? 1 : 0)) : (t < HALO_PPW ? 1 : 0);
}
// D = stages of the weight ring (D - 1 tap steps of weights in flight; 2 = the round-1 double buffer).
constexpr size_t halo_lds_bytes(int bn, int d) {
  return ((size_t)2 * HALO_LDS_ROWS * BK + (size_t)d * bn * BK) * sizeof(half_t) + bn * sizeof(float);
}

// ---------------------------------------------------------------------------------------------
// Software-pipelined, K-split variant of the halo kernel (plan tile 7).  Measured on the kernel above
// (round 2's ablation builds of that kernel, profiles/r02_ablate_halo.txt): ONE workgroup alone on a CU needs 23 us for its 45 tap
// steps and every phase of a step is exposed - fragment reads 9-11 us, MFMAs 5 us, DMA issue 3 us, barriers and
// launch 6.5 us - because a wave reads its fragments, waits for them and only then issues its MFMAs; overlap
// exists only between two co-resident workgroups.  Here
//   * the four waves are 2 (pixel halves) x 2 (K halves: 32 of the chunk's 64 channels each), so a wave owns a
//     64-pixel x 64-channel accumulator tile: 8 ds_read_b128 per 8 MFMAs instead of 12 (every weight byte in LDS
//     is read by two waves instead of four), the K halves are summed through LDS once, in the epilogue;
//   * the fragments are double-buffered in registers: the reads of tap step s+1 and the DMA of step s+D are
//     issued between the MFMAs of step s (one barrier per step, placed where every wave has drained its reads of
//     step s, so the DMA may refill that ring stage at once: D-1 steps of lead);
//   * the LDS swizzle is keyed on the halo COLUMN ((hx >> 1) & 7) instead of the linear halo row: the 16 lanes of a
//     ds_read_b128 group then cover 16 consecutive columns under every tap shift - conflict-free (the row-keyed
//     swizzle of the kernel above measures 40 % bank-conflict cycles, profiles/r02_sq_counters.json);
//   * the DMA is buffer_load ... lds with a loop-invariant per-lane offset and a scalar running offset, the final chunk
//     of the K range is a separate instantiation (no DMA past the end), and nothing in a tap step is conditional: one
//     basic block per nine steps, so the interleaving above is what the scheduler emits.
// ---------------------------------------------------------------------------------------------
template <int D, int WR, bool GNL = false>
constexpr int halo_ks_wait_count(int tap) {
  // VMEM ops a wave has issued after the weight tile of step s+1 when it waits in step s (tap `tap`): per step, in
  // order, [WR weight pieces of step s'+D] [one halo piece of the next chunk when tap(s') < HALO_PPW]
  // (the halo piece issued in the same step as the awaited tile is waited for too: the count then does not depend on
  // the order of the DMA instructions inside one step)
  int n = 0;
  for (int i = 2; i <= D - 1; ++i) n += WR + halo_x_issued<GNL>(tap - D + i);
  if (tap == 8 && n > 2 * WR) n = 2 * WR;   // the next chunk's halo (last piece went out at tap 5) is read right after
  return n;
}
template <int D, int WR, bool GNL = false>
__device__ __forceinline__ void halo_ks_wait(int tap) {
  switch (tap) {
    case 0: wait_vmcnt_barrier<halo_ks_wait_count<D, WR, GNL>(0)>(); break;
    case 1: wait_vmcnt_barrier<halo_ks_wait_count<D, WR, GNL>(1)>(); break;
    case 2: wait_vmcnt_barrier<halo_ks_wait_count<D, WR, GNL>(2)>(); break;
    case 3: wait_vmcnt_barrier<halo_ks_wait_count<D, WR, GNL>(3)>(); break;
    case 4: wait_vmcnt_barrier<halo_ks_wait_count<D, WR, GNL>(4)>(); break;
    case 5: wait_vmcnt_barrier<halo_ks_wait_count<D, WR, GNL>(5)>(); break;
    case 6: wait_vmcnt_barrier<halo_ks_wait_count<D, WR, GNL>(6)>(); break;
    case 7: wait_vmcnt_barrier<halo_ks_wait_count<D, WR, GNL>(7)>(); break;
    default: wait_vmcnt_barrier<halo_ks_wait_count<D, WR, GNL>(8)>(); break;
  }
}

// ... in the final chunk of a workgroup's K range: weight tiles are issued only while they exist (tap + D < 9), no halo
// pieces; steps before the chunk (u < 0) count as one full weight tile each (conservative: their halo pieces are waited for)
template <int D, int WR>
constexpr int halo_ks_wait_count_last(int tap) {
  int n = 0;
  for (int u = tap - D + 2; u <= tap - 1; ++u) n += (u < 0 || u + D < 9) ? WR : 0;
  return n;
}
template <int D, int WR>
__device__ __forceinline__ void halo_ks_wait_last(int tap) {
  switch (tap) {
    case 0: wait_vmcnt_barrier<halo_ks_wait_count_last<D, WR>(0)>(); break;
    case 1: wait_vmcnt_barrier<halo_ks_wait_count_last<D, WR>(1)>(); break;
    case 2: wait_vmcnt_barrier<halo_ks_wait_count_last<D, WR>(2)>(); break;
    case 3: wait_vmcnt_barrier<halo_ks_wait_count_last<D, WR>(3)>(); break;
    case 4: wait_vmcnt_barrier<halo_ks_wait_count_last<D, WR>(4)>(); break;
    case 5: wait_vmcnt_barrier<halo_ks_wait_count_last<D, WR>(5)>(); break;
    case 6: wait_vmcnt_barrier<halo_ks_wait_count_last<D, WR>(6)>(); break;
    default: wait_vmcnt_barrier<halo_ks_wait_count_last<D, WR>(7)>(); break;
  }
}

// GNL (round 5): GroupNorm(+SiLU) of the INPUT applied in the halo loader (ResnetBlock2D norm1 -> SiLU -> conv1
// and norm2 -> SiLU -> conv2, unet.py:472-481): x0 is the UN-normalised tensor, its producer left (sum, sumsq) partials
// (gnf_partial / gnf_entries); every workgroup folds the ones of its sample into a per-channel (mean, scale, shift) table in LDS
// (the GNF prologue of gemm_pipe_body) and every wave transforms the halo pieces IT fetched, in place, once they have landed -
// each halo element once per workgroup, not once per tap: read back its own 16 bytes, (x - mean) * scale + shift in packed
// fp16, SiLU in fp32, write back.  Pixels outside the image keep the zeros of the range-checked DMA (the conv pads the
// NORMALISED tensor).  The pieces of the next chunk go out two in step 0, one in steps 1-4 and are transformed in steps 3-7
// (landed: the counted wait of step t covers everything issued in steps <= t - (D - 1)); step 8 reads the halo as before.
// The GroupNorm launch and the round trip of the normalised tensor through HBM are gone.  D <= 4.
template <int D, int DBG = 0, bool GNL = false>
__global__ __launch_bounds__(256, halo_lds_bytes(64, D) <= 80 * 1024 ? 2 : 1) void conv3x3_halo_ks_kernel(IgemmArgs a) {
  constexpr int BN = 64, BM = 128, WR = 2, ROWB = BK * 2;   // ROWB: bytes per LDS row
  static_assert(halo_ks_wait_count<D, WR, GNL>(1) <= 63 && (D - 1) * WR <= 63, "vmcnt range");
  static_assert(halo_lds_bytes(64, D) >= 32 * 1024 + BM * (BN + 8) * 2 + BN * 4, "epilogue buffers fit the K-loop buffers");
  static_assert(!GNL || (D <= 4 && DBG == 0), "GroupNorm in the loader: rings of <= 4 stages");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Xh = smem;                                          // [2][HALO_LDS_ROWS][BK] halves
  char* const Ws = smem + 2 * HALO_LDS_ROWS * ROWB;               // [D][BN][BK]
  // [BN] bias + timestep-embedding row, written in the epilogue only.  GNL: behind the epilogue's staging buffers (the K-loop
  // buffers are free by then) - its place behind the ring holds the GroupNorm table [Ctot] mean | scale | shift (halves), so the
  // 320-channel convs of the 64x64 level still fit two workgroups per CU (81 792 B with the 4-stage ring)
  float* sconst = GNL ? reinterpret_cast<float*>(smem + 32 * 1024 + BM * (BN + 8) * 2) : reinterpret_cast<float*>(Ws + D * BN * ROWB);
  half_t* const gn_tab = reinterpret_cast<half_t*>(Ws + D * BN * ROWB);
  float* const gn_stat = reinterpret_cast<float*>(Xh + HALO_LDS_ROWS * ROWB);   // [64] mean | [64] rstd: prologue only (halo buffer 1 is idle until step 0)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wk = wave & 1;
  const int H = a.Ho, W = a.Wo;                 // output (= upsampled input) image; nearest-x2 is folded into the halo gather
  const int ush = a.up >> 1;

  const int n_tiles = (a.N + BN - 1) / BN;
  const int m_tiles = a.B * a.tiles_y * a.tiles_x;
  const int nwg = m_tiles * n_tiles;
  int bid = blockIdx.x;
  {
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    bid = base + idx;
  }
  const int bn_idx = a.n_fast ? bid % n_tiles : bid / m_tiles, mt = a.n_fast ? bid / n_tiles : bid % m_tiles;
  const int b = mt / (a.tiles_y * a.tiles_x);
  const int trem = mt - b * (a.tiles_y * a.tiles_x);
  const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
  const int y0 = ty * 8, x0 = tx * 16;
  const int n_blk = bn_idx * BN;

  const int nch = a.Ctot / BK;
  const int split = blockIdx.y;
  const int ch_begin = split * a.nk_per_split;
  int ch_end = ch_begin + a.nk_per_split;
  if (ch_end > nch) ch_end = nch;

  // DMA through buffer_load_dwordx4 ... lds: the per-lane part of an address is a loop-invariant 32-bit VGPR offset, the
  // part that advances per tap / chunk is the scalar soffset, so a DMA instruction costs NO vector ALU work inside the
  // tap loop (the 64-bit global_load_lds form needs ~6 VALU per piece; measured in round 2, profiles/r02_ablate_halo.txt: DMA issue was
  // the largest exposed cost of the loop, 4.4 of 16.8 us).  Rows outside the image read as zeros through the buffer
  // range check (offset >= num_records -> 0); weight rows past N are clamped to row N-1 (their outputs are not stored).
  constexpr unsigned kOob = 0x80000000u;
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.w), 0, (int)((size_t)a.N * a.K * 2), 0x00020000);
  const size_t xpix = (size_t)a.B * a.Hi * a.Wi;
  unsigned hoff0[HALO_PPW], hoff1[HALO_PPW];            // byte offset of this lane's 16 bytes of halo piece j at channel 0
  // piece of slot j.  The 24th slot (wave 3, j = 5) has no piece of its own: plain kernel - it fetches piece 22 a second time
  // (same bytes); GNL - it must not touch another wave's piece (that wave transforms it in place), so it re-fetches the wave's
  // OWN previous piece, and that piece is transformed once, behind the later of its two fetches (slot 5 instead of slot 4)
  auto piece_of = [&](int j) { return (wave + 4 * j < HALO_PIECES) ? wave + 4 * j : (GNL ? wave + 4 * (j - 1) : HALO_PIECES - 1); };
#pragma unroll
  for (int j = 0; j < HALO_PPW; ++j) {
    const int p = piece_of(j);
    const int hr = 8 * p + (lane >> 3);
    const int hy = hr / HALO_W, hx = hr - hy * HALO_W;
    const int iy = y0 - 1 + hy, ix = x0 - 1 + hx;
    const bool ok = (hr < HALO_ROWS) && iy >= 0 && iy < H && ix >= 0 && ix < W;
    const unsigned pix = (unsigned)((b * a.Hi + (iy >> ush)) * a.Wi + (ix >> ush));   // unet.py:498-500 nearest upsample
    const unsigned sw = (unsigned)(((lane & 7) ^ ((hx >> 1) & 7)) * 16);   // column-keyed swizzle
    hoff0[j] = ok ? pix * (unsigned)a.C0 * 2u + sw : kOob;
    hoff1[j] = ok ? pix * (unsigned)a.C1 * 2u + sw : kOob;
  }
  const int pchunk = tid & 7, lrow = tid >> 3;
  const int wchunk = pchunk ^ ((lrow >> 1) & 7);
  unsigned woff[WR];
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    int n = n_blk + lrow + 32 * i;
    if (n > a.N - 1) n = a.N - 1;
    woff[i] = (unsigned)n * (unsigned)a.K * 2u + (unsigned)wchunk * 16u;
  }

  // the halo of one channel chunk: which of the two concatenated sources it comes from is decided once per chunk
  // (HALO_SRC declares rs / off[] / soff for chunk `ch` as plain locals)
#define HALO_SRC(rs, off, soff, ch)                                                                                        \
  const bool rs##_second = !GNL && (ch) * BK >= a.C0; /* wave-uniform; GNL: single source (6 VGPRs of offsets less) */    \
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(                                                     \
      const_cast<half_t*>(rs##_second ? a.x1 : a.x0), 0, (int)(xpix * (rs##_second ? a.C1 : a.C0) * 2), 0x00020000);       \
  const int soff = (rs##_second ? (ch) * BK - a.C0 : (ch) * BK) * 2;                                                       \
  unsigned off[HALO_PPW];                                                                                                  \
  _Pragma("unroll") for (int j_ = 0; j_ < HALO_PPW; ++j_) off[j_] = rs##_second ? hoff1[j_] : hoff0[j_];
  auto issue_x_piece = [&](int j, const __amdgpu_buffer_rsrc_t& rs, unsigned off, int soff, int xstage) {
    if constexpr ((DBG & 8) != 0) return;
    char* dst = Xh + xstage * (HALO_LDS_ROWS * ROWB) + piece_of(j) * 1024;
    dma16_to_lds(rs, dst, off, soff);
  };
  // GNL: GroupNorm(+SiLU) of slot j of this wave, in place, in halo buffer `xstage` holding channel chunk `ch`.  Only called
  // once the piece has landed (this wave's own DMA: no other wave touches these 16 bytes before the next barrier).
  // Branch-free on purpose (selects, no exec-masked store): a tap step stays ONE basic block, so the MFMA / read / DMA interleaving
  // of the loop body is still what the scheduler emits.
  auto gn_transform = [&](int j, int xstage, int ch) {
    if constexpr (GNL) {
      char* pp = Xh + xstage * (HALO_LDS_ROWS * ROWB) + piece_of(j) * 1024 + lane * 16;
      const half8 raw = *reinterpret_cast<const half8*>(pp);
      half8 v = raw;
      // the lane's channel group is the swizzled 16-B slot it fetched: bits 4-6 of its source offset (pixel offsets are multiples
      // of 128 B); a lane outside the image (kOob) reads some row of the table and stores nothing
      const int k0 = ch * BK + (int)((hoff0[j] >> 4) & 7u) * 8;
      const half8 gm = *reinterpret_cast<const half8*>(gn_tab + k0);
      const half8 gs = *reinterpret_cast<const half8*>(gn_tab + a.Ctot + k0);
      const half8 gh = *reinterpret_cast<const half8*>(gn_tab + 2 * a.Ctot + k0);
      v = __builtin_elementwise_fma(v - gm, gs, gh);   // rounding relative to |x - mean| (gemm_pipe_body GNF)
#pragma unroll
      for (int e = 0; e < 8; ++e) {                    // SiLU in fp32 (unet.py:474, :480: every GroupNorm in front of a 3x3 conv has one)
        const float f = (float)v[e];
        v[e] = (half_t)(f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * f)));
      }
      // what goes back: the raw bytes for a pixel outside the image (the DMA's zeros: the conv pads the NORMALISED tensor) and for
      // wave 3's slot 4, whose piece slot 5 fetches a second time and transforms (a second landing after a transform here would
      // put raw data back; transforming twice would be wrong; writing raw over raw is harmless)
      const bool keep_raw = hoff0[j] == kOob || (j == HALO_PPW - 2 && wave + 4 * (HALO_PPW - 1) >= HALO_PIECES);
      half8 o;
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] = keep_raw ? raw[e] : v[e];
      *reinterpret_cast<half8*>(pp) = o;
    }
  };
  const int total_steps = (ch_end - ch_begin) * 9;
  int iw_koff = ch_begin * BK * 2, iw_tap = 0, iw_step = 0;   // issue cursor of the weight ring (koff in bytes)
  auto issue_next_w = [&]() {
    if constexpr ((DBG & 4) != 0) return;
    char* ws = Ws + ((unsigned)iw_step % D) * (BN * ROWB) + wave * 1024;
#pragma unroll
    for (int i = 0; i < WR; ++i)
      dma16_to_lds(rs_w, ws + i * 4096, woff[i], iw_koff);
    ++iw_step;
    iw_koff += a.Ctot * 2;                                // next tap, same chunk
    if (++iw_tap == 9) {
      iw_tap = 0;
      iw_koff += (BK - 9 * a.Ctot) * 2;                   // tap 0 of the next chunk
    }
  };

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, hi = lane >> 5;
  // fragment byte offsets: this wave reads the 16-B chunks kc = 2 * (2 * wk + s) + hi, s = 0, 1 of every row
  int hrb[2];                                             // halo row of this lane's pixel (tap (0,0)), in bytes
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ml = (wm * 2 + i) * 32 + frow;
    hrb[i] = ((ml >> 4) * HALO_W + (ml & 15)) * ROWB;
  }
  const int hx0 = frow & 15;
  int xsw[3][2], wsw[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int kc = 2 * (2 * wk + s) + hi;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) xsw[kx][s] = (kc ^ (((hx0 + kx) >> 1) & 7)) * 16;
    wsw[s] = frow * ROWB + (kc ^ ((frow >> 1) & 7)) * 16;
  }

  float const_b = 0.f, const_t = 0.f;
  if (!a.slab && tid < BN && n_blk + tid < a.N) {
    if (a.bias) const_b = a.bias[n_blk + tid];
    if (a.temb) const_t = a.temb[(size_t)b * a.temb_stride + n_blk + tid];
  }

  auto read_step = [&](half8 (&xf)[2][2], half8 (&wf)[2][2], const char* xs, const char* ws, int tap) {
    const int ky = tap / 3, kx = tap - ky * 3;
    const int toffb = (ky * HALO_W + kx) * ROWB;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if constexpr ((DBG & 2) != 0) asm volatile("" : "=v"(xf[s][i]));
        else xf[s][i] = *reinterpret_cast<const half8*>(xs + hrb[i] + toffb + xsw[kx][s]);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if constexpr ((DBG & 2) != 0) asm volatile("" : "=v"(wf[s][j]));
        else wf[s][j] = *reinterpret_cast<const half8*>(ws + j * 32 * ROWB + wsw[s]);
      }
    }
  };
  auto mfma_step = [&](half8 (&xf)[2][2], half8 (&wf)[2][2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          if constexpr ((DBG & 1) != 0) asm volatile("" ::"v"(wf[s][j]), "v"(xf[s][i]));
          else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s][j], xf[s][i], acc[i][j], 0, 0, 0);
        }
  };

  half8 xfA[2][2], wfA[2][2], xfB[2][2], wfB[2][2];
  if (ch_begin < ch_end) {
    // GNL prologue, part 1 (oldest VMEM ops of the wave, before the first DMA; inside this branch, so that no path reaches the
    // join behind it with these loads outstanding - the compiler would drain vmcnt(0), ring and all, there): this sample's partial
    // statistics - eight lanes per group, 16 entries (8 float4) each, entries <= 128; entries past the count are allocated but
    // stale, so they are loaded unconditionally (no branch, no early wait) and masked - and gamma / beta of the channels whose
    // table rows this thread writes
    floatx4 gnl_v[GNL ? 8 : 1];
    float gnl_g[GNL ? 8 : 1], gnl_b[GNL ? 8 : 1];
    if constexpr (GNL) {
      const int g = tid >> 3, j = tid & 7;
      const floatx4* src = reinterpret_cast<const floatx4*>(a.gnf_partial + (((size_t)b * a.gnf_G + (g < a.gnf_G ? g : 0)) * kGnMaxSlabs + j * 16) * 2);
#pragma unroll
      for (int k = 0; k < 8; ++k) gnl_v[k] = src[k];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = min(tid + 256 * i, a.Ctot - 1);
        gnl_g[i] = a.gnf_gamma[k];
        gnl_b[i] = a.gnf_beta[k];
      }
    }
    HALO_SRC(rs0, off0, soff0, ch_begin)
#pragma unroll
    for (int j = 0; j < HALO_PPW; ++j) issue_x_piece(j, rs0, off0[j], soff0, 0);
#pragma unroll
    for (int p = 0; p < D; ++p) {
      asm volatile("" ::: "memory");                      // keep the DMA issue order: the counted wait below relies on it
      if (p < total_steps) issue_next_w();
    }
    if constexpr (GNL) {
      // part 2: fold (fixed order), statistics -> LDS, per-channel table -> LDS; the barrier below publishes it
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int e0 = (tid & 7) * 16 + 2 * k;            // first of the two entries of this float4
        const bool l0 = e0 < a.gnf_entries, l1 = e0 + 1 < a.gnf_entries;
        s += (l0 ? gnl_v[k][0] : 0.f) + (l1 ? gnl_v[k][2] : 0.f);
        q += (l0 ? gnl_v[k][1] : 0.f) + (l1 ? gnl_v[k][3] : 0.f);
      }
#pragma unroll
      for (int o = 1; o < 8; o <<= 1) {
        s += __shfl_xor(s, o);
        q += __shfl_xor(q, o);
      }
      const int cpg = a.Ctot / a.gnf_G;
      if ((tid & 7) == 0 && (tid >> 3) < a.gnf_G) {
        const float inv_n = 1.0f / ((float)cpg * (float)(a.Hi * a.Wi));
        const float mean = s * inv_n;
        gn_stat[tid >> 3] = mean;
        gn_stat[64 + (tid >> 3)] = rsqrtf(fmaxf(q * inv_n - mean * mean, 0.f) + a.gnf_eps);
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int k = tid + 256 * i;
        if (k < a.Ctot) {
          const int g = k / cpg;
          const float sc = gnl_g[i] * gn_stat[64 + g];
          const half_t mh = (half_t)gn_stat[g];
          gn_tab[k] = mh;
          gn_tab[a.Ctot + k] = (half_t)sc;
          gn_tab[2 * a.Ctot + k] = (half_t)(gnl_b[i] - (gn_stat[g] - (float)mh) * sc);
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (total_steps >= D) wait_vmcnt_barrier<(D - 1) * WR>();   // the first halo and weight tile 0 have landed (GNL: and the table is visible)
    else wait_vmcnt_barrier<0>();
    if constexpr (GNL) {   // the first chunk's halo: every wave normalises the pieces it fetched, then all of them meet
#pragma unroll
      for (int j = 0; j < HALO_PPW; ++j) gn_transform(j, 0, ch_begin);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    read_step(xfA, wfA, Xh, Ws, 0);
  }
  int st = 0;
  // One chunk = nine tap steps.  PAR = parity of the chunk's first step (which register set holds its fragments); LAST = the
  // final chunk of this workgroup's K range: no next halo, no weight tiles past the end, the waits count what is left.
  auto chunk = [&](auto par, auto last, int ch) {
    constexpr int PAR = decltype(par)::value;
    constexpr bool LAST = decltype(last)::value;
    const int xst = (ch - ch_begin) & 1;
    const bool first_chunk = ch == ch_begin;
    HALO_SRC(rsn, offn, soffn, (LAST ? ch : ch + 1))
#pragma unroll
    for (int tap = 0; tap < 9; ++tap, ++st) {
      auto body = [&](half8 (&cx)[2][2], half8 (&cw)[2][2], half8 (&nx)[2][2], half8 (&nw)[2][2]) {
        if (LAST && tap == 8) {                           // the final step: nothing left to fetch
          mfma_step(cx, cw);
          return;
        }
        // this wave's fragment reads of step st are in registers: after the barrier NO wave reads ring stage st % D any more
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        // weight tile st+1 (at tap 8: and the next halo) has landed
        if constexpr (LAST) halo_ks_wait_last<D, WR>(tap);
        else if (D > 4 && first_chunk) {
          // fewer halo pieces behind the weight tiles than in steady state: count the tiles only - but at tap 8 the next
          // chunk's halo (last piece issued at tap 5, two steps = 2 * WR weight pieces ago) must have landed as well
          if (tap == 8) wait_vmcnt_barrier<((D - 2) * WR < 2 * WR ? (D - 2) * WR : 2 * WR)>();
          else wait_vmcnt_barrier<(D - 2) * WR>();
        }
        else halo_ks_wait<D, WR, GNL>(tap);
        const int tapn = tap == 8 ? 0 : tap + 1;
        const char* xs = Xh + (tap == 8 ? xst ^ 1 : xst) * (HALO_LDS_ROWS * ROWB);
        const char* ws = Ws + ((unsigned)(st + 1) % D) * (BN * ROWB);
        read_step(nx, nw, xs, ws, tapn);                  // fragments of step st+1 -> the other register set
        const bool w_live = !LAST || tap + D < 9;
        if (w_live) issue_next_w();                       // weight tile st+D -> stage st % D
        if constexpr (GNL) {
          if (!LAST) {
            if (tap == 0) {
              issue_x_piece(0, rsn, offn[0], soffn, xst ^ 1);
              issue_x_piece(1, rsn, offn[1], soffn, xst ^ 1);
            } else if (tap <= 4) {
              issue_x_piece(tap + 1, rsn, offn[tap + 1 < HALO_PPW ? tap + 1 : 0], soffn, xst ^ 1);
            }
            // pieces issued in steps <= tap - 3 have landed behind the wait above (D <= 4): normalise them in place
            if (tap == 3) {
              gn_transform(0, xst ^ 1, ch + 1);
              gn_transform(1, xst ^ 1, ch + 1);
            } else if (tap >= 4 && tap <= 7) {
              gn_transform(tap - 2, xst ^ 1, ch + 1);
            }
          }
        } else {
          const bool x_live = !LAST && tap < HALO_PPW;
          if (x_live) issue_x_piece(tap, rsn, offn[tap < HALO_PPW ? tap : 0], soffn, xst ^ 1);
        }
        mfma_step(cx, cw);
        // issue order: an MFMA, then two of the next step's reads ... the DMA pieces behind the later MFMAs
        if constexpr (DBG == 0) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
          }
#pragma unroll
          for (int g = 0; g < 3; ++g) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
      };
      if (((tap + PAR) & 1) == 0) body(xfA, wfA, xfB, wfB);
      else body(xfB, wfB, xfA, wfA);
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  {
    using T = std::true_type;
    using F = std::false_type;
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int ch = ch_begin;
    for (; ch + 2 < ch_end; ch += 2) {
      chunk(P0{}, F{}, ch);
      chunk(P1{}, F{}, ch + 1);
    }
    if (ch + 2 == ch_end) {
      chunk(P0{}, F{}, ch);
      chunk(P1{}, T{}, ch + 1);
    } else if (ch + 1 == ch_end) {
      chunk(P0{}, T{}, ch);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();                                        // every wave is out of the K loop: LDS is free

  // ---- epilogue.  acc[i][j][r]: n = j*32 + (r&3) + 8*(r>>2) + 4*hi ; pixel block 2*wm + i, pixel frow ----
  // sum the two K halves: wave (wm, wk) keeps pixel block 2*wm + wk and hands the other one to its partner
  {
    floatx4* red = reinterpret_cast<floatx4*>(smem);      // [4 waves][2 j][4 q][64 lanes] = 32 KB
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        floatx4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = wk ? acc[0][j][4 * q + e] : acc[1][j][4 * q + e];
        red[((wave * 2 + j) * 4 + q) * 64 + lane] = v;
      }
    if (tid < BN) sconst[tid] = const_b + const_t;
    __syncthreads();
  }
  floatx16 fin[2];
  {
    const floatx4* red = reinterpret_cast<const floatx4*>(smem);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const floatx4 v = red[(((wave ^ 1) * 2 + j) * 4 + q) * 64 + lane];
#pragma unroll
        for (int e = 0; e < 4; ++e) fin[j][4 * q + e] = (wk ? acc[1][j][4 * q + e] : acc[0][j][4 * q + e]) + v[e];
      }
  }
  const int ml = (wm * 2 + wk) * 32 + frow;               // tile-local pixel of this lane
  if (a.slab) {
    const int y = y0 + (ml >> 4), x = x0 + (ml & 15);
    if (y < H && x < W) {
      const int m = (b * H + y) * W + x;
      float* prow = a.partial + ((size_t)split * a.M + m) * a.N;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = n_blk + j * 32 + 8 * q + 4 * hi;
          if (n < a.N) {
            floatx4 v = {fin[j][4 * q], fin[j][4 * q + 1], fin[j][4 * q + 2], fin[j][4 * q + 3]};
            out_store(reinterpret_cast<floatx4*>(prow + n), v);
          }
        }
    }
    return;
  }
  constexpr int OROW = BN + 8;
  half_t* ot = reinterpret_cast<half_t*>(smem + 32 * 1024);   // [BM][OROW], behind the reduction buffer
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int nl = j * 32 + 8 * q + 4 * hi;
      const floatx4 bb = *reinterpret_cast<const floatx4*>(sconst + nl);   // 0 beyond N
      half4 o = {(half_t)(fin[j][4 * q] + bb[0]), (half_t)(fin[j][4 * q + 1] + bb[1]), (half_t)(fin[j][4 * q + 2] + bb[2]),
                 (half_t)(fin[j][4 * q + 3] + bb[3])};
      *reinterpret_cast<half4*>(ot + ml * OROW + nl) = o;
    }
  __syncthreads();
  constexpr int WC = BN / 8;
  const bool gn = a.gn_partial != nullptr;   // block-uniform (launch_conv: N % 8 == 0)
  float fs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, fq[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int idx = tid; idx < BM * WC; idx += 256) {
    const int r = idx / WC, c = idx - r * WC;
    const int y = y0 + (r >> 4), x = x0 + (r & 15);
    const int n = n_blk + c * 8;
    if (y < H && x < W && n < a.N) {
      const size_t m = (size_t)(b * H + y) * W + x;
      half8 v = *reinterpret_cast<const half8*>(ot + r * OROW + c * 8);
      half_t* dst = a.out + m * a.N + n;
      if (n + 8 <= a.N) {
        if (a.res) {
          const half8 rr = *reinterpret_cast<const half8*>(a.res + m * a.N + n);
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)rr[e]);
        }
        out_store(reinterpret_cast<half8*>(dst), v);
        if (gn) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float f = (float)v[e];
            fs[e] += f;
            fq[e] = fmaf(f, f, fq[e]);
          }
        }
      } else {
        for (int e = 0; e < a.N - n; ++e) dst[e] = a.res ? (half_t)((float)v[e] + (float)a.res[m * a.N + n + e]) : v[e];
      }
    }
  }
  // GroupNorm statistics of the stored pixels (only those inside the image); the scratch is the K-half reduction
  // buffer at the start of the LDS, which nobody reads after the barrier above
  if (gn) tile_gn_stats<BN>(a, reinterpret_cast<float*>(smem), fs, fq, n_blk, b, ty * a.tiles_x + tx);
}

#undef HALO_SRC

// ---------------------------------------------------------------------------------------------
// Software-pipelined 1x1 GEMM (every Linear / 1x1 conv of the transformer blocks and the resnet shortcuts at stride 1;
// unet.py:62-118, :566-617).  igemm_kernel above reads the 16 fragments of a K step, waits for them, then issues its 16
// MFMAs: with one wave per SIMD the LDS latency and the barrier are exposed once per step and the matrix pipe idles
// 50-75 % of the loop (245-520 TFLOP/s at UNet batch 16, profiles/r03_op_profile_b16_before.txt).  Here the loop body is the
// one of the K-split halo conv kernel:
//   * fragments are double-buffered in registers - between the MFMAs of step s the wave issues the ds_read_b128 of
//     step s+1 and the LDS-DMA of step s+D; the body is ONE basic block (no conditionals: tiles past the end of K go
//     through a zero-sized buffer resource), so sched_group_barrier fixes the interleaving MFMA / 2 reads ... MFMA / DMA;
//   * operands come HBM -> LDS by buffer_load_dwordx4 ... lds with loop-invariant per-lane offsets and a scalar K offset
//     (no vector address arithmetic in the loop; rows past M read zeros through the buffer range check; the second
//     source of a skip concat is a second resource selected per tile on the scalar unit);
//   * one raw s_barrier per step behind counted waits: lgkmcnt(0) (this wave holds step s in registers, so after the
//     barrier nobody reads ring stage s % D any more) and vmcnt((D-2) tiles) (tile s+1 has landed).
// Before the epilogue re-uses the LDS the wave drains vmcnt(0): an LDS-DMA still in flight when a workgroup hands its
// LDS back would land in whatever workgroup is dispatched there next.
// Accumulator layout, swizzle and epilogue are igemm_kernel's (tile_epilogue: bias / timestep embedding / LayerNorm fold /
// GEGLU / residual / fused q|k|v / GroupNorm statistics / split-K slabs).
// ---------------------------------------------------------------------------------------------
// DBG (ablation builds, tools/r5_gemm_ablation.py; results are garbage): bit 0 = no weight DMA, bit 1 = no activation DMA
// GNF (round 5): GroupNorm of the input folded in (SpatialTransformer.norm -> proj_in, unet.py:528-531 + :553-556: eps 1e-6, no
// SiLU).  The producer of x left (sum, sumsq) partials per (sample, group); every workgroup folds the ones of its sample (its M
// tile lies inside one sample) into a per-channel scale / shift table in LDS - requested as the oldest VMEM ops of the wave, in
// flight beside the first tiles' DMA - and each wave applies it to its activation fragments with v_pk_fma_f16 between the LDS
// read and the MFMA: what the GroupNorm kernel would have stored as fp16 is formed in registers instead, the launch and the
// activation round trip are gone.
// (the body is a device function so that gn_*_side_kernel can run it beside a GroupNorm in one launch: bid_in / split are what the
// stand-alone kernel takes from blockIdx.x / .y)
template <int BM, int BN, int WGM, int WGN, int D, bool LNF, int DBG = 0, bool GNF = false>
__device__ __forceinline__ void gemm_pipe_body(const IgemmArgs& a, int bid_in, int split) {
  static_assert(!(GNF && LNF), "one fold at a time");   // D = 2: 64 KB of LDS on the 128 x 128 tile, two workgroups per CU
  static_assert(WGM * WGN == 4, "4 waves");
  constexpr int TM = BM / WGM / 32, TN = BN / WGN / 32;
  constexpr int XR = BM / 32, WR = BN / 32, PER = ((DBG & 2) ? 0 : XR) + ((DBG & 1) ? 0 : WR), ROWB = BK * 2, KK = BK / 16;
  static_assert(DBG != 3, "one operand must still arrive");
  static_assert((D - 1) * PER <= 63, "vmcnt range");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Xs = smem;                                   // [D][BM][BK] halves
  char* const Ws = smem + D * BM * ROWB;                   // [D][BN][BK]
  float* sconst = reinterpret_cast<float*>(smem + (size_t)D * (BM + BN) * ROWB);
  // GNF: [64] mean | [64] rstd floats, then [K] mean | [K] scale | [K] shift halves, behind the epilogue constants
  float* gn_stat = sconst + 2 * BN;
  half_t* gn_tab = reinterpret_cast<half_t*>(gn_stat + 128);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  const int nbm = (a.M + BM - 1) / BM, nbn = (a.N + BN - 1) / BN;
  const int nwg = nbm * nbn;
  int bid = bid_in;
  {   // XCD-aware order (see igemm_kernel): each XCD walks a contiguous run of tiles sharing a weight panel
    int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    bid = base + idx;
  }
  const int bn_idx = a.n_fast ? bid % nbn : bid / nbm, bm_idx = a.n_fast ? bid / nbn : bid % nbm;
  const int m_blk = bm_idx * BM, n_blk = bn_idx * BN;
  const int kt_begin = split * a.nk_per_split;
  int kt_end = kt_begin + a.nk_per_split;
  if (kt_end > a.nk_total) kt_end = a.nk_total;
  const int T = kt_end - kt_begin;                          // K steps of this workgroup (>= 1: launch_conv leaves no empty split)

  constexpr unsigned kOob = 0x80000000u;
  const int pchunk = tid & 7, lrow = tid >> 3;
  const unsigned lchunk = (unsigned)(pchunk ^ ((lrow >> 1) & 7)) * 16u;   // logical 16-B chunk this lane fetches (bank swizzle)
  unsigned xoff0[XR], xoff1[XR], woff[WR];
#pragma unroll
  for (int i = 0; i < XR; ++i) {
    const int m = m_blk + lrow + 32 * i;
    xoff0[i] = m < a.M ? (unsigned)m * (unsigned)a.C0 * 2u + lchunk : kOob;
    xoff1[i] = m < a.M ? (unsigned)m * (unsigned)a.C1 * 2u + lchunk : kOob;
  }
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    int n = n_blk + lrow + 32 * i;
    if (n > a.N - 1) n = a.N - 1;                           // rows past N: clamped (their outputs are never stored)
    woff[i] = (unsigned)n * (unsigned)a.K * 2u + lchunk;
  }
  const unsigned x0_bytes = (unsigned)((size_t)a.M * a.C0 * 2), x1_bytes = (unsigned)((size_t)a.M * a.C1 * 2);
  const unsigned w_bytes = (unsigned)((size_t)a.N * a.K * 2);

  int iw_kt = kt_begin, iw_stage = 0;                       // issue cursor of the ring
  auto issue_tile = [&]() {
    const bool live = iw_kt < kt_end;                       // wave-uniform; past the end: zero-sized resources, nothing is fetched
    const int k = iw_kt * BK;
    const bool second = k >= a.C0;                          // the skip-concat's second source (never with C1 == 0)
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<half_t*>(second ? a.x1 : a.x0), 0, (int)(live ? (second ? x1_bytes : x0_bytes) : 0u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<half_t*>(a.w), 0, (int)(live ? w_bytes : 0u), 0x00020000);
    const int xs_off = (second ? k - a.C0 : k) * 2;
    char* xs = Xs + iw_stage * (BM * ROWB) + wave * 1024;
    char* ws = Ws + iw_stage * (BN * ROWB) + wave * 1024;
    if constexpr ((DBG & 2) == 0) {
#pragma unroll
      for (int i = 0; i < XR; ++i) dma16_to_lds(rs_x, xs + i * 4096, second ? xoff1[i] : xoff0[i], xs_off);
    }
    if constexpr ((DBG & 1) == 0) {
#pragma unroll
      for (int i = 0; i < WR; ++i) dma16_to_lds(rs_w, ws + i * 4096, woff[i], k * 2);
    }
    ++iw_kt;
    iw_stage = (iw_stage + 1 == D) ? 0 : iw_stage + 1;
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  float ln_s1[TM] = {}, ln_s2[TM] = {};

  const int frow = lane & 31, hi = lane >> 5;
  const int fsw = (frow >> 1) & 7;
  int foff[KK];                                             // byte offset of this lane's chunk of k sub-step kk inside a row
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) foff[kk] = ((kk * 2 + hi) ^ fsw) * 16;
  const int xrow = (wm * TM * 32 + frow) * ROWB, wrow = (wn * TN * 32 + frow) * ROWB;

  // per-column epilogue constants and the residual tile, requested before the first DMA (oldest VMEM ops of the wave)
  const bool temb_uniform = a.temb != nullptr && (a.HoWo % BM) == 0;
  float const_b = 0.f, const_t = 0.f, const_c = 0.f;
  if (!a.slab && tid < BN) {
    const int n = n_blk + tid;
    if (n < a.N) {
      if (a.bias) const_b = a.bias[n];
      if (temb_uniform) const_t = a.temb[(size_t)(m_blk / a.HoWo) * a.temb_stride + n];
      if constexpr (LNF) const_c = a.ln_colsum[n];
    }
  }
  constexpr int RIT = BM * BN / 8 / 256;
  constexpr bool RES_PRE = RIT <= 4;
  half8 resv[RES_PRE ? RIT : 1];
  const bool use_resv = RES_PRE && a.res_pre && a.res != nullptr && !a.slab && a.out_mode == kOutHalf && n_blk < a.n_trans;
  if constexpr (RES_PRE) {
    if (use_resv) {
#pragma unroll
      for (int it = 0; it < RIT; ++it) {
        const int idx = tid + it * 256;
        const int r = idx / (BN / 8), c = idx - r * (BN / 8);
        const int m = m_blk + r, n = n_blk + c * 8;
        const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        resv[it] = (m < a.M && n + 8 <= a.ldo) ? *reinterpret_cast<const half8*>(a.res + (size_t)m * a.ldo + n) : z;
      }
    }
  }

  struct Frags {
    half8 x[KK][TM];
    half8 w[KK][TN];
    half8 gm[GNF ? KK : 1], gs[GNF ? KK : 1], gh[GNF ? KK : 1];   // GNF: mean / scale / shift of this lane's 8 channels of every k sub-step
  };
  int rd_kt = kt_begin;                         // K step whose fragments the next read_step fetches (GNF table offset)
  auto read_step = [&](Frags& f, int stage) {
    const char* xs = Xs + stage * (BM * ROWB) + xrow;
    const char* ws = Ws + stage * (BN * ROWB) + wrow;
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
      for (int i = 0; i < TM; ++i) f.x[kk][i] = *reinterpret_cast<const half8*>(xs + i * 32 * ROWB + foff[kk]);
#pragma unroll
      for (int j = 0; j < TN; ++j) f.w[kk][j] = *reinterpret_cast<const half8*>(ws + j * 32 * ROWB + foff[kk]);
      if constexpr (GNF) {
        const int k0 = rd_kt * BK + (kk * 2 + hi) * 8;
        f.gm[kk] = *reinterpret_cast<const half8*>(gn_tab + k0);
        f.gs[kk] = *reinterpret_cast<const half8*>(gn_tab + a.K + k0);
        f.gh[kk] = *reinterpret_cast<const half8*>(gn_tab + 2 * a.K + k0);
      }
    }
    ++rd_kt;
  };
  auto mfma_step = [&](const Frags& f) {
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        half8 xv = f.x[kk][i];
        // (x - fp16(mean)) * (rstd * gamma) + (beta - (mean - fp16(mean)) * rstd * gamma): the subtraction comes first so that every
        // rounding is relative to |x - mean|, not to |mean| (a group far from zero would otherwise lose digits in the shift)
        if constexpr (GNF) xv = __builtin_elementwise_fma(xv - f.gm[kk], f.gs[kk], f.gh[kk]);
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.w[kk][j], xv, acc[i][j], 0, 0, 0);
      }
      if constexpr (LNF) {   // row statistics of the A tile for the LayerNorm fold: VALU work in the MFMAs' issue shadow
        const half2v one2 = {(half_t)1.f, (half_t)1.f};
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const half2v p2 = {f.x[kk][i][2 * e], f.x[kk][i][2 * e + 1]};
            ln_s2[i] = __builtin_amdgcn_fdot2(p2, p2, ln_s2[i], false);
            ln_s1[i] = __builtin_amdgcn_fdot2(p2, one2, ln_s1[i], false);
          }
      }
    }
  };

  // GNF prologue, part 1 (before the first DMA: oldest VMEM ops of the wave): this sample's partial statistics - eight lanes
  // per group, 16 entries (8 float4) each, entries <= 128 - and gamma / beta of the channels whose table rows this thread writes
  floatx4 gnf_v[GNF ? 8 : 1];
  float gnf_g[GNF ? 8 : 1], gnf_b[GNF ? 8 : 1];
  if constexpr (GNF) {
    const int b = m_blk / a.HoWo;
    const int g = tid >> 3, j = tid & 7;
    const floatx4* src = reinterpret_cast<const floatx4*>(a.gnf_partial + (((size_t)b * a.gnf_G + (g < a.gnf_G ? g : 0)) * kGnMaxSlabs + j * 16) * 2);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int e0 = j * 16 + 2 * k;
      floatx4 v = {0.f, 0.f, 0.f, 0.f};
      if (e0 < a.gnf_entries) v = src[k];
      if (e0 + 1 >= a.gnf_entries) {
        v[2] = 0.f;
        v[3] = 0.f;
      }
      gnf_v[k] = v;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = tid + 256 * i;
      gnf_g[i] = k < a.K ? a.gnf_gamma[k] : 0.f;
      gnf_b[i] = k < a.K ? a.gnf_beta[k] : 0.f;
    }
  }
  Frags fA, fB;
#pragma unroll
  for (int p = 0; p < D; ++p) {
    asm volatile("" ::: "memory");                          // keep the DMA issue order: the counted waits rely on it
    issue_tile();
  }
  if constexpr (GNF) {
    // part 2: fold (fixed order), statistics -> LDS, per-channel scale / shift table -> LDS; the barrier below publishes it
    float s = 0.f, q = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      s += gnf_v[k][0] + gnf_v[k][2];
      q += gnf_v[k][1] + gnf_v[k][3];
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      s += __shfl_xor(s, o);
      q += __shfl_xor(q, o);
    }
    const int cpg = a.K / a.gnf_G;
    if ((tid & 7) == 0 && (tid >> 3) < a.gnf_G) {
      const float inv_n = 1.0f / ((float)cpg * (float)a.HoWo);
      const float mean = s * inv_n;
      gn_stat[tid >> 3] = mean;
      gn_stat[64 + (tid >> 3)] = rsqrtf(fmaxf(q * inv_n - mean * mean, 0.f) + a.gnf_eps);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int k = tid + 256 * i;
      if (k < a.K) {
        const int g = k / cpg;
        const float sc = gnf_g[i] * gn_stat[64 + g];
        const half_t mh = (half_t)gn_stat[g];
        gn_tab[k] = mh;
        gn_tab[a.K + k] = (half_t)sc;
        gn_tab[2 * a.K + k] = (half_t)(gnf_b[i] - (gn_stat[g] - (float)mh) * sc);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  wait_vmcnt_barrier<(D - 1) * PER>();                      // tile 0 has landed for every wave (GNF: and the table is visible)
  read_step(fA, 0);
  int rd_stage = 0;                                         // ring stage of the step being multiplied
  auto body = [&](Frags& cur, Frags& nxt, auto last) {
    constexpr bool LAST = decltype(last)::value;
    if constexpr (LAST) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      mfma_step(cur);
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");    // this wave's fragments of the current step are in registers
      wait_vmcnt_barrier<(D - 2) * PER>();                  // the next tile has landed; nobody reads stage rd_stage any more
      const int nstage = (rd_stage + 1 == D) ? 0 : rd_stage + 1;
      read_step(nxt, nstage);                               // fragments of the next step -> the other register set
      issue_tile();                                         // tile (step + D) -> the stage just freed
      mfma_step(cur);
      rd_stage = nstage;
      if constexpr (!LNF && !GNF) {
        constexpr int NM = KK * TM * TN, NR = KK * (TM + TN);
        constexpr int RPM = (NR + NM / 2 - 1) / (NM / 2);   // reads behind each MFMA of the first half
#pragma unroll
        for (int g = 0; g < NM / 2; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, RPM, 0);
        }
#pragma unroll
        for (int g = 0; g < NM - NM / 2; ++g) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x010, (PER + NM - NM / 2 - 1) / (NM - NM / 2), 0);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
  {
    using Tt = std::true_type;
    using Ff = std::false_type;
    int st = 0;
    for (; st + 2 < T; st += 2) {
      body(fA, fB, Ff{});
      body(fB, fA, Ff{});
    }
    if (T - st == 2) {
      body(fA, fB, Ff{});
      body(fB, fA, Tt{});
    } else {
      body(fA, fB, Tt{});
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the zero-sized tail DMAs too: the LDS is about to be re-used

  float ln_a[TM] = {}, ln_b[TM] = {};
  if constexpr (LNF) {
    const float inv_k = 1.0f / (float)a.K;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const float s1 = xor32_sum(ln_s1[i]), s2 = xor32_sum(ln_s2[i]);
      const float mean = s1 * inv_k;
      const float var = fmaxf(s2 * inv_k - mean * mean, 0.f);
      ln_a[i] = rsqrtf(var + a.ln_eps);
      ln_b[i] = -ln_a[i] * mean;
    }
  }
  tile_epilogue<BM, BN, WGM, WGN, TM, TN, LNF, (RES_PRE ? RIT : 0)>(a, acc, ln_a, ln_b, smem, sconst, const_b, const_t, const_c, m_blk,
                                                                    n_blk, wave, split, temb_uniform, resv, use_resv);
}

template <int BM, int BN, int WGM, int WGN, int D, bool LNF, int DBG = 0, bool GNF = false>
__global__ __launch_bounds__(256, D == 2 ? 2 : 1) void gemm_pipe_kernel(IgemmArgs a) {
  gemm_pipe_body<BM, BN, WGM, WGN, D, LNF, DBG, GNF>(a, blockIdx.x, blockIdx.y);
}

// ---------------------------------------------------------------------------------------------
// GroupNorm and an INDEPENDENT 1x1 GEMM side by side in one launch (round 5).  A resnet with a channel change (unet.py:470-489)
// runs norm1 -> conv1 -> norm2 -> conv2 on one chain and conv_shortcut(x) on another that meets it in conv2's residual add; as
// separate launches the 14 shortcut GEMMs of the SD2.1 step cost 10-20 us each and the norm1 launches beside them 8-23 us, each
// on a fraction of the chip (the single-launch GroupNorm runs on 64 workgroups, the shortcut GEMM at M = 512 on 160).  A side
// stream inside the captured graph costs more than it hides (round 4: fork / join pairs, +3.8 %).  Here ONE grid holds both:
// blocks [0, n_gn) run the GroupNorm body (gn_body.inc, the code of the stand-alone launch), blocks [n_gn_pad, ...) the pipelined
// GEMM body on the same input tensors (second reader: L2 / Infinity Cache hits).  No dependence between the two halves, no extra
// synchronisation; the consumer of both (conv1 / conv2) is a later launch as before.
// ---------------------------------------------------------------------------------------------
struct GnSideArgs {
  const half_t* x0;
  const half_t* x1;
  int C0, C1;
  const float* partial;
  int slabs;            // apply: entries to fold
  const float* gamma;
  const float* beta;
  half_t* y;
  int HW, G;
  float eps;
  int silu, pix_per_block;
  int gx, gy;           // the GroupNorm's own grid
  int n_gn_pad;         // first block of the GEMM half (n_gn rounded up to a multiple of 8: keeps the GEMM's XCD mapping)
};

template <int D>
__global__ __launch_bounds__(256) void gn_apply_side_kernel(GnSideArgs g, IgemmArgs a) {
  const int v = blockIdx.x;
  if (v < g.gx * g.gy) {
    groupnorm_apply_body(g.x0, g.C0, g.x1, g.C1, g.partial, g.slabs, g.gamma, g.beta, g.y, g.HW, g.G, g.eps, g.silu, g.pix_per_block,
                         v % g.gx, v / g.gx);
  } else if (v >= g.n_gn_pad) {
    gemm_pipe_body<64, 64, 2, 2, D, false>(a, v - g.n_gn_pad, 0);
  }
}

template <int VW, int D>
__global__ __launch_bounds__(256) void gn_fused_side_kernel(GnSideArgs g, IgemmArgs a) {
  const int v = blockIdx.x;
  if (v < g.gx * g.gy) {
    groupnorm_fused_body<VW, 256>(g.x0, g.C0, g.x1, g.C1, g.gamma, g.beta, g.y, g.HW, g.G, g.eps, g.silu, v % g.gx, v / g.gx);
  } else if (v >= g.n_gn_pad) {
    gemm_pipe_body<64, 64, 2, 2, D, false>(a, v - g.n_gn_pad, 0);
  }
}

// split-K combine + the same epilogue (bias, temb broadcast, residual) -> fp16.
// Round 5: every load of an item - up to eight slabs, bias, timestep embedding, residual - is requested before the first one is
// used (the slab loop with its run-time trip count paid one dependent round trip per slab and another for the epilogue operands:
// 4-6 in a row for 1 MB, 44 launches per step); summed in the slice order as before - bit-identical results.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(IgemmArgs a) {
  constexpr int SB = 8;
  const size_t total4 = (size_t)a.M * a.N / 4;
  const size_t slab = (size_t)a.M * a.N;
  const floatx4 z4 = {0.f, 0.f, 0.f, 0.f};
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total4;
       idx += (size_t)gridDim.x * blockDim.x) {
    const size_t e0 = idx * 4;
    const int m = (int)(e0 / a.N);
    const int n = (int)(e0 - (size_t)m * a.N);
    floatx4 p[SB];
#pragma unroll
    for (int z = 0; z < SB; ++z) p[z] = (z < a.splitk) ? *reinterpret_cast<const floatx4*>(a.partial + (size_t)z * slab + e0) : z4;
    const floatx4 bv = a.bias ? *reinterpret_cast<const floatx4*>(a.bias + n) : z4;
    const floatx4 tv = a.temb ? *reinterpret_cast<const floatx4*>(a.temb + (size_t)(m / a.HoWo) * a.temb_stride + n) : z4;
    const half4 hz = {0, 0, 0, 0};
    const half4 rr = a.res ? *reinterpret_cast<const half4*>(a.res + e0) : hz;
    floatx4 s = z4;
#pragma unroll
    for (int z = 0; z < SB; ++z)
      if (z < a.splitk) s += p[z];
    for (int z = SB; z < a.splitk; ++z) s += *reinterpret_cast<const floatx4*>(a.partial + (size_t)z * slab + e0);
    if (a.bias) s += bv;
    if (a.temb) s += tv;
    if (a.res) {
      s[0] += (float)rr[0];
      s[1] += (float)rr[1];
      s[2] += (float)rr[2];
      s[3] += (float)rr[3];
    }
    half4 o = {(half_t)s[0], (half_t)s[1], (half_t)s[2], (half_t)s[3]};
    out_store(reinterpret_cast<half4*>(a.out + e0), o);
  }
}

// The same combine for a tensor whose consumer is a GroupNorm (8x8 / 16x16 levels: every conv there is split-K or the
// weight-streaming kernel, so its GroupNorm statistics cannot come out of a tile epilogue): one workgroup per (PPB pixels of a
// sample) x all channels, per-(pixel, 4-channel quad) sums of the fp16-rounded outputs through LDS, then one thread per group adds
// its quads in a fixed order and writes entry blockIdx.x of gn_partial [B][G][kGnMaxSlabs][2] - the format the tile epilogues
// write, so the GroupNorm runs its fully parallel apply pass instead of the 64-workgroup two-pass kernel.
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(IgemmArgs a, int ppb) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* qs = reinterpret_cast<float*>(smem);            // [ppb][N / 4] (sum, sumsq) pairs
  const int NQ = a.N >> 2, t = threadIdx.x, b = blockIdx.y;
  const int p0 = blockIdx.x * ppb;
  const int items = ppb * NQ;
  for (int id = t; id < items; id += 256) {
    const int pl = id / NQ, qd = id - pl * NQ;
    const int p = p0 + pl;
    float fs = 0.f, fq = 0.f;
    if (p < a.HoWo) {
      const int m = b * a.HoWo + p, n = 4 * qd;
      const size_t e0 = (size_t)m * a.N + n;
      floatx4 s = {0.f, 0.f, 0.f, 0.f};
      for (int z = 0; z < a.splitk; ++z) s += *reinterpret_cast<const floatx4*>(a.partial + (size_t)z * a.M * a.N + e0);
      if (a.bias) s += *reinterpret_cast<const floatx4*>(a.bias + n);
      if (a.temb) s += *reinterpret_cast<const floatx4*>(a.temb + (size_t)b * a.temb_stride + n);
      if (a.res) {
        const half4 rr = *reinterpret_cast<const half4*>(a.res + e0);
        s[0] += (float)rr[0];
        s[1] += (float)rr[1];
        s[2] += (float)rr[2];
        s[3] += (float)rr[3];
      }
      const half4 o = {(half_t)s[0], (half_t)s[1], (half_t)s[2], (half_t)s[3]};
      out_store(reinterpret_cast<half4*>(a.out + e0), o);
      const float f0 = (float)o[0], f1 = (float)o[1], f2 = (float)o[2], f3 = (float)o[3];
      fs = (f0 + f1) + (f2 + f3);
      fq = fmaf(f0, f0, fmaf(f1, f1, fmaf(f2, f2, f3 * f3)));
    }
    qs[2 * id] = fs;
    qs[2 * id + 1] = fq;
  }
  __syncthreads();
  if (t < a.gn_G) {
    const int q0 = t * (a.gn_cpg >> 2), q1 = q0 + (a.gn_cpg >> 2);
    float s = 0.f, q = 0.f;
    for (int pl = 0; pl < ppb; ++pl)
      for (int qd = q0; qd < q1; ++qd) {
        s += qs[2 * (pl * NQ + qd)];
        q += qs[2 * (pl * NQ + qd) + 1];
      }
    float* dst = a.gn_partial + (((size_t)b * a.gn_G + t) * kGnMaxSlabs + blockIdx.x) * 2;
    dst[0] = s;
    dst[1] = q;
  }
}

// ---- generic direct convolution: any shape, one thread per output element (tiny/odd layers) ----
__global__ __launch_bounds__(256) void conv_generic_kernel(IgemmArgs a, int silu_out) {
  const bool geglu = a.out_mode == kOutGeglu;
  const int NO = geglu ? a.N / 2 : a.N;
  const size_t total = (size_t)a.M * NO;
  const int Hup = a.Hi * a.up, Wup = a.Wi * a.up, upshift = a.up >> 1;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(idx / NO), no = (int)(idx - (size_t)m * NO);
    const int b = m / a.HoWo;
    const int rem = m - b * a.HoWo;
    const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
    auto dot = [&](int n) {
      float acc = 0.f;
      for (int ky = 0; ky < a.ksize; ++ky)
        for (int kx = 0; kx < a.ksize; ++kx) {
          const int iy = oy * a.stride - a.pad + ky, ix = ox * a.stride - a.pad + kx;
          if (iy < 0 || iy >= Hup || ix < 0 || ix >= Wup) continue;
          const size_t pix = (size_t)b * a.Hi * a.Wi + (size_t)(iy >> upshift) * a.Wi + (ix >> upshift);
          const half_t* wrow = a.w + (size_t)n * a.K + (size_t)(ky * a.ksize + kx) * a.Ctot;
          const half_t* p0 = a.x0 + pix * a.C0;
          for (int c = 0; c < a.C0; ++c) acc += (float)p0[c] * (float)wrow[c];
          if (a.C1) {
            const half_t* p1 = a.x1 + pix * a.C1;
            for (int c = 0; c < a.C1; ++c) acc += (float)p1[c] * (float)wrow[a.C0 + c];
          }
        }
      return acc + (a.bias ? a.bias[n] : 0.f);
    };
    if (geglu) {   // interleaved rows: 32 value channels then their 32 gate channels
      const int nv = (no / 32) * 64 + (no % 32);
      a.out[idx] = (half_t)(dot(nv) * gelu_erf(dot(nv + 32)));
      continue;
    }
    float acc = dot(no);
    if (a.temb) acc += a.temb[(size_t)b * a.temb_stride + no];
    if (a.res) acc += (float)a.res[idx];
    if (silu_out) acc = acc / (1.f + __expf(-acc));
    if (a.out_mode == kOutHalfT)
      a.out[((size_t)b * a.N + no) * a.ldT + rem] = (half_t)acc;
    else
      a.out[idx] = (half_t)acc;
  }
}

// ---- tiny input-channel count (conv_in 4->320, K = 36): 16 output pixels per workgroup, the
// im2col patches live in LDS (broadcast reads), each thread keeps one output channel's K weights
// in registers.  Replaces the generic one-thread-per-output kernel (95 us -> a few us).
constexpr int SC_PIX = 16, SC_KMAX = 72;
__global__ __launch_bounds__(256) void conv_small_cin_kernel(IgemmArgs a, int silu_out) {
  __shared__ float patch[SC_PIX][SC_KMAX];
  const int m0 = blockIdx.x * SC_PIX;
  const int Hup = a.Hi * a.up, Wup = a.Wi * a.up, upshift = a.up >> 1;
  // every column of a patch row is written - K .. SC_KMAX - 1 with zeros: the dot product below runs over all SC_KMAX columns against
  // zero weights there, and 0 * (whatever the LDS held: NaN bit patterns on a fresh box) is NaN (round 6: the tiny test UNet's
  // K = 48 to_k projection failed as the first launch of a process, tools/ubench/poison.hip + SD_NAN_TRACE)
  for (int idx = threadIdx.x; idx < SC_PIX * SC_KMAX; idx += blockDim.x) {
    const int p = idx / SC_KMAX, k = idx - p * SC_KMAX;
    const int m = m0 + p;
    float v = 0.f;
    if (m < a.M && k < a.K) {
      const int b = m / a.HoWo;
      const int rem = m - b * a.HoWo;
      const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
      const int tap = k / a.Ctot, c = k - tap * a.Ctot;
      const int ky = tap / a.ksize, kx = tap - ky * a.ksize;
      const int iy = oy * a.stride - a.pad + ky, ix = ox * a.stride - a.pad + kx;
      if (iy >= 0 && iy < Hup && ix >= 0 && ix < Wup) {
        const size_t pix = (size_t)b * a.Hi * a.Wi + (size_t)(iy >> upshift) * a.Wi + (ix >> upshift);
        v = (c < a.C0) ? (float)a.x0[pix * a.C0 + c] : (float)a.x1[pix * a.C1 + (c - a.C0)];
      }
    }
    patch[p][k] = v;
  }
  __syncthreads();
  for (int n = threadIdx.x; n < a.N; n += blockDim.x) {
    float w[SC_KMAX];
#pragma unroll
    for (int k = 0; k < SC_KMAX; ++k) w[k] = (k < a.K) ? (float)a.w[(size_t)n * a.K + k] : 0.f;
    const float bv = a.bias ? a.bias[n] : 0.f;
    for (int p = 0; p < SC_PIX; ++p) {
      const int m = m0 + p;
      if (m >= a.M) break;
      float acc = bv;
#pragma unroll
      for (int k = 0; k < SC_KMAX; ++k) acc += w[k] * patch[p][k];   // (columns beyond K: zero weights x zero patch)
      if (a.temb) acc += a.temb[(size_t)(m / a.HoWo) * a.temb_stride + n];
      if (a.res) acc += (float)a.res[(size_t)m * a.N + n];
      if (silu_out) acc = acc / (1.f + __expf(-acc));
      a.out[(size_t)m * a.N + n] = (half_t)acc;
    }
  }
}

// ---- 4 input channels (conv_in 4->320, VAE conv_in 4->512): K = 36 on the MFMA ----
// The scalar kernel above spends 30-40 us on SD2.1's conv_in (profiles/r02_final_op_profile.txt) for 0.2 GFLOP.  Here a
// workgroup builds the im2col rows of 128 output pixels (9 taps x 4 channels = 72 B each, zero-padded to three 16-deep
// MFMA steps) and 64 weight rows in LDS - same row swizzle as igemm_kernel - runs 6 MFMAs per wave and leaves through
// the shared tile epilogue (bias, residual = the ControlNet conditioning embedding, coalesced fp16 stores).
__global__ __launch_bounds__(256) void conv3x3_cin4_kernel(IgemmArgs a) {
  constexpr int BM = 128, BN = 64, ROWB = BK * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const Xs = smem;                       // [BM][64 halves]; only k < 48 is read
  char* const Ws = smem + BM * ROWB;           // [BN][64 halves]
  float* sconst = reinterpret_cast<float*>(smem + (BM + BN) * ROWB);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nbm = (a.M + BM - 1) / BM;
  const int bn_idx = blockIdx.x / nbm, bm_idx = blockIdx.x % nbm;
  const int m_blk = bm_idx * BM, n_blk = bn_idx * BN;
  typedef unsigned long long u64;
  // one thread per LDS row: threads 0..127 an im2col row, 128..191 a weight row; 8-B pieces at k = 4t, t = 0..11
  if (tid < BM + BN) {
    const bool is_x = tid < BM;
    const int r = is_x ? tid : tid - BM;
    char* row = (is_x ? Xs : Ws) + r * ROWB;
    const int sw = (r >> 1) & 7;
    u64 v[12];
#pragma unroll
    for (int t = 0; t < 12; ++t) v[t] = 0ull;
    if (is_x) {
      const int m = m_blk + r;
      if (m < a.M) {
        const int b = m / a.HoWo, rem = m - b * a.HoWo;
        const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = oy - 1 + t / 3, ix = ox - 1 + t % 3;
          if (iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi)
            v[t] = *reinterpret_cast<const u64*>(a.x0 + ((size_t)(b * a.Hi + iy) * a.Wi + ix) * 4);
        }
      }
    } else {
      const int n = n_blk + r;
      if (n < a.N) {
#pragma unroll
        for (int t = 0; t < 9; ++t) v[t] = *reinterpret_cast<const u64*>(a.w + (size_t)n * 36 + t * 4);
      }
    }
#pragma unroll
    for (int t = 0; t < 12; ++t)   // logical 16-B chunk t/2 lives in physical slot (t/2) ^ sw
      *reinterpret_cast<u64*>(row + (((t >> 1) ^ sw) * 16) + (t & 1) * 8) = v[t];
  }
  float const_b = 0.f;
  if (tid < BN && n_blk + tid < a.N && a.bias) const_b = a.bias[n_blk + tid];
  __syncthreads();
  const int wm = wave >> 1, wn = wave & 1;
  const int frow = lane & 31, hi = lane >> 5, fsw = (frow >> 1) & 7;
  floatx16 acc[2][1];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][0][r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < 3; ++ks) {
    const int koff = ((2 * ks + hi) ^ fsw) * 16;
    const half8 wf = *reinterpret_cast<const half8*>(Ws + (wn * 32 + frow) * ROWB + koff);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const half8 xf = *reinterpret_cast<const half8*>(Xs + ((wm * 2 + i) * 32 + frow) * ROWB + koff);
      acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf, xf, acc[i][0], 0, 0, 0);
    }
  }
  const float ln0[2] = {0.f, 0.f};
  tile_epilogue<BM, BN, 2, 2, 2, 1, false>(a, acc, ln0, ln0, smem, sconst, const_b, 0.f, 0.f, m_blk, n_blk, wave, 0, false);
}

// ---- N <= 8 output channels (conv_out 320->4): one wavefront per output pixel ----
template <int NMAX>
__global__ __launch_bounds__(256) void conv_small_n_kernel(IgemmArgs a, float* out_nchw) {
  const int lane = threadIdx.x & 63;
  const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (m >= a.M) return;
  const int b = m / a.HoWo;
  const int rem = m - b * a.HoWo;
  const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
  const int Hup = a.Hi * a.up, Wup = a.Wi * a.up, upshift = a.up >> 1;
  float acc[NMAX];
#pragma unroll
  for (int n = 0; n < NMAX; ++n) acc[n] = 0.f;
  const int chunks = a.Ctot >> 3;
  for (int tap = 0; tap < a.ksize * a.ksize; ++tap) {
    const int ky = tap / a.ksize, kx = tap - ky * a.ksize;
    const int iy = oy * a.stride - a.pad + ky, ix = ox * a.stride - a.pad + kx;
    if (iy < 0 || iy >= Hup || ix < 0 || ix >= Wup) continue;   // wave-uniform
    const size_t pix = (size_t)b * a.Hi * a.Wi + (size_t)(iy >> upshift) * a.Wi + (ix >> upshift);
    for (int ch = lane; ch < chunks; ch += 64) {
      const int c = ch * 8;
      half8 xv = (c < a.C0) ? *reinterpret_cast<const half8*>(a.x0 + pix * a.C0 + c)
                            : *reinterpret_cast<const half8*>(a.x1 + pix * a.C1 + (c - a.C0));
#pragma unroll
      for (int n = 0; n < NMAX; ++n) {
        if (n < a.N) {
          half8 wv = *reinterpret_cast<const half8*>(a.w + (size_t)n * a.K + (size_t)tap * a.Ctot + c);
#pragma unroll
          for (int e = 0; e < 8; ++e) acc[n] += (float)xv[e] * (float)wv[e];
        }
      }
    }
  }
#pragma unroll
  for (int n = 0; n < NMAX; ++n) {
    float v = acc[n];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    acc[n] = v;
  }
  if (lane == 0) {
    for (int n = 0; n < a.N; ++n) {
      float v = acc[n] + (a.bias ? a.bias[n] : 0.f);
      if (out_nchw)
        out_nchw[((size_t)b * a.N + n) * a.HoWo + rem] = v;
      else
        a.out[(size_t)m * a.N + n] = (half_t)v;
    }
  }
}

// ---- N <= 4 output channels, 3x3 / stride 1 (the UNet's conv_out 320 -> 4, the VAE decoder's 128 -> 3): FOUR pixels of a row per
// lane group (round 5).  The one-wave-per-pixel kernel above walks its nine taps as nine dependent load -> FMA rounds (32 us for
// 8192 pixels: pure latency); here a group of LPC lanes (one per 8-channel chunk) requests the 3 x 6 input patch of four
// neighbouring pixels and the 9 x N weight chunks up front - 18 + 9 N independent 16-byte loads in flight per lane -, multiplies
// with v_dot2_f32_f16 and folds the LPC partial sums by butterfly.  64 / LPC groups per wave (C = 128: four groups of 16 lanes).
template <int LPC>
__global__ __launch_bounds__(256) void conv3x3_small_n_rows_kernel(IgemmArgs a, float* out_nchw) {
  constexpr int PG = 64 / LPC, PX = 4, NMAX = 4;
  const int lane = threadIdx.x & 63;
  const int sub = lane / LPC, cl = lane % LPC;
  const int gpr = a.Wo / PX;                                   // pixel groups per row
  const int total = a.B * a.Ho * gpr;
  const int g = (blockIdx.x * 4 + (threadIdx.x >> 6)) * PG + sub;
  const bool live = g < total && cl < (a.Ctot >> 3);
  const int gg = g < total ? g : total - 1;
  const int row = gg / gpr, gx = gg - row * gpr;
  const int b = row / a.Ho, oy = row - b * a.Ho, ox0 = gx * PX;
  const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
  half8 xv[3][PX + 2];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int iy = oy - 1 + r;
#pragma unroll
    for (int c = 0; c < PX + 2; ++c) {
      const int ix = ox0 - 1 + c;
      const bool ok = live && iy >= 0 && iy < a.Hi && ix >= 0 && ix < a.Wi;
      xv[r][c] = ok ? *reinterpret_cast<const half8*>(a.x0 + (((size_t)b * a.Hi + iy) * a.Wi + ix) * a.C0 + cl * 8) : z;
    }
  }
  half8 wv[NMAX][9];
#pragma unroll
  for (int n = 0; n < NMAX; ++n)
#pragma unroll
    for (int t = 0; t < 9; ++t)
      wv[n][t] = (live && n < a.N) ? *reinterpret_cast<const half8*>(a.w + (size_t)n * a.K + (size_t)t * a.Ctot + cl * 8) : z;
  float acc[NMAX][PX];
#pragma unroll
  for (int n = 0; n < NMAX; ++n)
#pragma unroll
    for (int px = 0; px < PX; ++px) acc[n][px] = 0.f;
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int px = 0; px < PX; ++px)
#pragma unroll
        for (int n = 0; n < NMAX; ++n)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const half2v x2 = {xv[r][px + kx][2 * e], xv[r][px + kx][2 * e + 1]};
            const half2v w2 = {wv[n][r * 3 + kx][2 * e], wv[n][r * 3 + kx][2 * e + 1]};
            acc[n][px] = __builtin_amdgcn_fdot2(x2, w2, acc[n][px], false);
          }
#pragma unroll
  for (int n = 0; n < NMAX; ++n)
#pragma unroll
    for (int px = 0; px < PX; ++px) {
      float v = acc[n][px];
#pragma unroll
      for (int o = LPC / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
      acc[n][px] = v;
    }
  // lanes 0-15 of the group store one (channel, pixel) each
  const int sel = cl & 15;
  float v = 0.f;
#pragma unroll
  for (int n = 0; n < NMAX; ++n)
#pragma unroll
    for (int px = 0; px < PX; ++px) v = (sel == n * PX + px) ? acc[n][px] : v;
  const int n = sel >> 2, px = sel & 3;
  if (g < total && cl < 16 && n < a.N) {
    v += a.bias ? a.bias[n] : 0.f;
    const int rem = oy * a.Wo + ox0 + px;
    if (out_nchw)
      out_nchw[((size_t)b * a.N + n) * a.HoWo + rem] = v;
    else
      a.out[((size_t)b * a.HoWo + rem) * a.N + n] = (half_t)v;
  }
}

const half_t* device_zero_chunk() {   // allocated on first use (always outside graph capture: eager warm-up run)
  static half_t* p = nullptr;
  if (!p) {
    SD_HIP(hipMalloc(reinterpret_cast<void**>(&p), 256));
    SD_HIP(hipMemset(p, 0, 256));
    SD_HIP(hipDeviceSynchronize());
  }
  return p;
}

IgemmArgs make_args(const ConvDesc& d) {
  IgemmArgs a{};
  a.x0 = d.x0;
  a.x1 = d.x1;
  a.w = d.w;
  a.bias = d.bias;
  a.temb = d.temb;
  a.res = d.res;
  a.out = d.out;
  a.partial = nullptr;
  a.C0 = d.C0;
  a.C1 = d.x1 ? d.C1 : 0;
  a.Ctot = a.C0 + a.C1;
  a.B = d.B;
  a.Hi = d.Hi;
  a.Wi = d.Wi;
  a.Ho = d.Ho;
  a.Wo = d.Wo;
  a.HoWo = d.Ho * d.Wo;
  a.ksize = d.ksize;
  a.stride = d.stride;
  a.up = d.up;
  a.pad = d.pad >= 0 ? d.pad : d.ksize / 2;
  a.M = d.B * d.Ho * d.Wo;
  a.N = d.N;
  a.K = d.ksize * d.ksize * a.Ctot;
  a.temb_stride = d.temb_stride;
  a.nk_total = a.K / BK;
  a.nk_per_split = a.nk_total;
  a.splitk = 1;
  a.slab = 0;
  a.out_mode = d.out_mode;
  a.ldT = d.ldT;
  a.debug = d.debug;
  a.prof = d.prof;
  a.zeros = device_zero_chunk();
  a.tiles_x = cdiv(d.Wo, 16);   // halo kernels: stride 1, so output = (upsampled) input extent
  a.tiles_y = cdiv(d.Ho, 8);
  a.ln_colsum = d.ln_colsum;
  a.ln_eps = d.ln_eps;
  a.n_trans = d.out_t ? d.n_trans : 0x7fffffff;
  a.ldo = d.out_t ? d.n_trans : d.N;
  a.out_t = d.out_t;
  a.vt_perm = d.out_t ? d.vt_perm : 0;
  static const int res_pre = tune_env_int("SD_RES_PREFETCH", 1) != 0;
  a.res_pre = res_pre;
  a.gn_partial = nullptr;
  a.gn_G = a.gn_cpg = a.gn_T = 0;
  a.gnf_partial = d.gnf_partial;
  a.gnf_gamma = d.gnf_gamma;
  a.gnf_beta = d.gnf_beta;
  a.gnf_eps = d.gnf_eps;
  a.gnf_G = d.gnf_groups;
  a.gnf_entries = d.gnf_entries;
  a.q_scale = d.q_scale;
  a.q_cols = d.out_t ? d.q_cols : 0;
  return a;
}

struct Plan {
  int tile;     // 1: 128x128, 2: 128x64, 3: 64x64, 4: 64x128
  int splitk;
  int staging = 0;   // from the tuned table (tile / 10): 0 two-stage LDS-DMA, 2 / 3 = 3- / 4-stage ring
};

// the K-split halo kernel also folds the nearest-x2 upsample into its gather and takes 8-pixel-wide images (half of each
// 8x16 tile is then padding: the 8x8 level streams weights, MFMA work is not what bounds it)
bool halo_ks_ok(const ConvDesc& d) {
  const int c1 = d.x1 ? d.C1 : 0;
  return d.ksize == 3 && d.stride == 1 && (d.up == 1 || d.up == 2) && d.pad < 0 && d.out_mode == kOutHalf && d.Wo >= 8 && d.Ho >= 8 &&
         d.C0 % BK == 0 && c1 % BK == 0 && d.N % 4 == 0;
}

bool gemm_pipe_ok(const IgemmArgs& a) {
  const size_t lim = (size_t)1 << 31;
  return a.ksize == 1 && a.stride == 1 && a.up == 1 && (size_t)a.M * std::max(a.C0, a.C1) * 2 < lim && (size_t)a.N * a.K * 2 < lim;
}

void tile_dims(int tile, int& bm, int& bn) {
  switch (tile) {
    case 7: bm = 128; bn = 64; break;    // halo conv (8x16-pixel tile), K-split waves + register double buffering
    case 1: bm = 128; bn = 128; break;
    case 2: bm = 128; bn = 64; break;
    case 3: bm = 64; bn = 64; break;
    default: bm = 64; bn = 128; break;
  }
}

// Heuristic (overridden per shape by the measured table in tuned_convs.inc when present):
// prefer the biggest tile whose grid, multiplied by the split-K it can afford (>= 16 K-tiles per
// split), still gives every CU work (>= ~1.5 workgroups per CU); deep-K small-M layers (8x8 /
// 16x16 levels stream weights: SURVEY.md 7.3(1)) end up split, shallow 1x1 GEMMs end up on the
// small tile with many workgroups.
// kind: 0 plain epilogue (bias / timestep embedding / residual), 1 LayerNorm-folded, 2 GEGLU (LayerNorm-folded or
// not), 3 fused q|k|v (LayerNorm-folded, V columns leave transposed).  staging: see launch_tile.
struct TunedConv { int kind, ksize, stride, up, ctot, n, m, tile, staging, splitk; };
#if __has_include("tuned_convs.inc")
static const TunedConv kTuned[] = {
#include "tuned_convs.inc"
};
static const int kNumTuned = sizeof(kTuned) / sizeof(kTuned[0]);
#else
static const TunedConv* kTuned = nullptr;
static const int kNumTuned = 0;
#endif

// SD_PLAN_TABLE=<file>: rows "{kind, ksize, stride, up, Ctot, N, M, tile, staging, splitk}" (the format of
// tuned_convs.inc) read at first use and consulted BEFORE the compiled-in table - how tools/tune_plans.py
// validates a freshly measured table in the same GPU session without a rebuild.
bool parse_plan_row(const char* line, TunedConv& r) {
  return sscanf(line, " {%d, %d, %d, %d, %d, %d, %d, %d, %d, %d}", &r.kind, &r.ksize, &r.stride, &r.up, &r.ctot, &r.n, &r.m, &r.tile,
                &r.staging, &r.splitk) == 10;
}
std::vector<TunedConv>& runtime_table() {
  static std::vector<TunedConv> table = [] {
    std::vector<TunedConv> t;
    const char* path = tune_env_set("SD_PLAN_TABLE") ? getenv("SD_PLAN_TABLE") : nullptr;
    if (!path) return t;
    FILE* f = fopen(path, "r");
    if (!f) {
      fprintf(stderr, "[sd] SD_PLAN_TABLE=%s cannot be opened - ignored\n", path);
      return t;
    }
    char line[512];
    while (fgets(line, sizeof(line), f)) {
      TunedConv r;
      if (parse_plan_row(line, r)) t.push_back(r);
    }
    fclose(f);
    fprintf(stderr, "[sd] SD_PLAN_TABLE=%s: %zu plans\n", path, t.size());
    return t;
  }();
  return table;
}

// measurement hook (sd_tune_set_candidate): while tile != 0 every conv whose constraints allow it runs this plan,
// so one profiled forward evaluates a candidate on every layer shape at once, in sequence (tools/tune_plans.py)
struct TuneCandidate { int tile = 0, staging = 0, splitk = 0; };
TuneCandidate g_tune;

int conv_kind(const ConvDesc& d) {
  if (d.out_t) return 3;
  if (d.out_mode == kOutGeglu) return 2;
  return d.ln_colsum ? 1 : 0;
}

Plan choose_plan(const ConvDesc& d, const IgemmArgs& a) {
  Plan p{d.tile, d.splitk};
  p.staging = d.staging;
  const bool geglu = d.out_mode == kOutGeglu;
  // the LayerNorm fold needs whole rows per workgroup, the fused q|k|v epilogue has no slab path
  const bool can_split = d.out_mode == kOutHalf && !d.ln_colsum && !d.out_t;
  // plan tiles: 1 = 128x128, 2 = 128x64, 3 = 64x64, 4 = 64x128 (igemm_kernel / gemm_pipe_kernel), 7 = the halo conv.  Codes 5 / 6
  // (round 2's first halo kernel) and 8 / 9 (256x128 / 256x256 GEMM tiles, measured in round 3 and selected nowhere) were removed
  // in round 4: a caller or table row that names them gets the heuristic.
  auto is_halo = [](int c) { return c == 7; };
  // plan tile 9 = the weight-streaming kernel of wstream.hip (needs the pre-tiled weights; staging 4 = four waves per workgroup,
  // anything else eight; its slab count follows from the wave count).  SD_WSTREAM=0 (with SD_TUNE) takes it out of every plan: A/B.
  static const int ws_mode = tune_env_int("SD_WSTREAM", 1);
  auto tile_ok = [&](int c) {
    if (c == 9) return ws_mode != 0 && d.w_tiled != nullptr && can_split && wstream_shape_ok(d);
    if (!((c >= 1 && c <= 4) || c == 7)) return false;
    if (is_halo(c)) return halo_ks_ok(d) && !d.ln_colsum && !d.out_t && !geglu;
    int bm, bn;
    tile_dims(c, bm, bn);
    if (geglu && c != 1 && c != 4) return false;       // GEGLU value/gate pairs need 64 n-columns per wave
    if (d.out_t && d.n_trans % bn != 0) return false;  // the q|k / v boundary must be a tile boundary
    return true;
  };
  if (!tile_ok(p.tile)) p.tile = 0;
  auto blocks_of = [&](int c) {
    int bm, bn;
    tile_dims(c, bm, bn);
    const long mt = is_halo(c) ? (long)a.B * a.tiles_x * a.tiles_y : cdiv(a.M, bm);
    return mt * cdiv(a.N, bn);
  };
  auto ksteps = [&](int c) { return is_halo(c) ? a.Ctot / BK : a.nk_total; };   // split-K granularity
  auto max_split = [&](int c) {
    // a split costs a dependent reduce launch (~6 us): worth it only while each split still runs >= 16 K
    // steps (measured: 20-step 1x1 GEMMs lose 2.5 us when split in two, 180-step 3x3 convs peak at 8 splits)
    const int per_min = is_halo(c) ? 2 : 16;     // halo splits whole 64-channel chunks (9 K steps each)
    int s = 1;
    while (can_split && s < 16 && ksteps(c) / (s * 2) >= per_min) s *= 2;
    return s;
  };
  const bool pinned = p.tile != 0 || p.splitk != 0 || p.staging != 0;   // the caller chose (operator-level A/B tests)
  if (!pinned && g_tune.tile != 0 && tile_ok(g_tune.tile)) {
    p.tile = g_tune.tile;
    p.staging = g_tune.staging;
    p.splitk = g_tune.splitk;
  } else if (!pinned) {
    const int kind = conv_kind(d);
    auto match = [&](const TunedConv& t) {
      return t.kind == kind && t.ksize == a.ksize && t.stride == a.stride && t.up == a.up && t.ctot == a.Ctot && t.n == a.N &&
             t.m == a.M && tile_ok(t.tile);
    };
    const TunedConv* hit = nullptr;
    for (const TunedConv& t : runtime_table())
      if (!hit && match(t)) hit = &t;
    for (int i = 0; i < kNumTuned && !hit; ++i)
      if (match(kTuned[i])) hit = &kTuned[i];
    if (hit) {
      p.tile = hit->tile;
      p.staging = hit->staging;
      p.splitk = hit->splitk;
    }
  }
  // (measured and dropped, round 3: sending every untuned 1x1 GEMM with M >= 8192 to the software-pipelined kernel's 128x64
  // tile - 10-25 % faster stand-alone at UNet batch 16 - made the batch-16 step 1.3 % and the batch-4 step 0.6 % SLOWER in
  // sequence; the pipelined kernel is only used where tools/tune_e2e.py accepted it end to end)
  // 8x8 level at any small batch: a 3x3 conv there is a weight stream (wstream.hip), whatever the table says for M = 128
  if (p.tile == 0 && tile_ok(9) && a.ksize == 3 && a.HoWo <= 64 && a.M <= 256) p.tile = 9;
  if (p.tile == 9) {
    p.splitk = 1;   // (the launch derives the slab count from the wave count)
    return p;
  }
  if (p.tile == 0 && tile_ok(7)) {
    // no measured plan for this shape: the K-split halo kernel with the 4-stage ring won every 3x3 / stride-1 shape that was
    // tuned (SD2.1-base, SDXL-base, SD1.5: tuned_convs.inc); split-K below as for the other kernels
    p.tile = 7;
    if (p.staging == 0) p.staging = 3;
  }
  if (p.tile == 0) {
    p.tile = 3;
    for (int c : {1, 2, 4, 3}) {
      if (!tile_ok(c)) continue;
      if (blocks_of(c) * max_split(c) >= 384 || c == 3) { p.tile = c; break; }
    }
    if (!tile_ok(p.tile)) p.tile = 4;
  }
  if (p.splitk == 0) {
    p.splitk = 1;
    const int ms = max_split(p.tile);
    const long want = p.tile == 7 ? 192 : 384;   // the pipelined halo kernel is at its best from ~160 workgroups (1 per CU)
    while (blocks_of(p.tile) * p.splitk < want && p.splitk < ms) p.splitk *= 2;
  }
  if (!can_split) p.splitk = 1;
  if (p.splitk > ksteps(p.tile)) p.splitk = ksteps(p.tile);
  if (p.splitk < 1) p.splitk = 1;
  return p;
}

// staging (the table's ring code): 0 = 2 weight stages (two workgroups per CU), 2 / 3 = 3 / 4 stages, 4 / 5 = 6 / 8 stages
// GNL: the [3][Ctot] fp16 GroupNorm table takes the place of the epilogue constants behind the ring (conv3x3_halo_ks_kernel)
constexpr size_t halo_gnl_lds_cap = 160 * 1024;
inline size_t halo_gnl_lds_bytes(int d, int ctot) {
  const size_t k_loop = halo_lds_bytes(64, d) - 64 * sizeof(float) + (((size_t)6 * ctot + 15) & ~(size_t)15);
  const size_t epilogue = 32 * 1024 + 128 * (64 + 8) * 2 + 64 * sizeof(float);
  return std::max(k_loop, epilogue);
}
template <int D, int DBG = 0, bool GNL = false>
void launch_halo_ks_d(const IgemmArgs& a, hipStream_t s) {
  const size_t lds = GNL ? halo_gnl_lds_bytes(D, a.Ctot) : halo_lds_bytes(64, D);
  static_assert(halo_lds_bytes(64, D) <= 160 * 1024, "LDS");
  auto k = conv3x3_halo_ks_kernel<D, DBG, GNL>;
  static DynLdsOnce once;
  once.set(k, GNL ? halo_gnl_lds_cap : lds);   // GNL: the table grows with Ctot - allow the full LDS once
  dim3 grid(a.B * a.tiles_x * a.tiles_y * cdiv(a.N, 64), a.splitk);
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
}

// plan tile 7; ring code as launch_halo (0 / 2 / 3 / 4 / 5 = 2 / 3 / 4 / 6 / 8 weight stages)
void launch_halo_ks(IgemmArgs a, int splitk, int staging, hipStream_t s) {
  const int nch = a.Ctot / BK;
  a.nk_total = nch;
  a.nk_per_split = cdiv(nch, splitk);
  a.splitk = cdiv(nch, a.nk_per_split);
  if (a.debug >= 32) {   // ablation builds of the 4-stage kernel: debug = 32 + DBG bits
    switch (a.debug - 32) {
      case 1: launch_halo_ks_d<4, 1>(a, s); return;
      case 2: launch_halo_ks_d<4, 2>(a, s); return;
      case 3: launch_halo_ks_d<4, 3>(a, s); return;
      case 4: launch_halo_ks_d<4, 4>(a, s); return;
      case 8: launch_halo_ks_d<4, 8>(a, s); return;
      case 12: launch_halo_ks_d<4, 12>(a, s); return;
      case 13: launch_halo_ks_d<4, 13>(a, s); return;
      case 14: launch_halo_ks_d<4, 14>(a, s); return;
      case 15: launch_halo_ks_d<4, 15>(a, s); return;
      default: break;
    }
  }
  if (a.gnf_partial) {   // GroupNorm in the loader: rings of 3 / 4 stages only (launch_conv checked the shape)
    if (staging >= 3 && halo_gnl_lds_bytes(4, a.Ctot) <= halo_gnl_lds_cap) launch_halo_ks_d<4, 0, true>(a, s);
    else launch_halo_ks_d<3, 0, true>(a, s);
    return;
  }
  if (staging >= 5) { launch_halo_ks_d<8>(a, s); return; }
  if (staging >= 4) { launch_halo_ks_d<6>(a, s); return; }
  if (staging >= 3) { launch_halo_ks_d<4>(a, s); return; }
  if (staging >= 2) { launch_halo_ks_d<3>(a, s); return; }
  launch_halo_ks_d<2>(a, s);
}

template <int BM, int BN, int WGM, int WGN, bool TRANS, bool GLDS, int NST, bool LNF = false, int KG = 1>
void launch_variant(const IgemmArgs& a, hipStream_t s) {
  const size_t lds = (size_t)KG * NST * (BM + BN) * (GLDS ? BK : LDS_ROW) * sizeof(half_t) + 2 * BN * sizeof(float);
  static_assert((size_t)KG * NST * (BM + BN) * BK * sizeof(half_t) + 2 * BN * sizeof(float) <= 160 * 1024, "LDS");
  static_assert((size_t)BN * (BM + 8) <= (size_t)NST * (BM + BN) * (GLDS ? BK : LDS_ROW), "transposed staging fits");
  static_assert((size_t)BM * (BN + 8) * 2 + 16 + (kGnScratchFloats + 2 * BN) * sizeof(float) <=
                    (size_t)NST * (BM + BN) * (GLDS ? BK : LDS_ROW) * sizeof(half_t),
                "GroupNorm statistics scratch fits behind the staged tile");
  dim3 grid(cdiv(a.M, BM) * cdiv(a.N, BN), a.splitk);
  auto k = igemm_kernel<BM, BN, WGM, WGN, TRANS, GLDS, NST, 0, LNF, KG>;
  static DynLdsOnce once;   // per instantiation, per device
  once.set(k, lds);
  hipLaunchKernelGGL(k, grid, dim3(256 * KG), lds, s, a);
}

template <int BM, int BN, int DBG>
void launch_debug(const IgemmArgs& a, hipStream_t s) {
  const size_t lds = (size_t)2 * (BM + BN) * BK * sizeof(half_t) + 2 * BN * sizeof(float);
  dim3 grid(cdiv(a.M, BM) * cdiv(a.N, BN), a.splitk);
  auto k = igemm_kernel<BM, BN, 2, 2, false, true, 2, DBG>;
  SD_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
}
template <int BM, int BN>
bool launch_debug_mode(const IgemmArgs& a, int dbg, hipStream_t s) {
  switch (dbg) {
    case 4: launch_debug<BM, BN, 4>(a, s); return true;
    case 5: launch_debug<BM, BN, 5>(a, s); return true;
    case 6: launch_debug<BM, BN, 6>(a, s); return true;
    case 14: launch_debug<BM, BN, 14>(a, s); return true;
    case 22: launch_debug<BM, BN, 22>(a, s); return true;
    case 30: launch_debug<BM, BN, 30>(a, s); return true;
    default: return false;
  }
}

// staging: 0 = LDS-DMA 2 stages, 1 = register staging (A/B reference), 2 / 3 / 4 / 5 = LDS-DMA ring of
// 3 / 4 / 6 / 8 stages (a ring that would not fit the 160 KB of LDS falls back to the deepest one that does),
constexpr size_t kLdsBudget = 160 * 1024;
template <int BM, int BN>
constexpr size_t ring_bytes(int nst) {
  return (size_t)nst * (BM + BN) * BK * sizeof(half_t);
}
template <int BM, int BN>
constexpr bool ring_fits(int nst) {
  return (size_t)nst * (BM + BN) * BK * sizeof(half_t) + 2 * BN * sizeof(float) <= kLdsBudget;
}
// staging 6 / 7 / 8: the software-pipelined 1x1 GEMM kernel with a ring of 3 / 4 / 2 stages (1x1, stride 1, non-transposed
// output; 32-bit buffer offsets); anything else that asks for them runs the 4-stage ring of igemm_kernel
template <int BM, int BN, int WGM, int WGN, int D, bool LNF>
void launch_pipe_dbg(const IgemmArgs& a, hipStream_t s) {   // ablation builds (a.debug = 64 + bits)
  constexpr size_t lds = (size_t)D * (BM + BN) * BK * sizeof(half_t) + 2 * BN * sizeof(float);
  dim3 grid(cdiv(a.M, BM) * cdiv(a.N, BN), a.splitk);
  if (a.debug == 65) {
    auto k = gemm_pipe_kernel<BM, BN, WGM, WGN, D, LNF, 1>;
    SD_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
  } else {
    auto k = gemm_pipe_kernel<BM, BN, WGM, WGN, D, LNF, 2>;
    SD_HIP(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
  }
}

template <int BM, int BN, int WGM, int WGN, int D, bool LNF>
void launch_pipe(const IgemmArgs& a, hipStream_t s) {
  constexpr size_t lds = (size_t)D * (BM + BN) * BK * sizeof(half_t) + 2 * BN * sizeof(float);
  static_assert(lds <= kLdsBudget, "LDS");
  static_assert((size_t)BM * (BN + 8) * 2 + 16 + (kGnScratchFloats + 2 * BN) * sizeof(float) <= (size_t)D * (BM + BN) * BK * sizeof(half_t),
                "staged tile + GroupNorm statistics scratch fit the K-loop buffers");
  static_assert((size_t)BN * (BM + 8) <= (size_t)D * (BM + BN) * BK, "transposed staging fits");
  dim3 grid(cdiv(a.M, BM) * cdiv(a.N, BN), a.splitk);
  if constexpr (D == 2 && !LNF) {   // the ablation builds exist for the 2-stage ring of the plain epilogue only
    if (a.debug == 65 || a.debug == 66) {
      launch_pipe_dbg<BM, BN, WGM, WGN, D, LNF>(a, s);
      return;
    }
  }
  if constexpr (BM == 64 && BN == 64 && !LNF) {   // GroupNorm-folded proj_in: 64 x 64 tile only (launch_conv routes it here)
    if (a.gnf_partial) {
      const size_t lds_g = lds + 128 * sizeof(float) + (size_t)3 * a.K * sizeof(half_t);
      auto kg = gemm_pipe_kernel<BM, BN, WGM, WGN, D, false, 0, true>;
      static DynLdsOnce once_g;
      once_g.set(kg, lds + 128 * sizeof(float) + (size_t)3 * 2048 * sizeof(half_t));   // K <= 2048 (launch_conv)
      hipLaunchKernelGGL(kg, grid, dim3(256), lds_g, s, a);
      return;
    }
  }
  auto k = gemm_pipe_kernel<BM, BN, WGM, WGN, D, LNF>;
  static DynLdsOnce once;
  once.set(k, lds);
  hipLaunchKernelGGL(k, grid, dim3(256), lds, s, a);
}
template <int BM, int BN, int WGM, int WGN, bool LNF>
void launch_ring(const IgemmArgs& a, int staging, hipStream_t s) {
  // staging 12 / 13: igemm_kernel's 3- / 4-stage ring with the in-workgroup split-K (two K groups of four waves, KG = 2) where two
  // rings fit the LDS and the epilogue is the plain one; anything else that names them gets the same ring without it
  if (staging == 12 || staging == 13) {
    if constexpr (!LNF) {
      if (a.nk_per_split >= 4) {
        if (staging == 13) {
          if constexpr (2 * ring_bytes<BM, BN>(4) + 2 * BN * sizeof(float) <= kLdsBudget) { launch_variant<BM, BN, WGM, WGN, false, true, 4, false, 2>(a, s); return; }
        }
        if constexpr (2 * ring_bytes<BM, BN>(3) + 2 * BN * sizeof(float) <= kLdsBudget) { launch_variant<BM, BN, WGM, WGN, false, true, 3, false, 2>(a, s); return; }
      }
    }
    staging -= 10;
  }
  if (staging == 8 && gemm_pipe_ok(a)) {   // 2-stage ring, two workgroups per CU (three on 128 x 64): profiles/r03_exp_pipe_d2.txt
    launch_pipe<BM, BN, WGM, WGN, 2, LNF>(a, s);
    return;
  }
  if ((staging == 6 || staging == 7) && gemm_pipe_ok(a)) {
    if (staging == 7) {
      if constexpr (ring_fits<BM, BN>(4)) { launch_pipe<BM, BN, WGM, WGN, 4, LNF>(a, s); return; }
    }
    launch_pipe<BM, BN, WGM, WGN, 3, LNF>(a, s);
    return;
  }
  if (staging >= 6) staging = 3;
  if (staging >= 5) {
    if constexpr (ring_fits<BM, BN>(8)) { launch_variant<BM, BN, WGM, WGN, false, true, 8, LNF>(a, s); return; }
  }
  if (staging >= 4) {
    if constexpr (ring_fits<BM, BN>(6)) { launch_variant<BM, BN, WGM, WGN, false, true, 6, LNF>(a, s); return; }
  }
  if (staging >= 3) {
    if constexpr (ring_fits<BM, BN>(4)) { launch_variant<BM, BN, WGM, WGN, false, true, 4, LNF>(a, s); return; }
  }
  if (staging >= 2) { launch_variant<BM, BN, WGM, WGN, false, true, 3, LNF>(a, s); return; }
  launch_variant<BM, BN, WGM, WGN, false, true, 2, LNF>(a, s);
}

template <int BM, int BN, int WGM, int WGN>
void launch_tile(const IgemmArgs& a, bool trans, int staging, hipStream_t s) {
  if (a.ln_colsum) {   // LayerNorm-folded 1x1 GEMM
    launch_ring<BM, BN, WGM, WGN, true>(a, staging == 1 ? 0 : staging, s);
    return;
  }
  if (trans) {
    if (staging == 1) launch_variant<BM, BN, WGM, WGN, true, false, 2>(a, s);
    else if (staging >= 2) launch_variant<BM, BN, WGM, WGN, true, true, 3>(a, s);
    else launch_variant<BM, BN, WGM, WGN, true, true, 2>(a, s);
  } else {
    if (staging == 1) launch_variant<BM, BN, WGM, WGN, false, false, 2>(a, s);
    else launch_ring<BM, BN, WGM, WGN, false>(a, staging, s);
  }
}

}  // namespace

bool conv_fast_path_ok(const ConvDesc& d) {
  const int c1 = d.x1 ? d.C1 : 0;
  if (d.C0 % BK != 0 || c1 % BK != 0) return false;
  if (d.N % 4 != 0) return false;
  if (d.out_mode == kOutGeglu && d.N % 64 != 0) return false;
  if (!(d.ksize == 1 || d.ksize == 3)) return false;
  if (!(d.up == 1 || d.up == 2)) return false;
  return true;
}

// Would the plan of this conv be the weight-streaming kernel (plan tile 9) if its fragment-major weight copy existed?  The ONE
// predicate UNet::conv_w allocates that copy by (a second, looser copy of this rule left ~430 MB of copies per handle
// that no kernel reads).
bool conv_plan_is_wstream(const ConvDesc& d0) {
  if (!conv_fast_path_ok(d0) || !wstream_shape_ok(d0)) return false;
  ConvDesc d = d0;
  if (!d.w_tiled) d.w_tiled = d.w;   // choose_plan only tests the pointer
  const IgemmArgs a = make_args(d);
  return choose_plan(d, a).tile == 9;
}

size_t conv_workspace_bytes(const ConvDesc& d) {
  if (!conv_fast_path_ok(d)) return 0;
  IgemmArgs a = make_args(d);
  Plan p = choose_plan(d, a);
  // SD_TUNE=1 (tools/tune_plans.py): room for any split-K candidate of the sweep
  static const bool tuning = getenv("SD_TUNE") != nullptr;
  const bool can_split = d.out_mode == kOutHalf && !d.ln_colsum && !d.out_t;
  int splits = (tuning && can_split) ? std::max(p.splitk, 16) : p.splitk;
  // the weight-streaming kernel always leaves slabs (one per NW input-channel slices; NW = 4 is the upper bound), and so does
  // any plan once a GroupNorm twin is attached to the op (d.n_twins: launch_conv then forces the slab path)
  bool slab = d.n_twins > 0;
  if (d.w_tiled && can_split && wstream_shape_ok(d)) {
    splits = std::max(splits, wstream_splits(d, 4));
    slab = true;
  }
  if (splits == 1 && slab) return (size_t)a.M * a.N * sizeof(float);
  return splits > 1 ? (size_t)splits * a.M * a.N * sizeof(float) : 0;   // upper bound (launch may use fewer splits)
}

// sd_tune_set_plan_table: replace the run-time table by the rows of `text` (tuned_convs.inc format, one per line) - how
// tools/tune_e2e.py tries one plan after another against the graph replay time of the whole step in ONE process
int conv_plan_table_set(const char* text) {
  std::vector<TunedConv>& t = runtime_table();
  t.clear();
  if (!text) return 0;
  const char* p = text;
  while (*p) {
    const char* e = strchr(p, '\n');
    std::string line(p, e ? (size_t)(e - p) : strlen(p));
    TunedConv r;
    if (parse_plan_row(line.c_str(), r)) t.push_back(r);
    if (!e) break;
    p = e + 1;
  }
  return (int)t.size();
}

void conv_tune_set_candidate(int tile, int staging, int splitk) {
  g_tune.tile = tile;
  g_tune.staging = staging;
  g_tune.splitk = splitk;
}

namespace {
// GroupNorm statistics from the epilogue: fills a.gn_* and returns the entries per (sample, group) the launch will write,
// or 0 when this launch cannot produce them (bm: rows per m-tile of an igemm tile, 0 for the 8x16-pixel halo tiles)
int setup_gn_stats(const ConvDesc& d, IgemmArgs& a, int bm) {
  a.gn_partial = nullptr;
  if (!d.gn_partial || d.gn_groups < 1 || a.splitk > 1 || a.slab || d.out_mode != kOutHalf || d.out_t || d.debug) return 0;
  if (a.N % d.gn_groups != 0 || a.N % 8 != 0) return 0;
  const int cpg = a.N / d.gn_groups;
  if (cpg > 64) return 0;                       // a group may span two 64-column n-tiles, not three
  int T;
  if (bm == 0) {
    T = a.tiles_x * a.tiles_y;
  } else {
    if (a.HoWo % bm != 0) return 0;             // every m-tile inside one sample
    T = a.HoWo / bm;
  }
  if (2 * T > kGnMaxSlabs) return 0;
  a.gn_partial = d.gn_partial;
  a.gn_G = d.gn_groups;
  a.gn_cpg = cpg;
  a.gn_T = T;
  return 2 * T;
}
}  // namespace

namespace {
// slab combine of a split-K / weight-streaming launch: plain, or - when the consumer is a GroupNorm over <= 256 pixels per sample -
// with the GroupNorm statistics of the result (returns the entries per (sample, group) it wrote, else 0)
int launch_slab_combine(const ConvDesc& d, IgemmArgs& a, hipStream_t s) {
  static const int stats_mode = tune_env_int("SD_REDUCE_STATS", 0);   // measured: the step loses 0.25 ms with it (LAB_NOTES.md r5) - off unless asked for
  a.gn_partial = nullptr;
  if (stats_mode != 0 && d.gn_partial && d.gn_groups >= 1 && d.gn_groups <= 256 && a.N % d.gn_groups == 0 && (a.N / d.gn_groups) % 4 == 0 &&
      a.HoWo <= 256 && d.out_mode == kOutHalf && !d.out_t) {
    int ppb = std::max(1, a.B * a.HoWo / 512);
    const int slabs = cdiv(a.HoWo, ppb);
    const size_t lds = (size_t)ppb * (a.N / 4) * 2 * sizeof(float);
    if (slabs <= kGnMaxSlabs && lds <= 64 * 1024) {
      a.gn_partial = d.gn_partial;
      a.gn_G = d.gn_groups;
      a.gn_cpg = a.N / d.gn_groups;
      a.gn_T = slabs;
      hipLaunchKernelGGL(splitk_reduce_stats_kernel, dim3(slabs, a.B), dim3(256), lds, s, a, ppb);
      return slabs;
    }
  }
  const size_t total4 = (size_t)a.M * a.N / 4;
  const int blocks = (int)std::min<size_t>((total4 + 255) / 256, 2048);
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, s, a);
  return 0;
}
}  // namespace

namespace {
// tile order (IgemmArgs::n_fast) from an estimate of the bytes each order pulls through the fabric into the 8 XCD L2s:
//   m fastest: every weight panel once; the activations once per XCD when they fit an L2, else once per n-tile
//   n fastest: the activations once; the weights once per XCD when they fit an L2, else once per m-tile
// (SD_TILE_ORDER=2: round 3's first rule, activations x (n-tiles - 1) > 7 x weights - same step time at batch 2, but it
// sent the 1280 -> 1280 GEMMs of the 16x16 level n-fast: 27 MB of fabric reads per launch for 4.6 MB of operands)
void choose_tile_order(IgemmArgs& a, int tile) {
  int bm, bn;
  tile_dims(tile, bm, bn);
  const bool halo_tile = tile == 7;
  const double nbn = (double)cdiv(a.N, bn);
  const double nbm = halo_tile ? (double)a.B * a.tiles_x * a.tiles_y : (double)cdiv(a.M, bm);
  const double a_bytes = 2.0 * a.B * a.Hi * a.Wi * a.Ctot, w_bytes = 2.0 * a.N * a.K;
  const double l2 = 3.5e6;   // what one 4-MB L2 keeps of an operand next to the other one's stream
  const double m_fast_cost = w_bytes + a_bytes * (a_bytes <= l2 ? std::min(8.0, nbn) : nbn);
  const double n_fast_cost = a_bytes + w_bytes * (w_bytes <= l2 ? std::min(8.0, nbm) : nbm);
  static const int forced = tune_env_int("SD_TILE_ORDER", -1);   // A/B switch: 0 / 1 / 2
  if (forced == 2) a.n_fast = nbn > 1 && a_bytes * (nbn - 1) > 7.0 * w_bytes;
  else a.n_fast = forced >= 0 ? (forced != 0) : (nbn > 1 && n_fast_cost < m_fast_cost);
}
}  // namespace

// GroupNorm(+SiLU) of the input applied in the loader of the 3x3 halo kernel (ConvDesc::gnf_* on a ksize-3 conv): shapes it takes
bool conv_gn_loader_ok(const ConvDesc& d) {
  const int c1 = d.x1 ? d.C1 : 0;
  const int ctot = d.C0 + c1;
  return d.ksize == 3 && d.stride == 1 && d.up == 1 && halo_ks_ok(d) && !d.ln_colsum && !d.out_t && d.out_mode == kOutHalf && !d.debug &&
         d.gnf_groups >= 1 && d.gnf_groups <= 32 && ctot % d.gnf_groups == 0 && ctot <= 2048 && d.Hi == d.Ho && d.Wi == d.Wo &&
         halo_gnl_lds_bytes(3, ctot) <= halo_gnl_lds_cap;
}

int launch_conv(const ConvDesc& d, const ConvWorkspace& ws, hipStream_t s) {
  SD_REQUIRE(conv_fast_path_ok(d), kInvalidArgument, "launch_conv: shape not MFMA-tileable (C0=%d C1=%d N=%d k=%d)",
             d.C0, d.C1, d.N, d.ksize);
  SD_REQUIRE(!d.ln_colsum || (d.ksize == 1 && !d.x1 && d.bias && d.out_mode != kOutHalfT), kInvalidArgument,
             "LayerNorm fold needs a 1x1 single-source GEMM with a (folded) bias");
  SD_REQUIRE(!d.out_t || (d.out_mode == kOutHalf && d.n_trans % 64 == 0 && d.n_trans < d.N && (d.Ho * d.Wo) % 8 == 0 &&
                          d.ldT % 8 == 0 && !d.res && !d.temb),
             kInvalidArgument, "fused q|k|v: n_trans %d N %d HoWo %d ldT %d", d.n_trans, d.N, d.Ho * d.Wo, d.ldT);
  SD_REQUIRE(!d.vt_perm || (d.out_t && (d.Ho * d.Wo) % 16 == 0), kInvalidArgument, "permuted V^T needs the fused q|k|v epilogue and HoWo %% 16 == 0");
  SD_REQUIRE(d.q_cols == 0 || (d.out_t && d.q_cols % 4 == 0 && d.q_cols <= d.n_trans), kInvalidArgument, "pre-scaled queries need the fused q|k|v epilogue (q_cols %d)", d.q_cols);
  {   // plan tile 10: the weight-stationary GEGLU kernel (wsgemm.hip) wherever its pre-tiled weights exist and nothing else was
      // asked for (SD_WSGEMM=0 with SD_TUNE: the tiled kernels, A/B; a tuner candidate in force also keeps it off)
    static const bool wsg_on = tune_env_int("SD_WSGEMM", 1) != 0;
    const bool forced = d.tile == 10;
    if (d.w_ws && wsgemm_shape_ok(d) && (forced || (wsg_on && wsgemm_wanted(d) && d.tile == 0 && d.splitk == 0 && d.staging == 0 && g_tune.tile == 0))) {
      launch_wsgemm(d, s);
      return 0;
    }
    SD_REQUIRE(!forced, kInvalidArgument, "plan tile 10 (wsgemm.hip) needs the pre-tiled weights and an eligible GEGLU shape");
  }
  {   // plan tile 11 on the library's own rule (SD_BVGEMM=0 with SD_TUNE: off, A/B)
    static const bool bv_on = tune_env_int("SD_BVGEMM", 1) != 0;
    if (bv_on && d.w_bv && d.tile == 0 && d.splitk == 0 && d.staging == 0 && g_tune.tile == 0 && bvgemm_wanted(d)) {
      launch_bvgemm(d, 0, s);
      return 0;
    }
  }
  if (d.tile == 11) {   // plan tile 11: weights global -> VGPR (bvgemm.hip); staging 1 - 4 force a variant (launch_bvgemm)
    SD_REQUIRE(d.w_bv && bvgemm_shape_ok(d), kInvalidArgument, "plan tile 11 (bvgemm.hip) needs the pre-tiled weights and an eligible 1x1 shape");
    launch_bvgemm(d, d.staging, s);
    return 0;
  }
  IgemmArgs a = make_args(d);
  Plan p = choose_plan(d, a);
  bool halo = p.tile == 7;
  const bool twins = d.n_twins > 0;
  SD_REQUIRE(!twins || (d.out_mode == kOutHalf && !d.ln_colsum && !d.out_t && !d.debug && reduce_twin_ok(a.HoWo, a.N, d.n_twins, d.twin)),
             kInvalidArgument, "GroupNorm twins need a plain fp16 output and whole (sample, group) slices (HoWo=%d N=%d)", a.HoWo, a.N);
  if (d.gnf_partial && d.ksize == 3) {   // GroupNorm(+SiLU) applied in the halo loader of the K-split 3x3 kernel (GNL)
    SD_REQUIRE(conv_gn_loader_ok(d) && d.gnf_silu == 1 && d.gnf_gamma && d.gnf_beta && d.gnf_entries >= 1 && d.gnf_entries <= 128 && !twins, kInvalidArgument,
               "GroupNorm in the conv loader: Ctot=%d groups=%d entries=%d up=%d", a.Ctot, d.gnf_groups, d.gnf_entries, d.up);
    p.tile = 7;
    if (p.staging != 2 && p.staging != 3) p.staging = 3;
    p.splitk = 1;
  } else if (d.gnf_partial) {   // GroupNorm folded into this 1x1 GEMM: gemm_pipe_kernel's 64 x 64 tile (the only GNF instantiation)
    SD_REQUIRE(d.ksize == 1 && d.stride == 1 && !d.x1 && d.out_mode == kOutHalf && !d.ln_colsum && !d.out_t && !twins && gemm_pipe_ok(a) &&
                   d.gnf_gamma && d.gnf_beta && d.gnf_groups >= 1 && d.gnf_groups <= 32 && a.K % d.gnf_groups == 0 && a.K <= 2048 &&
                   d.gnf_entries >= 1 && d.gnf_entries <= 128 && a.HoWo % 64 == 0,
               kInvalidArgument, "GroupNorm fold: K=%d groups=%d entries=%d HoWo=%d", a.K, d.gnf_groups, d.gnf_entries, a.HoWo);
    p.tile = 3;
    if (p.staging != 6 && p.staging != 7 && p.staging != 8) p.staging = 6;
    p.splitk = 1;
  }
  halo = p.tile == 7;
  if (p.tile == 9) {
    // weight-streaming kernel: slabs, then the group-organised combine (with the consumer's GroupNorm twins) or the plain one
    const int nw = p.staging == 4 ? 4 : 8;
    const int S = wstream_splits(d, nw);
    const size_t need = (size_t)S * a.M * a.N * sizeof(float);
    SD_REQUIRE(ws.partial && ws.partial_bytes >= need, kInternal, "wstream workspace too small (%zu < %zu)", ws.partial_bytes, need);
    static const bool log_ws = tune_env_set("SD_LOG_CONVS");
    if (log_ws)
      fprintf(stderr, "[sd conv] k%d up%d C0=%d C1=%d M=%d N=%d K=%d tile=9 nw=%d slabs=%d twins=%d\n", a.ksize, a.up, a.C0, a.C1, a.M, a.N,
              a.K, nw, S, d.n_twins);
    launch_wstream(d, ws.partial, nw, s);
    a.partial = ws.partial;
    a.splitk = S;
    a.slab = 1;
    int entries = 0;
    if (twins) launch_reduce_twin(a.partial, S, a.M, a.N, a.HoWo, a.bias, a.temb, a.temb_stride, a.res, a.out, d.n_twins, d.twin, s);
    else entries = launch_slab_combine(d, a, s);
    SD_HIP(hipGetLastError());
    return entries;
  }
  if (halo) {
    const int nch = a.Ctot / BK;
    a.nk_per_split = cdiv(nch, p.splitk);
    a.splitk = cdiv(nch, a.nk_per_split);
  } else {
    a.splitk = p.splitk;
    a.nk_per_split = cdiv(a.nk_total, p.splitk);
    a.splitk = cdiv(a.nk_total, a.nk_per_split);   // no empty splits
  }
  a.slab = (a.splitk > 1 || twins) ? 1 : 0;
  if (a.slab) {
    size_t need = (size_t)a.splitk * a.M * a.N * sizeof(float);
    SD_REQUIRE(ws.partial && ws.partial_bytes >= need, kInternal, "split-K workspace too small (%zu < %zu)",
               ws.partial_bytes, need);
    a.partial = ws.partial;
  }
  const bool trans = d.out_mode == kOutHalfT;
  const int st = p.staging;
  choose_tile_order(a, p.tile);
  static const bool log_plans = tune_env_set("SD_LOG_CONVS");
  if (log_plans)
    fprintf(stderr, "[sd conv] k%d s%d up%d C0=%d C1=%d M=%d N=%d K=%d mode=%d tile=%d splitk=%d\n", a.ksize, a.stride, a.up,
            a.C0, a.C1, a.M, a.N, a.K, d.out_mode, p.tile, a.splitk);
  int gn_entries = 0;
  if (halo) {
    gn_entries = setup_gn_stats(d, a, 0);
    launch_halo_ks(a, a.splitk, st, s);
  } else if (d.debug && d.debug < 64) {   // ablation builds exist for two tiles only (tools/prof_conv.py)
    const bool ok = p.tile == 1 ? launch_debug_mode<128, 128>(a, d.debug, s) : launch_debug_mode<64, 64>(a, d.debug, s);
    SD_REQUIRE(ok && !trans, kInvalidArgument, "no ablation kernel for debug mode %d", d.debug);
    return 0;
  } else {
    if (!trans) {
      int bm, bn;
      tile_dims(p.tile >= 1 && p.tile <= 3 ? p.tile : 4, bm, bn);
      gn_entries = setup_gn_stats(d, a, bm);
    }
    switch (p.tile) {
      case 1: launch_tile<128, 128, 2, 2>(a, trans, st, s); break;
      case 2: launch_tile<128, 64, 2, 2>(a, trans, st, s); break;
      case 3: launch_tile<64, 64, 2, 2>(a, trans, st, s); break;
      default: launch_tile<64, 128, 2, 2>(a, trans, st, s); break;
    }
  }
  if (twins) {
    launch_reduce_twin(a.partial, a.splitk, a.M, a.N, a.HoWo, a.bias, a.temb, a.temb_stride, a.res, a.out, d.n_twins, d.twin, s);
  } else if (a.slab) {
    gn_entries = launch_slab_combine(d, a, s);
  }
  SD_HIP(hipGetLastError());
  return gn_entries;
}

// ---- GroupNorm + independent 1x1 GEMM in one launch (gn_*_side_kernel) ----
bool gn_side_gemm_ok(const ConvDesc& d) {
  if (!(d.ksize == 1 && d.stride == 1 && d.up == 1 && d.out_mode == kOutHalf && !d.ln_colsum && !d.out_t && d.n_twins == 0 && !d.gnf_partial &&
        !d.gn_partial && !d.temb && !d.debug && conv_fast_path_ok(d)))
    return false;
  const IgemmArgs a = make_args(d);
  return gemm_pipe_ok(a) && a.M >= 256;   // (M = 128: the 8x8 level's shortcut GEMMs are split-K weight streams, a 40-workgroup side GEMM loses)
}

namespace {
constexpr int kSideD = 4;   // ring depth of the side GEMM: 64 KB + constants, the plan most shortcut shapes have stand-alone
IgemmArgs side_args(const ConvDesc& d) {
  SD_REQUIRE(gn_side_gemm_ok(d), kInvalidArgument, "side GEMM: not a plain 1x1 GEMM of the pipelined kernel (C0=%d C1=%d N=%d)", d.C0, d.C1, d.N);
  IgemmArgs a = make_args(d);
  a.splitk = 1;
  a.slab = 0;
  choose_tile_order(a, 3);
  return a;
}
constexpr size_t kSideLds = (size_t)kSideD * (64 + 64) * BK * sizeof(half_t) + 2 * 64 * sizeof(float);
}  // namespace

void launch_gn_apply_side(const half_t* x0, int C0, const half_t* x1, int C1, const float* partial, int entries, const float* gamma,
                          const float* beta, half_t* y, int B, int HW, int G, float eps, int silu, int slabs, int ppb, const ConvDesc& side,
                          hipStream_t s) {
  IgemmArgs a = side_args(side);
  GnSideArgs g{x0, x1, C0, x1 ? C1 : 0, partial, entries, gamma, beta, y, HW, G, eps, silu, ppb, slabs, B, 0};
  g.n_gn_pad = (slabs * B + 7) / 8 * 8;
  auto k = gn_apply_side_kernel<kSideD>;
  static DynLdsOnce once;
  once.set(k, kSideLds);
  hipLaunchKernelGGL(k, dim3(g.n_gn_pad + cdiv(a.M, 64) * cdiv(a.N, 64)), dim3(256), kSideLds, s, g, a);
  SD_HIP(hipGetLastError());
}

void launch_gn_fused_side(int vw, const half_t* x0, int C0, const half_t* x1, int C1, const float* gamma, const float* beta, half_t* y, int B,
                          int HW, int G, float eps, int silu, const ConvDesc& side, hipStream_t s) {
  IgemmArgs a = side_args(side);
  GnSideArgs g{x0, x1, C0, x1 ? C1 : 0, nullptr, 0, gamma, beta, y, HW, G, eps, silu, 0, G, B, 0};
  g.n_gn_pad = (G * B + 7) / 8 * 8;
  const dim3 grid(g.n_gn_pad + cdiv(a.M, 64) * cdiv(a.N, 64));
  static DynLdsOnce o8, o4, o2;
  if (vw == 8) {
    auto k = gn_fused_side_kernel<8, kSideD>;
    o8.set(k, kSideLds);
    hipLaunchKernelGGL(k, grid, dim3(256), kSideLds, s, g, a);
  } else if (vw == 4) {
    auto k = gn_fused_side_kernel<4, kSideD>;
    o4.set(k, kSideLds);
    hipLaunchKernelGGL(k, grid, dim3(256), kSideLds, s, g, a);
  } else {
    auto k = gn_fused_side_kernel<2, kSideD>;
    o2.set(k, kSideLds);
    hipLaunchKernelGGL(k, grid, dim3(256), kSideLds, s, g, a);
  }
  SD_HIP(hipGetLastError());
}

int launch_conv_generic(const ConvDesc& d, int act_silu_out, hipStream_t s) {
  IgemmArgs a = make_args(d);
  if (a.ksize == 3 && a.stride == 1 && a.up == 1 && a.Ctot == 4 && !d.x1 && d.pad < 0 && d.out_mode == kOutHalf && a.N % 8 == 0 &&
      !d.temb && !act_silu_out) {
    // (+ room for the GroupNorm statistics scratch of the shared tile epilogue behind the staged 128 x 64 tile)
    const size_t lds = std::max((size_t)(128 + 64) * BK * 2 + 2 * 64 * sizeof(float),
                                (size_t)128 * (64 + 8) * 2 + 16 + (kGnScratchFloats + 2 * 64) * sizeof(float));
    a.splitk = 1;
    const int gn_entries = setup_gn_stats(d, a, 128);
    hipLaunchKernelGGL(conv3x3_cin4_kernel, dim3(cdiv(a.M, 128) * cdiv(a.N, 64)), dim3(256), lds, s, a);
    SD_HIP(hipGetLastError());
    return gn_entries;
  }
  if (a.K <= SC_KMAX && d.out_mode == kOutHalf && a.N >= 64) {
    hipLaunchKernelGGL(conv_small_cin_kernel, dim3(cdiv(a.M, SC_PIX)), dim3(256), 0, s, a, act_silu_out);
    SD_HIP(hipGetLastError());
    return 0;
  }
  SD_REQUIRE(d.out_mode != kOutGeglu || d.N % 64 == 0, kUnsupported, "generic GEGLU needs N %% 64 == 0 (N=%d)", d.N);
  size_t total = (size_t)a.M * a.N;
  int blocks = (int)std::min<size_t>((total + 255) / 256, 65535);
  hipLaunchKernelGGL(conv_generic_kernel, dim3(blocks), dim3(256), 0, s, a, act_silu_out);
  SD_HIP(hipGetLastError());
  return 0;
}

void launch_conv_small_n(const ConvDesc& d, float* out_nchw_f32, hipStream_t s) {
  IgemmArgs a = make_args(d);
  SD_REQUIRE(a.N <= 8 && a.Ctot % 8 == 0 && a.C0 % 8 == 0, kInvalidArgument, "conv_small_n: N=%d Ctot=%d", a.N,
             a.Ctot);
  // 3x3 / stride 1 / N <= 4: four pixels of a row per lane group, every load in flight at once (conv3x3_small_n_rows_kernel);
  // SD_CONV_OUT_ROWS=0 (with SD_TUNE) keeps the one-wave-per-pixel kernel: A/B
  static const int rows_mode = tune_env_int("SD_CONV_OUT_ROWS", 1);
  const int chunks = a.Ctot / 8;
  const int lpc = chunks <= 16 ? 16 : (chunks <= 32 ? 32 : 64);
  if (rows_mode != 0 && a.ksize == 3 && a.stride == 1 && a.up == 1 && a.pad == 1 && !d.x1 && a.N <= 4 && chunks <= 64 &&
      a.Wo % 4 == 0 && a.Hi == a.Ho && a.Wi == a.Wo) {
    const int pg = 64 / lpc;
    const int groups = a.B * a.Ho * (a.Wo / 4);
    const dim3 grid(cdiv(cdiv(groups, pg), 4));
    if (lpc == 16) hipLaunchKernelGGL(conv3x3_small_n_rows_kernel<16>, grid, dim3(256), 0, s, a, out_nchw_f32);
    else if (lpc == 32) hipLaunchKernelGGL(conv3x3_small_n_rows_kernel<32>, grid, dim3(256), 0, s, a, out_nchw_f32);
    else hipLaunchKernelGGL(conv3x3_small_n_rows_kernel<64>, grid, dim3(256), 0, s, a, out_nchw_f32);
    SD_HIP(hipGetLastError());
    return;
  }
  hipLaunchKernelGGL(conv_small_n_kernel<8>, dim3(cdiv(a.M, 4)), dim3(256), 0, s, a, out_nchw_f32);
  SD_HIP(hipGetLastError());
}

}  // namespace sd
