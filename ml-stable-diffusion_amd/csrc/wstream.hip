// Small-M layers (the 8x8 / 16x16 levels at CFG batch 2: unet.py:731-795 mid block, :282-350 / :151-225 the deepest down / up
// blocks): a convolution there is a WEIGHT STREAM - 29.5 MB of fp16 weights for 3.8 GFLOP at 1280 -> 1280 @8x8 - and what the
// tiled kernels of igemm.hip expose is their serial K loop (36 barrier-separated steps per workgroup), not HBM:
// 21.8 us against 8.6-9.4 us for the bare stream (profiles/r04_ubench_weight_stream_cold.txt).
//
//   wstream_kernel      - no K loop at all.  A wave owns ONE 32-output-channel strip x ONE 32-input-channel slice x all taps
//                         of a 128-pixel block: it requests its whole weight slice (18 x 1 KB for a 3x3 conv) straight into
//                         VGPRs in MFMA fragment order from PRE-TILED weights (one fully coalesced 16-B-per-lane load per
//                         fragment, nothing through LDS, all of them in flight at once), copies its private 32-channel slice of
//                         the input halo into its own LDS region (no workgroup barrier: only this wave reads it) and issues the
//                         MFMAs as the fragments land (the compiler's in-order vmcnt waits).  The NW waves of a workgroup
//                         are NW K-slices of the same strip and are summed through LDS once; K-splits across workgroups leave
//                         fp32 slabs like every split-K kernel of the library.
//   reduce_twin_kernel  - the slab combine, organised per (sample, GroupNorm group): it adds bias / timestep embedding /
//                         residual, stores the fp16 tensor AND - because a workgroup holds whole (sample, group) slices in
//                         registers - the GroupNorm(+SiLU) of it that the consuming resnet / SpatialTransformer asks for
//                         (unet.py:430-451, :472-481, :528-531), also when the consumer normalises a channel CONCAT whose group
//                         boundaries fall on the tensor boundary (torch.cat of :213-216): the "twin" goes to its column range
//                         of the concatenated operand.  The separate GroupNorm launch and its round trip disappear.
#include "kernels.h"

namespace sd {

namespace {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

struct TwinArgs {
  const float* partial;   // [S][M][N] fp32 slabs
  int S, M, N, HW;
  const float* bias;      // [N] or null
  const float* temb;      // [B][temb_stride] or null
  int temb_stride;
  const half_t* res;      // [M][N] or null
  half_t* out;            // [M][N]
  int cb;                 // channels per workgroup: the largest twin group
  int cmin;               // the smallest twin group (cb or cb / 2)
  int nt;
  GnTwin tw[2];
};

// grid (N / cb, B), NT threads.  Thread items: (pixel, 4-channel quad) pairs of the (sample, channel block) slice, kept in
// registers between the statistics and the apply pass (HW * cb / 4 <= NT * MAXI).  Fixed-order reductions only.
// The first version (256 threads, up to 20 items each, one item's slab loads behind the previous item's store) ran 10-30 us:
// 32-64 workgroups of four waves with ONE memory round trip per item.  Here every slab round is MAXI independent loads per
// thread on 16 waves, and nothing is stored before the last load has returned.
template <int NT, int MAXI, int SCH>
__global__ __launch_bounds__(NT) void reduce_twin_kernel(TwinArgs a) {
  constexpr int NWV = NT / 64;
  __shared__ float red[NWV][4];
  const int t = threadIdx.x, b = blockIdx.y, c0 = blockIdx.x * a.cb;
  const int Q = a.cb >> 2, items = a.HW * Q;
  const float* __restrict__ partial = a.partial;
  floatx4 v[MAXI];
  int p_[MAXI], n_[MAXI];
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    const int id = t + NT * i;
    const int p = id / Q, qd = id - p * Q;
    p_[i] = id < items ? p : -1;
    n_[i] = c0 + 4 * qd;
    v[i] = floatx4{0.f, 0.f, 0.f, 0.f};
  }
  // ONE memory round trip for everything this workgroup reads before the statistics: the first SCH slabs of every item, the
  // per-column constants and the residual are requested back to back (a round per slab and a round per operand kind
  // cost the first version five to seven dependent round trips: 10-32 us per launch)
  const size_t slab = (size_t)a.M * a.N;
  floatx4 ld[SCH][MAXI], cb_[MAXI], ct_[MAXI];
  half4 rr[MAXI];
#pragma unroll
  for (int z = 0; z < SCH; ++z)
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      ld[z][i] = floatx4{0.f, 0.f, 0.f, 0.f};
      if (p_[i] >= 0 && z < a.S) ld[z][i] = *reinterpret_cast<const floatx4*>(partial + z * slab + (size_t)(b * a.HW + p_[i]) * a.N + n_[i]);
    }
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    rr[i] = half4{0, 0, 0, 0};
    cb_[i] = ct_[i] = floatx4{0.f, 0.f, 0.f, 0.f};
    if (p_[i] >= 0) {
      if (a.bias) cb_[i] = *reinterpret_cast<const floatx4*>(a.bias + n_[i]);
      if (a.temb) ct_[i] = *reinterpret_cast<const floatx4*>(a.temb + (size_t)b * a.temb_stride + n_[i]);
      if (a.res) rr[i] = *reinterpret_cast<const half4*>(a.res + (size_t)(b * a.HW + p_[i]) * a.N + n_[i]);
    }
  }
#pragma unroll
  for (int z = 0; z < SCH; ++z)   // slice order: the sum does not depend on the chunking
#pragma unroll
    for (int i = 0; i < MAXI; ++i) v[i] += ld[z][i];
  for (int z = SCH; z < a.S; ++z) {   // (more slabs than one batch holds: a further round each)
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
      if (p_[i] >= 0) v[i] += *reinterpret_cast<const floatx4*>(partial + z * slab + (size_t)(b * a.HW + p_[i]) * a.N + n_[i]);
  }
#pragma unroll
  for (int i = 0; i < MAXI; ++i) v[i] += cb_[i] + ct_[i];
  float s0 = 0.f, q0 = 0.f, s1 = 0.f, q1 = 0.f;   // (sum, sumsq) of the two cmin-wide halves of the block
#pragma unroll
  for (int i = 0; i < MAXI; ++i) {
    if (p_[i] >= 0) {
      const half4 o = {(half_t)(v[i][0] + (float)rr[i][0]), (half_t)(v[i][1] + (float)rr[i][1]), (half_t)(v[i][2] + (float)rr[i][2]),
                       (half_t)(v[i][3] + (float)rr[i][3])};
      *reinterpret_cast<half4*>(a.out + (size_t)(b * a.HW + p_[i]) * a.N + n_[i]) = o;
      // statistics of what a separate GroupNorm launch would read: the fp16-rounded tensor
      const floatx4 f = {(float)o[0], (float)o[1], (float)o[2], (float)o[3]};
      v[i] = f;
      const float fs = (f[0] + f[1]) + (f[2] + f[3]);
      const float fq = fmaf(f[0], f[0], fmaf(f[1], f[1], fmaf(f[2], f[2], f[3] * f[3])));
      if (n_[i] - c0 < a.cmin) {
        s0 += fs;
        q0 += fq;
      } else {
        s1 += fs;
        q1 += fq;
      }
    }
  }
  // the first twin's affine is requested before the barrier (one round trip less behind the statistics)
  constexpr bool HOIST = MAXI <= 5;   // (the 10-item form has no registers left for it)
  floatx4 g0[HOIST ? MAXI : 1], be0[HOIST ? MAXI : 1];
#pragma unroll
  for (int i = 0; i < (HOIST ? MAXI : 0); ++i)
    if (p_[i] >= 0) {
      g0[i] = *reinterpret_cast<const floatx4*>(a.tw[0].gamma + a.tw[0].c_off + n_[i]);
      be0[i] = *reinterpret_cast<const floatx4*>(a.tw[0].beta + a.tw[0].c_off + n_[i]);
    }
  s0 = wave_sum_f(s0);
  q0 = wave_sum_f(q0);
  s1 = wave_sum_f(s1);
  q1 = wave_sum_f(q1);
  if ((t & 63) == 0) {
    red[t >> 6][0] = s0;
    red[t >> 6][1] = q0;
    red[t >> 6][2] = s1;
    red[t >> 6][3] = q1;
  }
  __syncthreads();
  s0 = q0 = s1 = q1 = 0.f;
#pragma unroll
  for (int w = 0; w < NWV; ++w) {   // fixed order
    s0 += red[w][0];
    q0 += red[w][1];
    s1 += red[w][2];
    q1 += red[w][3];
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (k >= a.nt) break;
    const GnTwin tw = a.tw[k];
    const bool whole = tw.cpg == a.cb && a.cmin != a.cb;   // this twin's group is the whole block, the statistics are kept per half
    const float inv_n = 1.0f / ((float)tw.cpg * (float)a.HW);
    const float sa = whole ? s0 + s1 : s0, qa = whole ? q0 + q1 : q0;
    const float mean0 = sa * inv_n;
    const float rstd0 = rsqrtf(fmaxf(qa * inv_n - mean0 * mean0, 0.f) + tw.eps);
    const float sb = whole ? sa : s1, qb = whole ? qa : q1;
    const float mean1 = sb * inv_n;
    const float rstd1 = rsqrtf(fmaxf(qb * inv_n - mean1 * mean1, 0.f) + tw.eps);
    floatx4 g[MAXI], be[MAXI];
#pragma unroll
    for (int i = 0; i < MAXI; ++i)
      if (p_[i] >= 0) {
        g[i] = (HOIST && k == 0) ? g0[HOIST ? i : 0] : *reinterpret_cast<const floatx4*>(tw.gamma + tw.c_off + n_[i]);
        be[i] = (HOIST && k == 0) ? be0[HOIST ? i : 0] : *reinterpret_cast<const floatx4*>(tw.beta + tw.c_off + n_[i]);
      }
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
      if (p_[i] >= 0) {
        const bool second = n_[i] - c0 >= a.cmin;
        const float mean = second ? mean1 : mean0, rstd = second ? rstd1 : rstd0;
        half4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float sc = rstd * g[i][e];
          float y = fmaf(v[i][e], sc, be[i][e] - mean * sc);
          if (tw.silu) y = y * __builtin_amdgcn_rcpf(1.0f + __expf(-y));
          o[e] = (half_t)y;
        }
        *reinterpret_cast<half4*>(tw.y + (size_t)(b * a.HW + p_[i]) * tw.ld + tw.c_off + n_[i]) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
struct WsArgs {
  const half_t* x0;
  const half_t* x1;
  const half_t* wt;   // pre-tiled weights: [N / 32][nslices][TAPS][2][64 lanes][8 halves]
  float* partial;     // [S][M][N]
  int C0, C1, nslices;
  int B, Hi, Wi, H, W, ush;   // source image, output (= upsampled source) image, log2 of the nearest upsample factor
  int M, N;
};

constexpr int WS_REGION = 16384;   // per-wave LDS: the 32-channel halo slice (<= 200 pixels x 80 B), later its 128 x 32 fp32 tile
constexpr int WS_ROWB = 80;        // 64 B of channels + 16 B pad: the 16 lanes of a ds_read_b128 group hit distinct banks

// NW waves; TAPS 9 (3x3, stride 1, pad 1) or 1 (1x1); WW = image width of the 3x3 form (8: a 128-pixel block is two 8-row
// sub-tiles, 16: one), ignored for TAPS == 1.  grid (N / 32, ceil(nslices / NW), ceil(M / 128)).
template <int NW, int TAPS, int WW>
__global__ __launch_bounds__(NW * 64) void wstream_kernel(WsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int strip = blockIdx.x, split = blockIdx.y, zb = blockIdx.z;
  const int slice = split * NW + wave;
  const bool live = slice < a.nslices;   // wave-uniform
  char* const region = smem + wave * WS_REGION;

  constexpr int NSUB = TAPS == 9 ? 16 / WW : 1;          // 8-row sub-tiles per block
  constexpr int HW_ = WW + 2, HPS = 10 * HW_;            // halo row length, halo pixels per sub-tile
  constexpr int HP = TAPS == 9 ? NSUB * HPS : 128;       // pixels staged per wave
  constexpr int NL = (HP + 15) / 16;                     // 16-pixel load groups
  constexpr int NF = TAPS * 2;                           // weight fragments per wave

  floatx16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  if (live) {
    // ---- activations first (they are waited for first; loads of one wave return in order) ----
    const int c = slice * 32;
    const bool second = c >= a.C0;                       // wave-uniform: the skip-concat's second source
    const half_t* xs = second ? a.x1 : a.x0;
    const int Cs = second ? a.C1 : a.C0, coff = second ? c - a.C0 : c;
    const int piece = lane & 3;
    half8 xa[NL];
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int px = i * 16 + (lane >> 2);
      bool ok;
      size_t pix;
      if constexpr (TAPS == 9) {
        const int sub = px / HPS, r = px - sub * HPS;
        const int hy = r / HW_, hx = r - hy * HW_;
        const int sg = zb * NSUB + sub, tpi = a.H >> 3;  // global sub-tile, sub-tiles per image
        const int b = sg / tpi, y0 = (sg - b * tpi) * 8;
        const int iy = y0 + hy - 1, ix = hx - 1;
        ok = px < HP && b < a.B && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        pix = ((size_t)b * a.Hi + (iy >> a.ush)) * a.Wi + (ix >> a.ush);   // unet.py:498-500 nearest upsample
      } else {
        const int m = zb * 128 + px;
        ok = m < a.M;
        pix = (size_t)m;
      }
      const half8 z = {0, 0, 0, 0, 0, 0, 0, 0};
      xa[i] = ok ? *reinterpret_cast<const half8*>(xs + pix * Cs + coff + piece * 8) : z;
    }
    // ---- the whole weight slice of this wave: NF fully coalesced 1-KB fragments, all in flight ----
    const half_t* wp = a.wt + (((size_t)strip * a.nslices + slice) * NF * 64 + lane) * 8;
    half8 bf[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) bf[j] = __builtin_nontemporal_load(reinterpret_cast<const half8*>(wp + (size_t)j * 512));
    // ---- halo slice -> this wave's LDS region (only this wave reads it: no barrier) ----
#pragma unroll
    for (int i = 0; i < NL; ++i) {
      const int px = i * 16 + (lane >> 2);
      if (px < HP) *reinterpret_cast<half8*>(region + px * WS_ROWB + piece * 16) = xa[i];
    }
    // ---- fragments: lane (m = lane & 31, hi = lane >> 5) of pixel tile i reads 16 B of its pixel's row under the tap shift ----
    int hb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ml = i * 32 + (lane & 31);
      if constexpr (TAPS == 9) {
        const int sub = ml / (8 * WW), rem = ml - sub * (8 * WW);
        const int y = rem / WW, x = rem - y * WW;
        hb[i] = (sub * HPS + y * HW_ + x) * WS_ROWB + (lane >> 5) * 16;
      } else {
        hb[i] = ml * WS_ROWB + (lane >> 5) * 16;
      }
    }
    // fragments of tap t+1 are read while the MFMAs of tap t issue (register double buffer; the first version read two
    // fragments, waited, issued two MFMAs: the LDS latency was exposed 36 times per wave)
    auto read_tap = [&](half8 (&xf)[8], int tap) {
      const int toff = TAPS == 9 ? ((tap / 3) * HW_ + (tap % 3)) * WS_ROWB : 0;
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[h * 4 + i] = *reinterpret_cast<const half8*>(region + hb[i] + toff + h * 32);
    };
    half8 xfa[8], xfb[8];
    read_tap(xfa, 0);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      half8(&cur)[8] = (tap & 1) ? xfb : xfa;
      half8(&nxt)[8] = (tap & 1) ? xfa : xfb;
      if (tap + 1 < TAPS) read_tap(nxt, tap + 1);
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[tap * 2 + h], cur[h * 4 + i], acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- sum the NW K-slices through LDS (each wave's own region, its fragment reads are behind it) ----
  {
    floatx4* rg = reinterpret_cast<floatx4*>(region);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int q = 0; q < 4; ++q) rg[(i * 4 + q) * 64 + lane] = floatx4{acc[i][4 * q], acc[i][4 * q + 1], acc[i][4 * q + 2], acc[i][4 * q + 3]};
  }
  __syncthreads();
  // The tile is 128 rows x 128 B of fp32: thread -> (row, 16-B piece) so that eight lanes write one whole 128-B line (the
  // accumulator order - 32 rows x 16 B per store instruction - made every slab store a partial-line write).
  // acc[i][4q .. 4q+3] of lane l2: n = 8 q + 4 (l2 >> 5) + {0..3}, m = 32 i + (l2 & 31)   (mfma(weights, activations))
  constexpr int PER = 1024 / (NW * 64);   // floatx4 items per thread
#pragma unroll
  for (int it = 0; it < PER; ++it) {
    const int e = it * (NW * 64) + tid;               // row-major: e = ml * 8 + piece
    const int ml = e >> 3, pc = e & 7;
    const int id = (((ml >> 5) * 4 + (pc >> 1)) * 64) + (pc & 1) * 32 + (ml & 31);
    floatx4 s = reinterpret_cast<const floatx4*>(smem)[id];
#pragma unroll
    for (int w = 1; w < NW; ++w) s += reinterpret_cast<const floatx4*>(smem + w * WS_REGION)[id];
    const int m = zb * 128 + ml, n = strip * 32 + 4 * pc;
    if (m < a.M) out_store(reinterpret_cast<floatx4*>(a.partial + ((size_t)split * a.M + m) * a.N + n), s);
  }
}

// [N][taps][Ctot] -> the fragment-major layout above; one thread per 16-byte piece
__global__ __launch_bounds__(256) void wstream_retile_kernel(const half_t* __restrict__ w, half_t* __restrict__ wt, int N, int Ctot, int taps) {
  const int nslices = Ctot / 32;
  const size_t total = (size_t)(N / 32) * nslices * taps * 2 * 64;
  for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    size_t r = idx >> 6;
    const int h = (int)(r & 1);
    r >>= 1;
    const int tap = (int)(r % taps);
    r /= taps;
    const int slice = (int)(r % nslices);
    const int strip = (int)(r / nslices);
    const int n = strip * 32 + (lane & 31), c = slice * 32 + h * 16 + (lane >> 5) * 8;
    *reinterpret_cast<half8*>(wt + idx * 8) = *reinterpret_cast<const half8*>(w + ((size_t)n * taps + tap) * Ctot + c);
  }
}

template <int NW, int TAPS, int WW>
void launch_ws(const WsArgs& a, int splits, hipStream_t s) {
  auto k = wstream_kernel<NW, TAPS, WW>;
  const size_t lds = (size_t)NW * WS_REGION;
  static DynLdsOnce once;
  once.set(k, lds);
  hipLaunchKernelGGL(k, dim3(a.N / 32, splits, cdiv(a.M, 128)), dim3(NW * 64), lds, s, a);
}

}  // namespace

bool wstream_shape_ok(const ConvDesc& d) {
  const int c1 = d.x1 ? d.C1 : 0;
  if (d.out_mode != kOutHalf || d.ln_colsum || d.out_t || d.stride != 1 || d.pad >= 0) return false;
  if (d.N % 32 != 0 || d.C0 % 32 != 0 || c1 % 32 != 0 || d.C0 < 32) return false;
  if (d.ksize == 3) return (d.Wo == 8 || d.Wo == 16) && d.Ho % 8 == 0 && (d.up == 1 || d.up == 2) && d.Ho == d.Hi * d.up && d.Wo == d.Wi * d.up;
  return d.ksize == 1 && d.up == 1;
}

int wstream_splits(const ConvDesc& d, int nw) {
  const int nslices = (d.C0 + (d.x1 ? d.C1 : 0)) / 32;
  return cdiv(nslices, nw);
}

size_t wstream_tiled_halves(int N, int Ctot, int ksize) { return (size_t)N * Ctot * ksize * ksize; }

void launch_wstream_retile(const half_t* w, half_t* wt, int N, int Ctot, int ksize, hipStream_t s) {
  SD_REQUIRE(N % 32 == 0 && Ctot % 32 == 0 && (ksize == 1 || ksize == 3), kInvalidArgument, "wstream retile: N=%d Ctot=%d k=%d", N, Ctot, ksize);
  const size_t total = (size_t)N * Ctot * ksize * ksize / 8;
  hipLaunchKernelGGL(wstream_retile_kernel, dim3((unsigned)std::min<size_t>((total + 255) / 256, 4096)), dim3(256), 0, s, w, wt, N, Ctot, ksize * ksize);
  SD_HIP(hipGetLastError());
}

// slabs [splits][M][N] of the conv into `partial`; nw = waves (K slices) per workgroup, 4 or 8.  Returns the slab count.
int launch_wstream(const ConvDesc& d, float* partial, int nw, hipStream_t s) {
  SD_REQUIRE(wstream_shape_ok(d) && d.w_tiled, kInvalidArgument, "wstream: shape not eligible (k=%d C0=%d C1=%d N=%d %dx%d)", d.ksize, d.C0,
             d.C1, d.N, d.Ho, d.Wo);
  WsArgs a{};
  a.x0 = d.x0;
  a.x1 = d.x1;
  a.wt = d.w_tiled;
  a.partial = partial;
  a.C0 = d.C0;
  a.C1 = d.x1 ? d.C1 : 0;
  a.nslices = (a.C0 + a.C1) / 32;
  a.B = d.B;
  a.Hi = d.Hi;
  a.Wi = d.Wi;
  a.H = d.Ho;
  a.W = d.Wo;
  a.ush = d.up >> 1;
  a.M = d.B * d.Ho * d.Wo;
  a.N = d.N;
  if (nw != 4) nw = 8;
  const int splits = cdiv(a.nslices, nw);
  if (d.ksize == 3) {
    if (d.Wo == 8) {
      if (nw == 4) launch_ws<4, 9, 8>(a, splits, s);
      else launch_ws<8, 9, 8>(a, splits, s);
    } else {
      if (nw == 4) launch_ws<4, 9, 16>(a, splits, s);
      else launch_ws<8, 9, 16>(a, splits, s);
    }
  } else {
    if (nw == 4) launch_ws<4, 1, 8>(a, splits, s);
    else launch_ws<8, 1, 8>(a, splits, s);
  }
  SD_HIP(hipGetLastError());
  return splits;
}

bool reduce_twin_ok(int HW, int N, int n_twins, const GnTwin* tw) {
  if (n_twins < 1 || n_twins > 2) return false;
  int cb = 0, cmin = 1 << 30;
  for (int k = 0; k < n_twins; ++k) {
    if (tw[k].cpg < 4 || tw[k].cpg % 4 != 0 || tw[k].c_off % tw[k].cpg != 0 || tw[k].ld % 4 != 0) return false;
    cb = std::max(cb, tw[k].cpg);
    cmin = std::min(cmin, tw[k].cpg);
  }
  if (!(cmin == cb || 2 * cmin == cb)) return false;
  if (N % cb != 0) return false;
  return (long)HW * (cb / 4) <= 512L * 10;
}

void launch_reduce_twin(const float* partial, int S, int M, int N, int HW, const float* bias, const float* temb, int temb_stride,
                        const half_t* res, half_t* out, int n_twins, const GnTwin* tw, hipStream_t s) {
  SD_REQUIRE(reduce_twin_ok(HW, N, n_twins, tw) && M % HW == 0 && S >= 1, kInvalidArgument, "reduce_twin: HW=%d N=%d twins=%d", HW, N, n_twins);
  TwinArgs a{};
  a.partial = partial;
  a.S = S;
  a.M = M;
  a.N = N;
  a.HW = HW;
  a.bias = bias;
  a.temb = temb;
  a.temb_stride = temb_stride;
  a.res = res;
  a.out = out;
  a.nt = n_twins;
  a.cb = 0;
  a.cmin = 1 << 30;
  for (int k = 0; k < n_twins; ++k) {
    a.tw[k] = tw[k];
    a.cb = std::max(a.cb, tw[k].cpg);
    a.cmin = std::min(a.cmin, tw[k].cpg);
  }
  const long items = (long)HW * (a.cb / 4);
  const dim3 grid(N / a.cb, M / HW);
  // (1024-thread workgroups cap a thread at 128 registers: the 5-item form spilled; 512 threads x 10 items keep the same number
  // of loads in flight per workgroup without scratch)
  if (items <= 256) hipLaunchKernelGGL((reduce_twin_kernel<256, 1, 8>), grid, dim3(256), 0, s, a);
  else if (items <= 1024) hipLaunchKernelGGL((reduce_twin_kernel<512, 2, 8>), grid, dim3(512), 0, s, a);
  else if (items <= 2560) hipLaunchKernelGGL((reduce_twin_kernel<512, 5, 5>), grid, dim3(512), 0, s, a);
  else hipLaunchKernelGGL((reduce_twin_kernel<512, 10, 2>), grid, dim3(512), 0, s, a);
  SD_HIP(hipGetLastError());
}

}  // namespace sd
