// Launch interfaces of the hand-written gfx950 kernels.  All activations are channels-last
// fp16: an image tensor is [B][H][W][C] == a token matrix [M = B*H*W][C]; the reference's BC1S
// sequence layout (layer_norm.py:25) is the same bytes read as [B][S][C].
#pragma once
#include "sd_common.h"

namespace sd {

// ---------------------------------------------------------------------------------------------
// K4/K5: implicit-GEMM convolution / 1x1 GEMM on MFMA (igemm.hip)
//   out[m][n] = epilogue( sum_k X[m][k] * W[n][k] ),  k = tap*(C0+C1) + c
// ---------------------------------------------------------------------------------------------
enum OutMode : int {
  kOutHalf = 0,        // half out[M][N]
  kOutHalfT = 1,       // half out[B][N][ldT]   (token-transposed: V^T for attention)
  kOutGeglu = 2,       // half out[M][N/2] = (v+bv) * gelu_erf(g+bg); W rows interleaved 32/32
};

// GroupNorm (+ SiLU) of a conv's OUTPUT written by the conv's own slab-combine pass (wstream.hip reduce_twin_kernel): the
// consumer is torch.nn.GroupNorm over this tensor (unet.py:430-451, :528-531) or over a channel concat in which this tensor
// occupies the group-aligned column range [c_off, c_off + N) (torch.cat of unet.py:213-216 followed by norm1).
struct GnTwin {
  half_t* y = nullptr;           // [M][ld]: the normalised copy goes to columns [c_off, c_off + N)
  int ld = 0, c_off = 0;
  int cpg = 0;                   // channels per group of the consuming GroupNorm
  const float* gamma = nullptr;  // the consumer's affine, indexed by ITS channel (c_off + n)
  const float* beta = nullptr;
  float eps = 1e-5f;
  int silu = 0;
};

struct ConvDesc {
  const half_t* x0 = nullptr;   // [B][Hi][Wi][C0]
  const half_t* x1 = nullptr;   // optional second source (channel concat), [B][Hi][Wi][C1]
  int C0 = 0, C1 = 0;
  const half_t* w = nullptr;    // [N][K] K-major, K = ksize*ksize*(C0+C1), tap-major
  const float* bias = nullptr;  // [N] or null
  const float* temb = nullptr;  // [B][temb_stride] broadcast over pixels, or null
  int temb_stride = 0;
  const half_t* res = nullptr;  // [M][N] residual, or null
  half_t* out = nullptr;
  int B = 1, Hi = 1, Wi = 1;    // source spatial size (before nearest x2 upsampling)
  int Ho = 1, Wo = 1;           // output spatial size
  int ksize = 1, stride = 1, up = 1;
  int pad = -1;                 // -1: ksize / 2 on every side; 0 with stride 2 = the VAE encoder's (0,1,0,1) padding
  int N = 0;
  int out_mode = kOutHalf;
  int ldT = 0;                  // kOutHalfT: padded token stride
  // tuning overrides (0 = heuristic)
  int tile = 0;                 // 1: 128x128  2: 128x64  3: 64x64 4: 64x128
  int splitk = 0;
  long long* prof = nullptr;    // debug bit 2: device buffer of 8 timestamps
  int debug = 0;                // ablation hooks for tools/microbench (results are garbage when != 0)
  int staging = 0;              // 0: LDS-DMA 2-stage, 1: HBM->VGPR->LDS (A/B reference), 2: LDS-DMA 3-stage ring
  // LayerNorm folded into this 1x1 GEMM (x is the UN-normalised input): w = W*gamma, bias = b + W.beta,
  // ln_colsum[n] = sum_k w[n][k]; the kernel derives mean / rstd of each row from its own A tiles
  const float* ln_colsum = nullptr;
  float ln_eps = 1e-5f;
  // fused q|k|v: output columns [n_trans, N) are written token-transposed to out_t [B][N-n_trans][ldT]
  // (attention's V^T), columns [0, n_trans) to out with row length n_trans.  Needs Ho*Wo % 8 == 0.
  half_t* out_t = nullptr;
  int n_trans = 0;
  int vt_perm = 0;              // out_t in the key order of AttnDesc::vt_perm (needs Ho*Wo % 16 == 0)
  // fused q|k|v for attention8: output columns [0, q_cols) - the queries - leave multiplied by q_scale (d^-0.5 * log2 e), applied
  // to the fp32 accumulator BEFORE the one rounding to fp16 (AttnDesc::q_prescaled).  0 columns = off.
  float q_scale = 1.f;
  int q_cols = 0;
  // GroupNorm statistics of the output from the conv's own epilogue (the consumer is a GroupNorm over exactly this
  // tensor, gn_groups groups): partial sums go to gn_partial [B][gn_groups][kGnMaxSlabs][2].  launch_conv returns how
  // many entries per (sample, group) it wrote - 0 when the chosen plan cannot (split-K, ragged tiles): the GroupNorm
  // then runs its own statistics pass.
  float* gn_partial = nullptr;
  int gn_groups = 0;
  // GroupNorm folded into this 1x1 GEMM (SpatialTransformer.norm -> proj_in, unet.py:528-531, :553-556): x0 is the UN-normalised
  // tensor, gnf_partial holds gnf_entries (sum, sumsq) partials per (sample, group) written by x0's producer
  // ([B][gnf_groups][kGnMaxSlabs][2]); the kernel turns them into per-channel scale / shift and applies them to its A fragments
  // (v_pk_fma_f16 between the LDS read and the MFMA).  Needs a single source, Ho * Wo a multiple of the M tile, C0 <= 2048.
  const float* gnf_partial = nullptr;
  const float* gnf_gamma = nullptr;
  const float* gnf_beta = nullptr;
  float gnf_eps = 1e-6f;
  int gnf_groups = 0, gnf_entries = 0;
  // The same fields on a 3x3 / stride-1 conv (round 5): GroupNorm + SiLU (gnf_silu must be 1) of the input applied in the halo LOADER of
  // conv3x3_halo_ks_kernel (ResnetBlock2D norm1 -> conv1, norm2 -> conv2: unet.py:472-481) - every wave normalises the halo pieces
  // it fetched, in LDS, once per workgroup; pixels outside the image stay zero (the conv pads the normalised tensor).  Shapes:
  // conv_gn_loader_ok; the concatenated channel count indexes gamma / beta / the partials.
  int gnf_silu = 0;
  // weights in the fragment-major layout of wstream.hip (launch_wstream_retile), or null: plan tile 9 needs them
  const half_t* w_tiled = nullptr;
  // weights in the fragment-major layout of wsgemm.hip (launch_wsgemm_retile), or null: the weight-stationary GEGLU kernel
  // (plan tile 10) needs them; launch_conv takes that kernel whenever they are there and no other plan was forced
  const half_t* w_ws = nullptr;
  // weights in the fragment-major layout of bvgemm.hip (launch_bvgemm_retile; plain or GEGLU row order by out_mode), or null:
  // plan tile 11 (weights global -> VGPR, large-M 1x1 GEMMs) needs them
  const half_t* w_bv = nullptr;
  // n_twins > 0: the output leaves through fp32 slabs and reduce_twin_kernel, which also writes the GroupNorm twins
  // (needs Ho * Wo <= 256 and reduce_twin_ok; launch_conv forces the slab path whatever the plan's split-K)
  GnTwin twin[2];
  int n_twins = 0;
};

constexpr int kGnMaxSlabs = 256;   // entries per (sample, group) of a GroupNorm partial buffer

struct ConvWorkspace {
  float* partial = nullptr;     // split-K slabs
  size_t partial_bytes = 0;
};

// Returns the workspace bytes this conv needs with its current heuristic (for planning).
size_t conv_workspace_bytes(const ConvDesc& d);
// returns the number of GroupNorm partial entries per (sample, group) written to d.gn_partial (0: none)
int launch_conv(const ConvDesc& d, const ConvWorkspace& ws, hipStream_t s);
bool conv_fast_path_ok(const ConvDesc& d);
bool conv_gn_loader_ok(const ConvDesc& d);   // gnf_* on a 3x3 conv (gnf_groups set): the halo kernel can normalise this input in its loader
// tuning hook: plan (tile 1-6, staging 0-5, splitk) forced on every conv that admits it; tile 0 = off
void conv_tune_set_candidate(int tile, int staging, int splitk);
int conv_plan_table_set(const char* text);   // rows of tuned_convs.inc format; returns the number of plans read

// wstream.hip: the small-M weight-streaming kernel (plan tile 9) and the group-organised slab combine
bool conv_plan_is_wstream(const ConvDesc& d);            // choose_plan would take plan tile 9 given the pre-tiled weights
bool wstream_shape_ok(const ConvDesc& d);                 // shape admits plan tile 9 (d.w_tiled not looked at)
int wstream_splits(const ConvDesc& d, int nw);            // slabs launch_wstream writes with nw waves per workgroup
size_t wstream_tiled_halves(int N, int Ctot, int ksize);
void launch_wstream_retile(const half_t* w, half_t* wt, int N, int Ctot, int ksize, hipStream_t s);
int launch_wstream(const ConvDesc& d, float* partial, int nw, hipStream_t s);
bool reduce_twin_ok(int HW, int N, int n_twins, const GnTwin* tw);
void launch_reduce_twin(const float* partial, int S, int M, int N, int HW, const float* bias, const float* temb, int temb_stride,
                        const half_t* res, half_t* out, int n_twins, const GnTwin* tw, hipStream_t s);

// wsgemm.hip (round 6): weight-stationary GEGLU projection of the 320-channel level (plan tile 10)
bool wsgemm_shape_ok(const ConvDesc& d);
bool wsgemm_wanted(const ConvDesc& d);                               // the library's rule for taking plan tile 10 on its own
size_t wsgemm_tiled_halves(int N);
void launch_wsgemm_retile(const half_t* w, half_t* wt, int N, bool geglu, hipStream_t s);
void launch_wsgemm(const ConvDesc& d, hipStream_t s);

// bvgemm.hip (round 6): 1x1 GEMM with the weights global -> VGPR, activations alone in LDS (plan tile 11)
bool bvgemm_shape_ok(const ConvDesc& d);
size_t bvgemm_tiled_halves(int N, int K);
void launch_bvgemm_retile(const half_t* w, half_t* wt, int N, int K, bool geglu, hipStream_t s);
void launch_bvgemm(const ConvDesc& d, int variant, hipStream_t s);   // variant 1-4 (bvgemm.hip), 0 = the library's choice
bool bvgemm_wanted(const ConvDesc& d);                                // the library's rule for taking plan tile 11 on its own

// calib.hip: box calibration for bench.py - out[0..6] = copy GB/s, dense MFMA TFLOP/s, us per launch of a 323-launch empty
// graph, us per launch of a 323-launch chain of short kernels on cold operands, us per launch of a 323-launch chain handing 8 MB
// over between the XCDs' L2s, ns per dependent load from HBM / from the caches
void run_calibration(int device, float* out);

// direct conv for tiny / odd shapes (any Cin, any N): fp32 accumulate, one thread per output
int launch_conv_generic(const ConvDesc& d, int act_silu_out, hipStream_t s);   // returns like launch_conv

// N <= 8 outputs, K % 8 == 0: one wavefront per output pixel.  out_nchw_f32: write float
// [B][N][Ho][Wo] (the UNet's noise_pred boundary), else half NHWC.
void launch_conv_small_n(const ConvDesc& d, float* out_nchw_f32, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// K1-K3: attention (attention.hip).  q [B][Sq][ldq] (head h at column h*d), k [B][Sk][ldk],
// vt [B][heads*d][ldv] (token-transposed), out [B][Sq][ldo].
// ---------------------------------------------------------------------------------------------
enum AttnImpl : int { kAttnOriginal = 0, kAttnSplitEinsum = 1, kAttnSplitEinsumV2 = 2 };

struct AttnDesc {
  const half_t* q = nullptr;
  const half_t* k = nullptr;
  const half_t* vt = nullptr;
  half_t* out = nullptr;
  int B = 1, heads = 1, d = 64, Sq = 0, Sk = 0;
  int ldq = 0, ldk = 0, ldv = 0, ldo = 0;
  int impl = kAttnOriginal;
  int variant = 0;   // A/B testing: 1 = never use the attention8.hip kernel
  // 1: every 16-key group of a vt row is stored [k0-3 | k8-11 | k4-7 | k12-15] - the order in which a lane of the key-major
  // P.V MFMA consumes keys, so its V^T fragment is ONE 16-byte LDS read (attention8.hip; written so by the fused q|k|v GEMM,
  // ConvDesc::vt_perm).  Only attention8 reads this layout; the general kernels need 0.
  int vt_perm = 0;
  // 1: q already carries d^-0.5 * log2(e) (multiplied into the fp32 accumulator of the producing GEMM, ConvDesc::q_scale): attention8
  // does not re-scale (and re-round) it.  Only with vt_perm.
  int q_prescaled = 0;
  // scratch for attention8's balanced form (attention8_sk_scratch says how much this problem wants): partial (m, l, O) results and
  // one arrival counter per query tile, ZEROED once by the owner (the kernel leaves them at zero).  Without it: the classic grid.
  float* sk_part = nullptr;
  size_t sk_part_bytes = 0;
  unsigned* sk_cnt = nullptr;
  int sk_cnt_n = 0;
  int sk_force = 0;   // 1: the balanced form wherever it can run (operator tests; the library's rule otherwise)
  int sk_upw = 0;     // > 0: (query tile, key tile) units per workgroup of the balanced form (operator tests)
};
// the factor a producer of pre-scaled queries multiplies in (head dim d)
inline float attention_q_prescale(int d) { return 1.4426950408889634f / sqrtf((float)d); }
void launch_attention(const AttnDesc& d, hipStream_t s);
bool attention_supported(int d);
// attention8.hip: head dim 64, S_k a multiple of 64 (the UNet's self-attention): LDS-DMA ring, two waves per SIMD, software-
// pipelined across key tiles.  launch_attention dispatches to it; AttnDesc::variant == 1 (or SD_ATTN8=0) keeps the general kernels.
bool attention8_ok(const AttnDesc& d);
// build-time decision (UNet builder, sd_op_attention): will attention8 run this shape?  Then V^T must be produced permuted.
bool attention8_shape_ok(int d, int Sq, int Sk);
void launch_attention8(const AttnDesc& d, hipStream_t s);
// true where attention8 would run the problem in its balanced form (query tiles dealt out evenly over the CUs, partial results
// merged by the last workgroup to arrive), with the scratch that takes: bytes for AttnDesc::sk_part, counters for sk_cnt
bool attention8_sk_scratch(const AttnDesc& d, size_t* part_bytes, int* n_counters);

// Cross-attention front half in one launch (xattn.hip): out = softmax(to_q(LayerNorm(x)) k^T / sqrt(d)) v per head,
// head dim 64, <= 96 keys (unet.py:87-118 with the prompt's K / V hoisted).  x [M][C] is the UN-normalised input; wq /
// bias / colsum are the LayerNorm-folded projection (UNet::fold_layernorm); k [B][L][C], vt [B][C][ldv], out [M][C].
struct XAttnDesc {
  const half_t* x = nullptr;
  const half_t* wq = nullptr;
  const float* bias = nullptr;
  const float* colsum = nullptr;
  const half_t* k = nullptr;
  const half_t* vt = nullptr;
  half_t* out = nullptr;
  int M = 0, C = 0, S = 0, L = 0, ldv = 0, heads = 0;   // S: query tokens per sample (M = B * S)
  float ln_eps = 1e-5f;
  int impl = kAttnOriginal;   // only checked: SPLIT_EINSUM_V2 rejects S % 512 != 0 like the streaming kernel
  int nst = 0;                // LDS-DMA ring depth 2-5 (0 = heuristic)
};
bool xattn_fused_ok(int C, int heads, int S, int L);
void launch_xattn_fused(const XAttnDesc& d, hipStream_t s);

// The whole cross-attention branch in one launch (xattn_out.hip): out = x + to_out(softmax(to_q(LayerNorm(x)) k^T / 8) v) + b_out
// (unet.py:586-591 around :87-118), C = 320 / 640 (5 / 10 heads of 64), 32 query tokens per workgroup, one wave per head.
// wq_t / wo_t: the LayerNorm-folded to_q weights and the to_out weights in the fragment-major layout of launch_xattn_out_retile
// (C * C halves each); q_bias / q_colsum as XAttnDesc::bias / colsum; o_bias [C]; x [M][C] is the UN-normalised block input and the
// residual; k [B][L][C], vt [B][C][ldv] (columns [L, ldv) zero), out [M][C].
struct XAttnOutDesc {
  const half_t* x = nullptr;
  const half_t* wq_t = nullptr;
  const float* q_bias = nullptr;
  const float* q_colsum = nullptr;
  const half_t* k = nullptr;
  const half_t* vt = nullptr;
  const half_t* wo_t = nullptr;
  const float* o_bias = nullptr;
  half_t* out = nullptr;
  int M = 0, C = 0, S = 0, L = 0, ldv = 0, heads = 0;
  float ln_eps = 1e-5f;
  int impl = kAttnOriginal;
  // five heads only: the SELF-attention's output projection in front, in the same launch (unet.py:588): x is then a1 (the
  // self-attention's output), the kernel first forms h1 = h0 + to_out1(a1) + o1_bias - which never goes to HBM - and runs the
  // cross-attention branch on it.  wo1_t fragment-major like wo_t; null = off.
  const half_t* h0 = nullptr;
  const half_t* wo1_t = nullptr;
  const float* o1_bias = nullptr;
};
// The head of a SpatialTransformer in one launch (xattn_out.hip gn_proj_qkv_kernel, round 6): GroupNorm apply (statistics from the
// producer's partials: gn_partial [B][G][kGnMaxSlabs][2], gn_entries of them per (sample, group); 0 = x is normalised already) ->
// proj_in + bias -> h (stored) -> LayerNorm-folded fused q|k|v (UNet::fold_layernorm: qkv_bias / qkv_colsum [3C]) -> qk [M][2C] with
// the queries multiplied by q_scale, vt [B][C][ldT] (vt_perm: AttnDesc::vt_perm).  wp_t (C x C) / wqkv_t (3C x C) fragment-major
// (launch_xattn_out_retile_nk).  C = 320 only.
struct GnProjQkvDesc {
  const half_t* x = nullptr;
  const float* gn_partial = nullptr;
  const float* gn_gamma = nullptr;
  const float* gn_beta = nullptr;
  int gn_entries = 0, gn_groups = 32;
  float gn_eps = 1e-6f;
  const half_t* wp_t = nullptr;
  const float* p_bias = nullptr;
  half_t* h = nullptr;
  const half_t* wqkv_t = nullptr;
  const float* qkv_bias = nullptr;
  const float* qkv_colsum = nullptr;
  float ln_eps = 1e-5f;
  half_t* qk = nullptr;
  half_t* vt = nullptr;
  int M = 0, C = 0, S = 0, ldT = 0;
  bool vt_perm = false;
  float q_scale = 1.f;
  int tok = 0;                // tokens per workgroup: 0 = the launch's rule, 32 / 64 (operator tests)
  long long* clk = nullptr;   // measurement builds of the operator test: [workgroup][wave][16] cycle stamps of the kernel's phases
};
bool gn_proj_qkv_ok(int C, int heads, int S, int M, int ldT, int G);
void launch_gn_proj_qkv(const GnProjQkvDesc& d, hipStream_t s);
bool xattn_out_ok(int C, int heads, int S, int L);
void launch_xattn_out_retile(const half_t* w, half_t* wt, int C, hipStream_t s);   // [C][C] row-major -> fragment-major
void launch_xattn_out(const XAttnOutDesc& d, hipStream_t s);
// The tail of a SpatialTransformer in one launch (xattn_out.hip ffn_proj_kernel): out = res2 + proj_out(res1 + ff.net.2(g) + b1) + b2
// (unet.py:591 FeedForward.net.2 + residual, :561-563 proj_out + residual); C = 320 only.  g [M][K1 = 4C] the GEGLU product,
// w1_t (C x K1) / w2_t (C x C) fragment-major (launch_xattn_out_retile_nk), res1 = the block's h2, res2 = the transformer's input,
// out [M][C].  gn_partial / gn_groups as ConvDesc: GroupNorm statistics of the output for its consumer; returns the entries per
// (sample, group) written (S / 32, or 0).
struct FfnProjDesc {
  const half_t* g = nullptr;
  const half_t* w1_t = nullptr;
  const float* b1 = nullptr;
  const half_t* res1 = nullptr;
  const half_t* w2_t = nullptr;
  const float* b2 = nullptr;
  const half_t* res2 = nullptr;
  half_t* out = nullptr;
  float* gn_partial = nullptr;
  int gn_groups = 0;
  int M = 0, C = 0, K1 = 0, S = 0;   // S: tokens per sample
};
bool ffn_proj_ok(int C, int K1, int M, int S);
int launch_ffn_proj(const FfnProjDesc& d, hipStream_t s);
void launch_xattn_out_retile_nk(const half_t* w, half_t* wt, int N, int K, hipStream_t s);   // [N][K] row-major -> fragment-major

// ---------------------------------------------------------------------------------------------
// K6/K7: norms (norm.hip)
// ---------------------------------------------------------------------------------------------
// y[m][:] = (x[m][:]-mean)*rstd * w + b over the channel dim (LayerNormANE, layer_norm.py:51-80)
void launch_layernorm(const half_t* x, const float* w, const float* b, half_t* y, int M, int C, float eps,
                      hipStream_t s);
// GroupNorm over NHWC with optional channel-concat second source: deterministic two-pass
// statistics (partial: groupnorm_scratch_floats() floats per call) + apply.
int groupnorm_num_slabs(int B, int HW);
size_t groupnorm_scratch_floats(int B, int HW, int G);
// producer_entries > 0: `partial` already holds that many (sum, sumsq) entries per (sample, group), written by the
// epilogue of the kernel that produced x0 (ConvDesc::gn_partial) - the statistics pass is skipped.
void launch_groupnorm(const half_t* x0, int C0, const half_t* x1, int C1, float* partial, const float* gamma,
                      const float* beta, half_t* y, int B, int HW, int G, float eps, int silu, hipStream_t s,
                      int producer_entries = 0, const ConvDesc* side = nullptr);
// `side` (round 5): an INDEPENDENT 1x1 GEMM - the resnet's conv_shortcut over the same input (unet.py:483-486) - that runs in the
// SAME launch as the GroupNorm's apply / single-launch kernel (igemm.hip gn_*_side_kernel: one grid, the first blocks are the
// GroupNorm, the rest the GEMM's tiles).  Must satisfy gn_side_gemm_ok; its result is complete when the launch is.
bool gn_side_gemm_ok(const ConvDesc& d);
void launch_gn_apply_side(const half_t* x0, int C0, const half_t* x1, int C1, const float* partial, int entries, const float* gamma,
                          const float* beta, half_t* y, int B, int HW, int G, float eps, int silu, int slabs, int ppb, const ConvDesc& side,
                          hipStream_t s);
void launch_gn_fused_side(int vw, const half_t* x0, int C0, const half_t* x1, int C1, const float* gamma, const float* beta, half_t* y, int B,
                          int HW, int G, float eps, int silu, const ConvDesc& side, hipStream_t s);
// true when a GroupNorm of this shape would run a separate statistics launch, i.e. producer statistics pay
bool groupnorm_wants_producer_stats(int HW, int C, int G);

// in-place softmax(scale * x) over each row of a [rows][cols] fp16 matrix (VAE single-head attention)
void launch_row_softmax(half_t* x, int rows, int cols, float scale, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// fp32 compute path of the VAE graphs (vae_f32.hip; torch2coreml.py:570-578, :726-733): fp32 NHWC activations
// ---------------------------------------------------------------------------------------------
struct ConvF32Desc {
  const float* x = nullptr;     // [B][Hi][Wi][Cin]
  const void* w = nullptr;      // w_kind 0: half [N][K], 1: float [N][K], 2: float [K][N];  K = ksize*ksize*Cin, tap-major
  int w_kind = 0;
  const float* bias = nullptr;  // [N] or null
  const float* res = nullptr;   // [M][N] or null
  float* out = nullptr;         // [M][N], M = B*Ho*Wo
  int B = 1, Hi = 1, Wi = 1, Cin = 0, Ho = 1, Wo = 1, N = 0;
  int ksize = 1, stride = 1, up = 1;
  int pad = -1;                 // -1: ksize / 2; 0 with stride 2: the VAE encoder's F.pad(x, (0, 1, 0, 1))
};
void launch_conv_f32(const ConvF32Desc& d, hipStream_t s);
size_t groupnorm_f32_scratch_bytes(int B, int G);
void launch_groupnorm_f32(const float* x, void* scratch, const float* gamma, const float* beta, float* y, int B, int HW, int C, int G,
                          float eps, int silu, hipStream_t s);
void launch_row_softmax_f32(float* x, int rows, int cols, float scale, hipStream_t s);
void launch_nchw_to_nhwc_f32(const void* src, int src_is_f32, float* dst, int B, int C, int H, int W, hipStream_t s);
void launch_nhwc_to_nchw_f32f32(const float* src, float* dst, int B, int C, int H, int W, hipStream_t s);

// ---------------------------------------------------------------------------------------------
// K8/K9 + boundary helpers (misc.hip)
// ---------------------------------------------------------------------------------------------
// sinusoidal embedding [cos|sin] (unet.py:703-728), fp32.  t: [n] fp32, freq: [dim/2] device
// copy of timestep_freq_table(dim, freq_shift) -> out [n][dim]
std::vector<float> timestep_freq_table(int dim, float freq_shift);
void launch_timestep_embedding(const float* t, const float* freq, float* out, int n, int dim, hipStream_t s);
// out[b][n] = act_out( sum_k W[n][k]*act_in(x[b][k]) + bias[n] ); x fp32 [B][ldx], out fp32 [B][ldo]
void launch_gemv(const half_t* w, const float* bias, const float* x, int ldx, float* out, int ldo, int B, int N,
                 int K, int silu_in, int silu_out, int accumulate, hipStream_t s);
void launch_nchw_to_nhwc(const void* src, int src_is_f32, half_t* dst, int B, int C, int H, int W, hipStream_t s);
void launch_nhwc_to_nchw_f32(const half_t* src, float* dst, int B, int C, int H, int W, hipStream_t s);
void launch_half_to_float(const half_t* src, float* dst, size_t n, hipStream_t s);
void launch_float_to_half(const float* src, half_t* dst, size_t n, hipStream_t s);
// y = a + b (fp16), used for ControlNet residual adds when not fused
void launch_add_half(const half_t* a, const half_t* b, half_t* y, size_t n, hipStream_t s);
// y = srcs[0] + ... + srcs[nsrc-1] (nsrc <= 4, fp32 accumulation): device-side ControlNet residual sum
void launch_sum_half(const half_t* const* srcs, int nsrc, half_t* y, size_t n, hipStream_t s);
// BC1S (B,C,1,S) fp16 -> token-major [B][S][C] fp16 (encoder_hidden_states boundary)
void launch_bc1s_to_tokens(const half_t* src, half_t* dst, int B, int C, int S, hipStream_t s);

// Device-resident denoising loop helpers (pipeline.py:500-573).  `step` is a device counter.
struct LoopTables {
  const float* timesteps;   // [n_steps]
  const float* coef;        // [n_steps][8]: cx, cm, ch0..ch2, a, b, flags  (cfg_sched_step_kernel, misc.hip)
  int* step;                // device scalar
  const float* in_scale;    // [n_steps] scale_model_input factor (sigma-space schedulers), or null
  unsigned* ticket;         // device scalar (0 between launches): arrival counter of cfg_sched_step_kernel
  // time-embedding table of the whole schedule (or null): [n_steps][temb_rows][temb_ld] rows precomputed by the UNet's time
  // path for every timestep; loop_prep copies row block `step` to temb_dst, so the step itself launches no time-path kernel
  const float* temb_tab;
  float* temb_dst;
  int temb_rows, temb_ld, temb_n;
  // ancestral samplers: [n_steps][Bimg * CHW] noise (already scaled by sigma_up) added to the latents after step `step`, or null
  const float* noise_tab = nullptr;
};
// latents fp32 NCHW [Bimg][4][H][W] -> UNet sample fp16 NHWC [cfg*Bimg][H][W][4], timestep buffer
void launch_loop_prep(const float* latents, half_t* sample, float* tbuf, LoopTables t, int Bimg, int C, int H,
                      int W, int cfg, hipStream_t s);
// eps = u + g*(c-u) (pipeline.py:561-562); m = a*x + b*eps; latents = cx*x + cm*m + sum ch_j * hist_j; step++
void launch_cfg_sched_step(const float* noise_pred, float* latents, float* eps_hist, LoopTables t, float guidance,
                           int Bimg, int CHW, int cfg, int hist, hipStream_t s);

void launch_lds_poison(hipStream_t s);   // debug (SD_POISON_LDS): NaN patterns into every CU's LDS
void launch_count_nonfinite_half(const void* p, size_t bytes, unsigned long long* out, hipStream_t s);   // debug scan (SD_NAN_TRACE)

}  // namespace sd
