// UNet2DConditionModel / UNet2DConditionModelXL / ControlNetModel as a static launch graph of
// the gfx950 kernels.  Every comment "unet.py:NNN" / "controlnet.py:NNN" cites the reference line
// whose arithmetic the emitted ops reproduce (python_coreml_stable_diffusion/).
//
// Design notes
//  * static shapes like the reference's Core ML models (pipeline.py:112-114): everything is
//    allocated once, each tensor owns its bytes (288 GB HBM), so the forward is a pure launch
//    list and is captured into one HIP graph (~700 launches -> one replay).
//  * channels-last fp16 activations; weights re-laid-out at load to [Cout][ky][kx][Cin];
//    torch.cat skip concat (unet.py:213-216) is never materialised: GroupNorm and the 1x1
//    shortcut read both sources.
//  * prompt-constant work is hoisted: to_k/to_v of every cross-attention (unet.py:95-96) run
//    when encoder_hidden_states changes, not every step (SURVEY.md Appendix E obs. 3); all 22
//    time_emb_proj(SiLU(emb)) (unet.py:477) are one batched GEMV per step.
#include "unet.h"

#ifndef SD_GN_QKV_DEFAULT
#define SD_GN_QKV_DEFAULT 1   // the one-launch head of a SpatialTransformer (UNet::transformer); 0 = the three launches it replaces
#endif

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace sd {

namespace {
constexpr int kTembCap = 65536;      // floats per batch row for the batched time_emb_proj outputs
inline int round_up(int x, int a) { return (x + a - 1) / a * a; }
}  // namespace

UNet::UNet(const sd_unet_config& cfg, const WeightStore& ws, int device) : cfg_(cfg), ws_(&ws), device_(device) {
  SD_REQUIRE(cfg.n_levels >= 1 && cfg.n_levels <= SD_MAX_LEVELS, kInvalidArgument, "n_levels=%d", cfg.n_levels);
  SD_REQUIRE(cfg.batch >= 1 && cfg.height >= 1 && cfg.width >= 1, kInvalidArgument, "bad batch/size");
  SD_REQUIRE(cfg.norm_num_groups >= 1 && cfg.norm_num_groups <= 64, kUnsupported, "norm_num_groups=%d",
             cfg.norm_num_groups);
  int ndev = 0;
  SD_HIP(hipGetDeviceCount(&ndev));
  SD_REQUIRE(ndev > 0, kHipError, "no HIP device visible: libsdmi355 has no CPU fallback");
  SD_REQUIRE(device >= 0 && device < ndev, kInvalidArgument, "device %d out of range (%d visible)", device, ndev);
  SD_HIP(hipSetDevice(device));
  SD_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  if (tune_env_int("SD_SIDE_TIME", 0) != 0) {
    SD_HIP(hipStreamCreateWithFlags(&side_, hipStreamNonBlocking));
    SD_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
    SD_HIP(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
  }
  if (cfg_.support_controlnet && !cfg_.is_controlnet && tune_env_int("SD_CN_CONCURRENT", 1) != 0) {
    SD_HIP(hipStreamCreateWithFlags(&cn_stream_, hipStreamNonBlocking));
    SD_HIP(hipEventCreateWithFlags(&ev_cn_fork_, hipEventDisableTiming));
    SD_HIP(hipEventCreateWithFlags(&ev_cn_join_, hipEventDisableTiming));
  }
  f32_ = cfg_.compute_fp32 != 0;
  SD_REQUIRE(!f32_ || cfg_.is_vae_decoder, kUnsupported,
             "compute_fp32 is the VAE graphs' option (torch2coreml.py:570-578, :726-733); the UNet kernels store fp16");
  if (cfg_.is_vae_decoder == 2)
    build_vae_encoder();
  else if (cfg_.is_vae_decoder)
    build_vae_decoder();
  else
    build_unet();
  ws_ = nullptr;
  SD_HIP(hipStreamSynchronize(stream_));
}

UNet::~UNet() {
  (void)hipSetDevice(device_);
  if (graph_) (void)hipGraphExecDestroy(graph_);
  if (loop_graph_) (void)hipGraphExecDestroy(loop_graph_);
  if (stream_) {
    (void)hipStreamSynchronize(stream_);
    (void)hipStreamDestroy(stream_);
  }
  if (side_) {
    (void)hipStreamSynchronize(side_);
    (void)hipStreamDestroy(side_);
    (void)hipEventDestroy(ev_fork_);
    (void)hipEventDestroy(ev_join_);
  }
  if (cn_stream_) {
    (void)hipStreamSynchronize(cn_stream_);
    (void)hipStreamDestroy(cn_stream_);
    (void)hipEventDestroy(ev_cn_fork_);
    (void)hipEventDestroy(ev_cn_join_);
  }
}

Tensor UNet::new_tensor(int B, int H, int W, int C) {
  Tensor t;
  t.B = B;
  t.H = H;
  t.W = W;
  t.C = C;
  t.p = arena_.alloc_n<half_t>(t.numel() * (f32_ ? 2 : 1));   // compute_fp32: the same bookkeeping over float elements
  return t;
}

// [Cout][Cin][k][k] (or Linear [Cout][Cin]) -> [Cout][k][k][Cin] fp16.  GEGLU (ff.net.0.proj,
// unet.py:613-617): rows re-ordered so value/gate channels interleave in blocks of 32 and the
// GEMM epilogue can multiply them in registers.
half_t* UNet::upload_conv_weight(const std::string& name, int cout, int cin, int k, bool geglu) {
  const HostTensor& t = ws_->get(name + ".weight");
  const size_t expect = (size_t)cout * cin * k * k;
  SD_REQUIRE(t.numel() == expect, kInvalidArgument, "%s.weight has %zu elements, expected %zu (%d,%d,%d,%d)",
             name.c_str(), t.numel(), expect, cout, cin, k, k);
  std::vector<half_t> host(expect);
  const int kk = k * k;
  for (int o = 0; o < cout; ++o) {
    int dst_o = o;
    if (geglu) {
      const int half_n = cout / 2;
      const bool gate = o >= half_n;
      const int j = gate ? o - half_n : o;
      dst_o = (j / 32) * 64 + (gate ? 32 : 0) + (j % 32);
    }
    for (int c = 0; c < cin; ++c)
      for (int t2 = 0; t2 < kk; ++t2)
        host[((size_t)dst_o * kk + t2) * cin + c] = (half_t)t.data[((size_t)o * cin + c) * kk + t2];
  }
  half_t* d = arena_.alloc_n<half_t>(expect);
  SD_HIP(hipMemcpy(d, host.data(), expect * sizeof(half_t), hipMemcpyHostToDevice));
  return d;
}

float* UNet::upload_vec(const std::string& name, int n, bool geglu) {
  const HostTensor& t = ws_->get(name);
  SD_REQUIRE((int)t.numel() == n, kInvalidArgument, "%s has %zu elements, expected %d", name.c_str(), t.numel(), n);
  std::vector<float> host(t.data);
  if (geglu) {
    const int half_n = n / 2;
    for (int o = 0; o < n; ++o) {
      const bool gate = o >= half_n;
      const int j = gate ? o - half_n : o;
      host[(j / 32) * 64 + (gate ? 32 : 0) + (j % 32)] = t.data[o];
    }
  }
  float* d = arena_.alloc_n<float>(n);
  SD_HIP(hipMemcpy(d, host.data(), (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}

Tensor UNet::conv(std::vector<Op>& ops, const std::string& name, const Tensor& x, const Tensor* x2, int cout, int k,
                  int stride, int up, bool bias, const float* temb, const half_t* res, int out_mode, int ldT,
                  bool silu_out, int pad) {
  const int cin = x.C + (x2 ? x2->C : 0);
  const bool geglu = out_mode == kOutGeglu;
  const float* b = bias ? upload_vec(name + ".bias", cout, geglu) : nullptr;
  if (f32_) {
    // fp32 compute path (torch2coreml.py:570-578 converts an SDXL checkpoint's own VAE with FLOAT32 precision): the weights stay
    // fp32 too - an fp32 checkpoint is not rounded to fp16 on the way in - in the [N][taps][Cin] layout of the fp16 path
    const HostTensor& t = ws_->get(name + ".weight");
    const size_t expect = (size_t)cout * cin * k * k;
    SD_REQUIRE(!geglu && t.numel() == expect, kInvalidArgument, "%s.weight has %zu elements, expected %zu", name.c_str(), t.numel(), expect);
    std::vector<float> host(expect);
    const int kk = k * k;
    for (int o = 0; o < cout; ++o)
      for (int c = 0; c < cin; ++c)
        for (int t2 = 0; t2 < kk; ++t2) host[((size_t)o * kk + t2) * cin + c] = t.data[((size_t)o * cin + c) * kk + t2];
    float* d = arena_.alloc_n<float>(expect);
    SD_HIP(hipMemcpy(d, host.data(), expect * sizeof(float), hipMemcpyHostToDevice));
    w_f32_pending_ = true;   // consumed by conv_w's fp32 branch (ConvF32Desc::w_kind 1)
    return conv_w(ops, name, reinterpret_cast<const half_t*>(d), b, x, x2, cout, k, stride, up, temb, res, out_mode, ldT, silu_out,
                  nullptr, pad);
  }
  const half_t* w = upload_conv_weight(name, cout, cin, k, geglu);
  return conv_w(ops, name, w, b, x, x2, cout, k, stride, up, temb, res, out_mode, ldT, silu_out, nullptr, pad);
}

// Several bias-free 1x1 projections of the same input as ONE GEMM: weights stacked along Cout
// (attn1.to_q | attn1.to_k, unet.py:93-95): fewer launches, bigger N, same arithmetic.
Tensor UNet::conv_stacked(std::vector<Op>& ops, const std::vector<std::string>& names, const Tensor& x, int cout_each) {
  const int cin = x.C, n = (int)names.size();
  std::vector<half_t> host((size_t)n * cout_each * cin);
  for (int i = 0; i < n; ++i) {
    const HostTensor& t = ws_->get(names[i] + ".weight");
    SD_REQUIRE(t.numel() == (size_t)cout_each * cin, kInvalidArgument, "%s.weight: bad shape", names[i].c_str());
    for (size_t j = 0; j < t.numel(); ++j) host[(size_t)i * cout_each * cin + j] = (half_t)t.data[j];
  }
  half_t* d = arena_.alloc_n<half_t>(host.size());
  SD_HIP(hipMemcpy(d, host.data(), host.size() * sizeof(half_t), hipMemcpyHostToDevice));
  return conv_w(ops, names[0], d, nullptr, x, nullptr, n * cout_each, 1, 1, 1, nullptr, nullptr, kOutHalf, 0, false);
}

// y = LN(x) . W^T + b  ==  rstd*(x . (W*gamma)^T) - rstd*mean*colsum(W*gamma) + (b + W.beta): the
// GEMM runs on the raw rows and applies the row statistics in its epilogue (igemm LNF kernels), so
// the separate LayerNorm launch, its HBM round trip and the fp16 rounding of LN(x) all disappear.
UNet::LnFold UNet::fold_layernorm(const std::string& ln, const std::vector<std::string>& names, int cin, int cout_each,
                                  bool geglu) {
  const HostTensor& g = ws_->get(ln + ".weight");
  const HostTensor& be = ws_->get(ln + ".bias");
  SD_REQUIRE((int)g.numel() == cin && (int)be.numel() == cin, kInvalidArgument, "%s: expected %d channels", ln.c_str(),
             cin);
  const int n = (int)names.size(), ntot = n * cout_each;
  SD_REQUIRE(!geglu || n == 1, kInternal, "GEGLU fold takes one projection");
  std::vector<half_t> w((size_t)ntot * cin);
  std::vector<float> colsum(ntot), bias(ntot);
  for (int i = 0; i < n; ++i) {
    const HostTensor& t = ws_->get(names[i] + ".weight");
    SD_REQUIRE(t.numel() == (size_t)cout_each * cin, kInvalidArgument, "%s.weight: bad shape", names[i].c_str());
    const HostTensor* tb = ws_->has(names[i] + ".bias") ? &ws_->get(names[i] + ".bias") : nullptr;
    SD_REQUIRE(!tb || (int)tb->numel() == cout_each, kInvalidArgument, "%s.bias: bad shape", names[i].c_str());
    for (int o = 0; o < cout_each; ++o) {
      int dst = i * cout_each + o;
      if (geglu) {   // same value/gate interleave as upload_conv_weight
        const int half_n = cout_each / 2;
        const bool gate = o >= half_n;
        const int j = gate ? o - half_n : o;
        dst = (j / 32) * 64 + (gate ? 32 : 0) + (j % 32);
      }
      double cs = 0.0, bb = tb ? (double)tb->data[o] : 0.0;
      for (int c = 0; c < cin; ++c) {
        const float wv = t.data[(size_t)o * cin + c];
        const half_t h = (half_t)(wv * g.data[c]);
        w[(size_t)dst * cin + c] = h;
        cs += (double)(float)h;          // column sum of what the MFMA actually multiplies
        bb += (double)wv * (double)be.data[c];
      }
      colsum[dst] = (float)cs;
      bias[dst] = (float)bb;
    }
  }
  LnFold f;
  f.w = arena_.alloc_n<half_t>(w.size());
  f.colsum = arena_.alloc_n<float>(ntot);
  f.bias = arena_.alloc_n<float>(ntot);
  SD_HIP(hipMemcpy(f.w, w.data(), w.size() * sizeof(half_t), hipMemcpyHostToDevice));
  SD_HIP(hipMemcpy(f.colsum, colsum.data(), ntot * sizeof(float), hipMemcpyHostToDevice));
  SD_HIP(hipMemcpy(f.bias, bias.data(), ntot * sizeof(float), hipMemcpyHostToDevice));
  return f;
}

bool UNet::can_fold_ln(const Tensor& x, int cout, bool geglu) const {
  ConvDesc d;
  d.C0 = x.C;
  d.N = cout;
  d.ksize = 1;
  d.out_mode = geglu ? kOutGeglu : kOutHalf;
  static const bool off = tune_env_set("SD_NO_LN_FOLD");   // A/B switch for measurements
  return !off && conv_fast_path_ok(d);
}

Tensor UNet::conv_w(std::vector<Op>& ops, const std::string& name, const half_t* w, const float* bias, const Tensor& x,
                    const Tensor* x2, int cout, int k, int stride, int up, const float* temb, const half_t* res,
                    int out_mode, int ldT, bool silu_out, ConvExtra* ex, int pad_override) {
  const bool geglu = out_mode == kOutGeglu;
  ConvDesc d;
  d.x0 = x.p;
  d.C0 = x.C;
  if (x2) {
    SD_REQUIRE(x2->B == x.B && x2->H == x.H && x2->W == x.W, kInternal, "%s: concat shape mismatch", name.c_str());
    d.x1 = x2->p;
    d.C1 = x2->C;
  }
  d.w = w;
  d.bias = bias;
  d.temb = temb;
  d.temb_stride = kTembCap;
  d.res = res;
  d.B = x.B;
  d.Hi = x.H;
  d.Wi = x.W;
  const int Hup = x.H * up, Wup = x.W * up, pad = k / 2;
  if (pad_override >= 0) {   // diffusers Downsample2D(padding=0): F.pad(x, (0, 1, 0, 1)) then a pad-0 stride-2 conv
    SD_REQUIRE(pad_override == 0 && stride == 2 && k == 3, kUnsupported, "%s: explicit padding %d", name.c_str(), pad_override);
    d.pad = 0;
    d.Ho = (Hup + 1 - k) / stride + 1;
    d.Wo = (Wup + 1 - k) / stride + 1;
  } else {
    d.Ho = (Hup + 2 * pad - k) / stride + 1;
    d.Wo = (Wup + 2 * pad - k) / stride + 1;
  }
  d.ksize = k;
  d.stride = stride;
  d.up = up;
  d.N = cout;
  d.out_mode = out_mode;
  d.ldT = ldT;
  Tensor out;
  std::string gn_in_name;   // set when a queued GroupNorm (pending_gn_in_) became part of this op
  if (ex) d.ln_colsum = ex->ln_colsum;
  if (ex) d.vt_perm = ex->vt_perm ? 1 : 0;
  if (ex) {
    d.q_scale = ex->q_scale;
    d.q_cols = ex->q_cols;
  }
  if (ex && ex->n_trans > 0) {   // fused q|k|v: [M][n_trans] row-major + V^T [B][cout - n_trans][ldT]
    out = new_tensor(x.B, d.Ho, d.Wo, ex->n_trans);
    ex->vt = arena_.alloc_n<half_t>((size_t)x.B * (cout - ex->n_trans) * ldT);
    d.out_t = ex->vt;
    d.n_trans = ex->n_trans;
  } else if (out_mode == kOutHalfT) {
    out.B = x.B; out.H = 1; out.W = ldT; out.C = cout;   // [B][cout][ldT]
    out.p = arena_.alloc_n<half_t>((size_t)x.B * cout * ldT);
  } else {
    out = new_tensor(x.B, d.Ho, d.Wo, geglu ? cout / 2 : cout);
  }
  d.out = out.p;
  if (f32_) {   // VAE handle with compute_fp32: the same op on the fp32 kernels (vae_f32.hip)
    SD_REQUIRE(!x2 && !temb && !ex && out_mode == kOutHalf && !silu_out, kInternal, "%s: not a VAE conv", name.c_str());
    ConvF32Desc fd;
    fd.x = reinterpret_cast<const float*>(x.p);
    fd.w = w;
    fd.w_kind = w_f32_pending_ ? 1 : 0;   // UNet::conv uploads fp32 weights for an fp32 handle
    w_f32_pending_ = false;
    fd.bias = bias;
    fd.res = reinterpret_cast<const float*>(res);
    fd.out = reinterpret_cast<float*>(out.p);
    fd.B = x.B; fd.Hi = x.H; fd.Wi = x.W; fd.Cin = x.C; fd.Ho = d.Ho; fd.Wo = d.Wo; fd.N = cout;
    fd.ksize = k; fd.stride = stride; fd.up = up; fd.pad = d.pad;
    ops.push_back([fd](hipStream_t s) { launch_conv_f32(fd, s); });
    char buf[256];
    snprintf(buf, sizeof(buf), "conv%dx%d fp32 %d->%d @%dx%d %s", k, k, x.C, cout, d.Ho, d.Wo, name.c_str());
    ops.back().label = buf;
    ops.back().flop = 2.0 * x.B * d.Ho * d.Wo * (double)cout * x.C * k * k;
    return out;
  }
  SD_REQUIRE(!ex || (conv_fast_path_ok(d) && !silu_out), kInternal, "%s: LayerNorm fold / fused q|k|v off the MFMA path",
             name.c_str());
  // a GroupNorm built later over exactly this tensor may ask for its statistics from this op's epilogue (GnHook)
  std::shared_ptr<GnHook> hook;
  if (out_mode == kOutHalf && !(ex && ex->n_trans > 0)) {
    hook = std::make_shared<GnHook>();
    out.gn = hook;
    hook->ops_pos = (int)ops.size();
    hook->ops_list = &ops;
  }
  auto with_hook = [hook](const ConvDesc& d0) {
    ConvDesc dd = d0;
    if (hook) {
      dd.gn_partial = hook->partial;
      dd.gn_groups = hook->groups;
      dd.n_twins = hook->n_twins;
      for (int k = 0; k < hook->n_twins; ++k) dd.twin[k] = hook->twin[k];
    }
    return dd;
  };
  if (conv_fast_path_ok(d) && !silu_out) {
    // small-M layers: the weight-streaming kernel (plan tile 9) needs the fragment-major copy of the weights
    static const int ws_mode = tune_env_int("SD_WSTREAM", 1);
    const int Mrows = x.B * d.Ho * d.Wo;
    // (SD_TUNE: every shape a plan sweep may send there; production: exactly the convs whose plan IS tile 9)
    static const bool tuning = getenv("SD_TUNE") != nullptr;
    const bool ws_candidate = tuning ? ((k == 3 && Mrows <= 512) || (k == 1 && Mrows <= 128)) : conv_plan_is_wstream(d);
    if (ws_mode != 0 && !ex && wstream_shape_ok(d) && ws_candidate) {
      const int cin_t = x.C + (x2 ? x2->C : 0);
      half_t* wt = arena_.alloc_n<half_t>(wstream_tiled_halves(cout, cin_t, k));
      launch_wstream_retile(w, wt, cout, cin_t, k, stream_);
      d.w_tiled = wt;
    }
    {   // widest-M GEGLU projection (K = 320): the weight-stationary kernel reads its own fragment-major copy (wsgemm.hip)
      static const bool wsg_on = tune_env_int("SD_WSGEMM", 1) != 0;
      if (wsg_on && wsgemm_wanted(d)) {
        half_t* wt = arena_.alloc_n<half_t>(wsgemm_tiled_halves(cout));
        launch_wsgemm_retile(w, wt, cout, geglu, stream_);
        d.w_ws = wt;
      }
    }
    {   // large-M 1x1 GEMMs: weights global -> VGPR (bvgemm.hip) reads its own fragment-major copy
      static const bool bv_on = tune_env_int("SD_BVGEMM", 1) != 0;
      if (bv_on && !d.w_ws && bvgemm_wanted(d)) {
        half_t* wt = arena_.alloc_n<half_t>(bvgemm_tiled_halves(cout, x.C));
        launch_bvgemm_retile(w, wt, cout, x.C, geglu, stream_);
        d.w_bv = wt;
      }
    }
    if (hook) {
      hook->twin_capable = !ex && d.Ho * d.Wo <= 256;
      hook->desc = d;
    }
    ws_need_ = std::max(ws_need_, conv_workspace_bytes(d));
    std::shared_ptr<GnIn> gi = std::move(pending_gn_in_);
    pending_gn_in_.reset();
    if (gi) {   // x is the UN-normalised tensor (gn_loader_conv checked the shape)
      gn_in_name = gi->norm_name;
      const half_t* xp = x.p;
      const int C = x.C, B = x.B, HW = x.H * x.W;
      ops.push_back([this, d, hook, with_hook, gi, xp, C, B, HW](hipStream_t s) {
        ConvDesc dd = with_hook(d);
        const int e = gi->hook->consume();
        if (e >= 1 && e <= 128) {   // the producer left its statistics: GroupNorm(+SiLU) in the halo loader
          dd.gnf_partial = gi->partial;
          dd.gnf_gamma = gi->gamma;
          dd.gnf_beta = gi->beta;
          dd.gnf_eps = gi->eps;
          dd.gnf_groups = gi->groups;
          dd.gnf_entries = e;
          dd.gnf_silu = gi->silu;
        } else {                    // split-K / ragged producer: the GroupNorm launch, then the plain conv
          launch_groupnorm(xp, C, nullptr, 0, gi->partial, gi->gamma, gi->beta, gi->y, B, HW, gi->groups, gi->eps, gi->silu, s, e);
          dd.x0 = gi->y;
        }
        const int n = launch_conv(dd, ws_conv_, s);
        if (hook) hook->produced(n);
      });
    } else {
      ops.push_back([this, d, hook, with_hook](hipStream_t s) {
        const int n = launch_conv(with_hook(d), ws_conv_, s);
        if (hook) hook->produced(n);
      });
    }
  } else {
    const int so = silu_out ? 1 : 0;
    ops.push_back([d, so, hook, with_hook](hipStream_t s) {
      const int n = launch_conv_generic(with_hook(d), so, s);
      if (hook) hook->produced(n);
    });
  }
  SD_REQUIRE(!pending_gn_in_, kInternal, "%s: a GroupNorm was queued for a conv that cannot take it", name.c_str());
  {
    const int cin = x.C + (x2 ? x2->C : 0);
    char buf[320];
    // trailing "#kind,ksize,stride,up,Ctot,N,M" is the plan-table key of this op (tools/tune_plans.py)
    const int kind = d.out_t ? 3 : (geglu ? 2 : (d.ln_colsum ? 1 : 0));
    snprintf(buf, sizeof(buf), "%s%s %d->%d @%dx%d M=%d K=%d %s #%d,%d,%d,%d,%d,%d,%d", k == 3 ? "conv3x3" : (geglu ? "geglu1x1" : "gemm1x1"),
             ex && ex->ln_colsum ? "+ln" : "", cin, cout, d.Ho, d.Wo, x.B * d.Ho * d.Wo, cin * k * k, name.c_str(), kind, k,
             stride, up, cin, cout, x.B * d.Ho * d.Wo);
    ops.back().label = buf;
    if (!gn_in_name.empty()) {   // "... <conv name> +gn(<norm name>) #key": the family prefix and the plan key stay where the tools look
      std::string l = buf;
      const size_t at = l.rfind(" #");
      ops.back().label = l.substr(0, at) + " +gn(" + gn_in_name + ")" + l.substr(at);
    }
    ops.back().flop = 2.0 * x.B * d.Ho * d.Wo * (double)cout * cin * k * k;
  }
  return out;
}

Tensor UNet::group_norm(std::vector<Op>& ops, const std::string& name, const Tensor& x, const Tensor* x2, float eps,
                        bool silu, const ConvDesc* side, const std::string& side_label, double side_flop) {
  SD_REQUIRE(!side || !f32_, kInternal, "%s: side GEMM on the fp32 path", name.c_str());
  const int C = x.C + (x2 ? x2->C : 0);
  const int G = cfg_.norm_num_groups;
  SD_REQUIRE(C % G == 0, kUnsupported, "%s: %d channels not divisible by %d groups", name.c_str(), C, G);
  if (f32_) {
    SD_REQUIRE(!x2, kInternal, "%s: concat GroupNorm on the fp32 path", name.c_str());
    void* scratch = arena_.alloc(groupnorm_f32_scratch_bytes(x.B, G));
    const float* gamma = upload_vec(name + ".weight", C);
    const float* beta = upload_vec(name + ".bias", C);
    Tensor y = new_tensor(x.B, x.H, x.W, C);
    const float* xp = reinterpret_cast<const float*>(x.p);
    float* yp = reinterpret_cast<float*>(y.p);
    const int B = x.B, HW = x.H * x.W, si = silu ? 1 : 0;
    ops.push_back([=](hipStream_t s) { launch_groupnorm_f32(xp, scratch, gamma, beta, yp, B, HW, C, G, eps, si, s); });
    ops.back().label = "groupnorm fp32 C=" + std::to_string(C) + " @" + std::to_string(x.H) + "x" + std::to_string(x.W) + " " + name;
    return y;
  }
  const float* gamma = upload_vec(name + ".weight", C);
  const float* beta = upload_vec(name + ".bias", C);
  Tensor y = new_tensor(x.B, x.H, x.W, C);
  // 8x8 / 16x16 levels: every source is the output of a conv that can leave through the group-organised slab combine, and the
  // group boundaries fall on the source boundaries -> the producers write this GroupNorm themselves (GnHook twins), no launch.
  // SD_GN_TWIN=0 (with SD_TUNE) keeps the launch: A/B.
  {
    static const int twin_mode = tune_env_int("SD_GN_TWIN", 0);   // measured and rejected (LAB_NOTES.md r5): off unless asked for
    static const int twin_max_hw = tune_env_int("SD_GN_TWIN_MAX_HW", 256);   // (A/B: 64 = the 8x8 level only)
    const int cpg = C / G;
    const Tensor* srcs[2] = {&x, x2};
    bool ok = twin_mode != 0 && !side && x.H * x.W <= std::min(256, twin_max_hw) && cpg % 4 == 0 && (!x2 || x.C % cpg == 0);
    for (int i = 0; i < 2 && ok; ++i) {
      const Tensor* t = srcs[i];
      if (!t) continue;
      const GnHook* hk = t->gn.get();
      ok = hk && hk->twin_capable && hk->ops_list == &ops && hk->n_twins < 2 && hk->desc.N == t->C && t->C % cpg == 0;
      if (ok) {
        GnTwin tw[2];
        int n = hk->n_twins;
        for (int k = 0; k < n; ++k) tw[k] = hk->twin[k];
        tw[n].cpg = cpg;
        tw[n].c_off = i == 0 ? 0 : x.C;
        tw[n].ld = C;
        ok = reduce_twin_ok(t->H * t->W, t->C, n + 1, tw);
      }
    }
    if (ok) {
      for (int i = 0; i < 2; ++i) {
        const Tensor* t = srcs[i];
        if (!t) continue;
        GnHook* hk = t->gn.get();
        GnTwin& tw = hk->twin[hk->n_twins++];
        tw.y = y.p;
        tw.ld = C;
        tw.c_off = i == 0 ? 0 : x.C;
        tw.cpg = cpg;
        tw.gamma = gamma;
        tw.beta = beta;
        tw.eps = eps;
        tw.silu = silu ? 1 : 0;
        ConvDesc dd = hk->desc;
        dd.n_twins = hk->n_twins;
        ws_need_ = std::max(ws_need_, conv_workspace_bytes(dd));
      }
      return y;
    }
  }
  float* partial = arena_.alloc_n<float>(groupnorm_scratch_floats(x.B, x.H * x.W, G));
  const half_t* p0 = x.p;
  const half_t* p1 = x2 ? x2->p : nullptr;
  const int C0 = x.C, C1 = x2 ? x2->C : 0, B = x.B, HW = x.H * x.W, si = silu ? 1 : 0;
  half_t* yp = y.p;
  // single-source GroupNorm that would need its own statistics launch: ask the op that produced x for them
  std::shared_ptr<GnHook> hook;
  if (!x2 && x.gn && !x.gn->partial && x.gn->n_twins == 0 && groupnorm_wants_producer_stats(HW, C, G)) {
    // (`entries` is handed over at launch time - the producer must run first, in the same launch list)
    SD_REQUIRE(x.gn->ops_list == &ops && x.gn->ops_pos >= 0 && x.gn->ops_pos < (int)ops.size(), kInternal,
               "%s: GroupNorm producer statistics need the producing op earlier in the same launch list", name.c_str());
    hook = x.gn;
    hook->partial = partial;
    hook->groups = G;
  }
  if (side) {
    const ConvDesc sd = *side;
    ops.push_back([=](hipStream_t s) {
      launch_groupnorm(p0, C0, p1, C1, partial, gamma, beta, yp, B, HW, G, eps, si, s, hook ? hook->consume() : 0, &sd);
    });
    ops.back().label = "groupnorm C=" + std::to_string(C) + " @" + std::to_string(x.H) + "x" + std::to_string(x.W) + " " + name + " || " + side_label;
    ops.back().flop = side_flop;
    return y;
  }
  ops.push_back([=](hipStream_t s) {
    launch_groupnorm(p0, C0, p1, C1, partial, gamma, beta, yp, B, HW, G, eps, si, s, hook ? hook->consume() : 0);
  });
  ops.back().label = "groupnorm C=" + std::to_string(C) + " @" + std::to_string(x.H) + "x" + std::to_string(x.W) + " " + name;
  return y;
}

Tensor UNet::layer_norm(std::vector<Op>& ops, const std::string& name, const Tensor& x) {
  // LayerNormANE eps default 1e-5 (layer_norm.py:21); checkpoint bias used as x_hat*w + b
  const float* w = upload_vec(name + ".weight", x.C);
  const float* b = upload_vec(name + ".bias", x.C);
  Tensor y = new_tensor(x.B, x.H, x.W, x.C);
  const half_t* xp = x.p;
  half_t* yp = y.p;
  const int M = x.M(), C = x.C;
  ops.push_back([=](hipStream_t s) { launch_layernorm(xp, w, b, yp, M, C, 1e-5f, s); });
  ops.back().label = "layernorm C=" + std::to_string(C) + " M=" + std::to_string(M) + " " + name;
  return y;
}

const float* UNet::register_temb(const std::string& name, int cout) {
  SD_REQUIRE(temb_used_ + cout <= kTembCap, kInternal, "time_emb_proj outputs exceed %d", kTembCap);
  temb_layers_.push_back({name, cout});
  const float* p = temb_all_ + temb_used_;
  temb_used_ += cout;
  return p;
}

// unet.py:470-489
Tensor UNet::resnet(std::vector<Op>& ops, const std::string& p, const Tensor& x, const Tensor* x2, int cout,
                    bool has_temb) {
  const int cin = x.C + (x2 ? x2->C : 0);
  // conv_shortcut(x) does not depend on norm1 -> conv1 -> norm2: it rides in norm1's launch (one grid: the GroupNorm's blocks, then
  // the GEMM's tiles - igemm.hip gn_*_side_kernel) instead of a launch of its own.  SD_GN_SIDE=0 (with SD_TUNE): separate launches (A/B).
  static const int side_mode = tune_env_int("SD_GN_SIDE", 1);
  static const int twin_mode = tune_env_int("SD_GN_TWIN", 0);
  static const int twin_max_hw = tune_env_int("SD_GN_TWIN_MAX_HW", 256);
  const bool twin_here = twin_mode != 0 && x.H * x.W <= std::min(256, twin_max_hw);   // norm1 may become a twin of its producers
  const half_t* side_out = nullptr;
  Tensor t0;
  if (cin != cout && side_mode != 0 && !twin_here && !f32_) {
    ConvDesc sd;
    sd.x0 = x.p;
    sd.C0 = x.C;
    if (x2) {
      sd.x1 = x2->p;
      sd.C1 = x2->C;
    }
    sd.B = x.B;
    sd.Hi = sd.Ho = x.H;
    sd.Wi = sd.Wo = x.W;
    sd.N = cout;
    if (gn_side_gemm_ok(sd)) {
      sd.w = upload_conv_weight(p + ".conv_shortcut", cout, cin, 1, false);
      sd.bias = upload_vec(p + ".conv_shortcut.bias", cout);
      Tensor sc = new_tensor(x.B, x.H, x.W, cout);
      sd.out = sc.p;
      side_out = sc.p;
      char buf[256];
      snprintf(buf, sizeof(buf), "gemm1x1 %d->%d @%dx%d M=%d K=%d %s.conv_shortcut", cin, cout, x.H, x.W, x.M(), cin, p.c_str());
      t0 = group_norm(ops, p + ".norm1", x, x2, cfg_.norm_eps, true, &sd, buf, 2.0 * x.M() * (double)cout * cin);
    }
  }
  const float* temb = has_temb ? register_temb(p + ".time_emb_proj", cout) : nullptr;
  // norm1 -> SiLU -> conv1 and norm2 -> SiLU -> conv2 as ONE op each where the 3x3 kernel can normalise its input in the halo loader
  // (gn_loader_conv; SD_GN_LOADER with SD_TUNE).  norm1 keeps its own launch when it carries the shortcut GEMM or a concat.
  Tensor h;
  bool conv1_done = false;
  if (!side_out && !x2) {
    if (temb && temb_join_pos_ < 0 && &ops == &main_ops_) temb_join_pos_ = (int)ops.size();
    conv1_done = gn_loader_conv(ops, p + ".norm1", p + ".conv1", x, cout, temb, nullptr, &h);
  }
  if (!conv1_done) {
    if (!side_out) t0 = group_norm(ops, p + ".norm1", x, x2, cfg_.norm_eps, true);
    if (temb && temb_join_pos_ < 0 && &ops == &main_ops_) temb_join_pos_ = (int)ops.size();   // first consumer of the time path
    h = conv(ops, p + ".conv1", t0, nullptr, cout, 3, 1, 1, true, temb, nullptr);
  }
  const bool fuse2 = gn_loader_ok(ops, h, cout);          // norm2 inside conv2's loader
  Tensor t1;
  if (!fuse2) t1 = group_norm(ops, p + ".norm2", h, nullptr, cfg_.norm_eps, true);
  const half_t* shortcut;
  if (side_out) {
    shortcut = side_out;
  } else if (cin != cout) {
    Tensor sc = conv(ops, p + ".conv_shortcut", x, x2, cout, 1, 1, 1, true, nullptr, nullptr);
    shortcut = sc.p;
  } else {
    SD_REQUIRE(x2 == nullptr, kInternal, "%s: concat input with identity shortcut", p.c_str());
    shortcut = x.p;
  }
  if (fuse2) {
    Tensor out2;
    SD_REQUIRE(gn_loader_conv(ops, p + ".norm2", p + ".conv2", h, cout, nullptr, shortcut, &out2), kInternal, "%s: norm2 -> conv2", p.c_str());
    return out2;
  }
  return conv(ops, p + ".conv2", t1, nullptr, cout, 3, 1, 1, true, nullptr, shortcut);
}

bool UNet::gn_loader_ok(const std::vector<Op>& ops, const Tensor& x, int cout) const {
  static const int mode = tune_env_int("SD_GN_LOADER", 0);
  static const int min_hw = tune_env_int("SD_GN_LOADER_MIN_HW", 1024);
  const int G = cfg_.norm_num_groups, C = x.C, HW = x.H * x.W;
  if (mode == 0 || f32_ || HW < min_hw || C % G != 0 || (C / G) > 64) return false;
  if (!x.gn || x.gn->partial || x.gn->n_twins != 0 || x.gn->ops_list != &ops) return false;   // the raw tensor's producer must be able to leave statistics
  ConvDesc d;
  d.x0 = x.p;
  d.C0 = C;
  d.B = x.B;
  d.Hi = d.Ho = x.H;
  d.Wi = d.Wo = x.W;
  d.N = cout;
  d.ksize = 3;
  d.gnf_groups = G;
  return conv_fast_path_ok(d) && conv_gn_loader_ok(d);
}

// GroupNorm(+SiLU) -> 3x3 conv as one op: the conv's halo loader normalises (conv3x3_halo_ks_kernel GNL) when x's producer left
// the statistics; VERDICT r4 item 2c.  SD_GN_LOADER=1 (with SD_TUNE) switches it on for levels of at least SD_GN_LOADER_MIN_HW
// pixels (default 1024: the 64x64 / 32x32 levels); measured in LAB_NOTES.md r5.
bool UNet::gn_loader_conv(std::vector<Op>& ops, const std::string& norm, const std::string& cv, const Tensor& x, int cout, const float* temb,
                          const half_t* res, Tensor* out) {
  if (!gn_loader_ok(ops, x, cout)) return false;
  const int G = cfg_.norm_num_groups, C = x.C, HW = x.H * x.W;
  auto gi = std::make_shared<GnIn>();
  gi->hook = x.gn;
  gi->partial = arena_.alloc_n<float>(groupnorm_scratch_floats(x.B, HW, G));
  gi->hook->partial = gi->partial;
  gi->hook->groups = G;
  gi->gamma = upload_vec(norm + ".weight", C);
  gi->beta = upload_vec(norm + ".bias", C);
  gi->y = new_tensor(x.B, x.H, x.W, C).p;
  gi->eps = cfg_.norm_eps;
  gi->groups = G;
  gi->silu = 1;
  gi->norm_name = norm;
  pending_gn_in_ = gi;
  *out = conv(ops, cv, x, nullptr, cout, 3, 1, 1, true, temb, res);
  return true;
}

Tensor UNet::attention(std::vector<Op>& ops, const Tensor& q, const half_t* k, const half_t* vt, int heads, int Sq,
                       int Sk, int ldk, int ldv, int ldq, bool vt_perm, bool q_prescaled) {
  Tensor o = new_tensor(q.B, q.H, q.W, q.C);
  AttnDesc d;
  d.q = q.p;
  d.k = k;
  d.vt = vt;
  d.out = o.p;
  d.B = q.B;
  d.heads = heads;
  d.d = q.C / heads;
  d.Sq = Sq;
  d.Sk = Sk;
  d.ldq = ldq;
  d.ldk = ldk;
  d.ldv = ldv;
  d.ldo = q.C;
  d.vt_perm = vt_perm ? 1 : 0;
  d.q_prescaled = (vt_perm && q_prescaled) ? 1 : 0;
  SD_REQUIRE(attention_supported(d.d), kUnsupported, "head dim %d (C=%d, heads=%d) unsupported", d.d, q.C, heads);
  if (d.vt_perm) {   // attention8's balanced form wants scratch (one buffer for all of this handle's launches: they run in sequence)
    size_t pb = 0;
    int nc = 0;
    if (attention8_sk_scratch(d, &pb, &nc)) {
      if (pb > sk_part_bytes_) {
        sk_part_ = reinterpret_cast<float*>(arena_.alloc(pb));
        sk_part_bytes_ = pb;
      }
      if (nc > sk_cnt_n_) {
        sk_cnt_ = arena_.alloc_n<unsigned>(nc);
        sk_cnt_n_ = nc;
        SD_HIP(hipMemsetAsync(sk_cnt_, 0, (size_t)nc * sizeof(unsigned), stream_));
      }
      d.sk_part = sk_part_;
      d.sk_part_bytes = sk_part_bytes_;
      d.sk_cnt = sk_cnt_;
      d.sk_cnt_n = sk_cnt_n_;
    }
  }
  ops.push_back([this, d](hipStream_t s) {
    AttnDesc dd = d;
    dd.impl = cfg_.attention_impl;   // run-time switch (the reference's global, unet.py:39)
    launch_attention(dd, s);
  });
  ops.back().label = "attention h=" + std::to_string(heads) + " d=" + std::to_string(d.d) + " Sq=" + std::to_string(Sq) +
                     " Sk=" + std::to_string(Sk);
  ops.back().flop = 4.0 * q.B * (double)q.C * Sq * Sk;
  return o;
}

// unet.py:586-591 (+ CrossAttention :87-118, FeedForward/GEGLU :594-617)
Tensor UNet::transformer_block(std::vector<Op>& ops, const std::string& b, const Tensor& h, int heads, const std::string* proj_out,
                               const Tensor* tres, bool* tail_done, const PreQkv* pre) {
  const int C = h.C, S = h.H * h.W, L = cfg_.context_len;
  // --- self attention.  MFMA-tileable widths: norm1 is folded into ONE fused q|k|v GEMM whose V
  // columns leave token-transposed (attention's V^T operand); otherwise LN + stacked q|k + V^T GEMMs.
  const int ldv = round_up(S, 8);
  Tensor qk;
  bool vt_perm = false;   // attention8.hip will run this self-attention: the fused q|k|v GEMM writes V^T in its key order
  bool q_pre = false;     // ... and the queries pre-scaled
  half_t* vtp;
  if (pre) {   // the SpatialTransformer's head launch (UNet::transformer) already wrote this block's q | k and V^T
    qk = pre->qk;
    vtp = pre->vt;
    vt_perm = pre->vt_perm;
    q_pre = pre->q_pre;
  } else if (can_fold_ln(h, 3 * C, false) && S % 8 == 0 && (2 * C) % 64 == 0) {
    LnFold f = fold_layernorm(b + ".norm1", {b + ".attn1.to_q", b + ".attn1.to_k", b + ".attn1.to_v"}, C, C, false);
    ConvExtra ex;
    ex.ln_colsum = f.colsum;
    ex.n_trans = 2 * C;
    vt_perm = attention8_shape_ok(C / heads, S, S) && S % 16 == 0;
    ex.vt_perm = vt_perm;
    if (vt_perm) {   // attention8 takes the queries with d^-0.5 * log2(e) already in them: multiplied into the fp32 accumulator here
      ex.q_scale = attention_q_prescale(C / heads);
      ex.q_cols = C;
      q_pre = true;
    }
    qk = conv_w(ops, b + ".attn1.to_qkv", f.w, f.bias, h, nullptr, 3 * C, 1, 1, 1, nullptr, nullptr, kOutHalf, ldv,
                false, &ex);
    vtp = ex.vt;
  } else {
    Tensor n1 = layer_norm(ops, b + ".norm1", h);
    qk = conv_stacked(ops, {b + ".attn1.to_q", b + ".attn1.to_k"}, n1, C);   // [M][2C]: q | k
    vtp = conv(ops, b + ".attn1.to_v", n1, nullptr, C, 1, 1, 1, false, nullptr, nullptr, kOutHalfT, ldv).p;
  }
  Tensor q = qk;
  q.C = C;   // logical width of q; rows are 2C apart
  Tensor a1 = attention(ops, q, qk.p + C, vtp, heads, S, S, 2 * C, ldv, 2 * C, vt_perm, q_pre);
  // attn1.to_out + residual -> norm2 -> to_q -> cross-attention -> attn2.to_out + residual as ONE launch at the 5-head level
  // (xattn_out.hip: 32 tokens x all heads per workgroup; h1 never goes to HBM): three launches of 11.6 + 15.3 + 11.6 us become one of
  // 23.7, step 4.493 -> 4.443 ms (profiles/r05_xattn_out_stage2_ab.log).  SD_XATTN_OUT (with SD_TUNE) for the A/B: 0 = the three
  // launches, 1 = only the cross-attention branch in one launch (-0.024 ms), 2 = that at the 10-head level too (64 workgroups of ten
  // waves: +0.04 ms), 3 = the default
  static const int xo_mode = tune_env_int("SD_XATTN_OUT", 3);
  const bool xo_branch = xo_mode != 0 && !f32_ && can_fold_ln(h, C, false) && xattn_fused_ok(C, heads, S, L) && xattn_out_ok(C, heads, S, L) &&
                         (heads == 5 || xo_mode == 2);
  const bool xo_pre = xo_branch && xo_mode == 3 && heads == 5;
  Tensor h1;
  if (!xo_pre) h1 = conv(ops, b + ".attn1.to_out.0", a1, nullptr, C, 1, 1, 1, true, nullptr, h.p);
  // --- cross attention: K / V^T of the prompt are computed by ctx_ops_ when the prompt changes
  const int ldvc = round_up(L, 8);
  Tensor k2 = conv(ctx_ops_, b + ".attn2.to_k", ctx_, nullptr, C, 1, 1, 1, false, nullptr, nullptr);
  Tensor vt2 = conv(ctx_ops_, b + ".attn2.to_v", ctx_, nullptr, C, 1, 1, 1, false, nullptr, nullptr, kOutHalfT, ldvc);
  Tensor a2;
  Tensor h2;
  bool branch_done = false;
  if (xo_branch) {
    LnFold f = fold_layernorm(b + ".norm2", {b + ".attn2.to_q"}, C, C, false);
    half_t* wq_t = arena_.alloc_n<half_t>((size_t)C * C);
    launch_xattn_out_retile(f.w, wq_t, C, stream_);
    const half_t* wo = upload_conv_weight(b + ".attn2.to_out.0", C, C, 1, false);
    half_t* wo_t = arena_.alloc_n<half_t>((size_t)C * C);
    launch_xattn_out_retile(wo, wo_t, C, stream_);
    h2 = new_tensor(h.B, h.H, h.W, C);
    XAttnOutDesc xd;
    xd.x = xo_pre ? a1.p : h1.p;
    if (xo_pre) {
      const half_t* wo1 = upload_conv_weight(b + ".attn1.to_out.0", C, C, 1, false);
      half_t* wo1_t = arena_.alloc_n<half_t>((size_t)C * C);
      launch_xattn_out_retile(wo1, wo1_t, C, stream_);
      xd.h0 = h.p;
      xd.wo1_t = wo1_t;
      xd.o1_bias = upload_vec(b + ".attn1.to_out.0.bias", C);
    }
    xd.wq_t = wq_t;
    xd.q_bias = f.bias;
    xd.q_colsum = f.colsum;
    xd.k = k2.p;
    xd.vt = vt2.p;
    xd.wo_t = wo_t;
    xd.o_bias = upload_vec(b + ".attn2.to_out.0.bias", C);
    xd.out = h2.p;
    xd.M = h.M();
    xd.C = C;
    xd.S = S;
    xd.L = L;
    xd.ldv = ldvc;
    xd.heads = heads;
    ops.push_back([this, xd](hipStream_t s) {
      XAttnOutDesc dd = xd;
      dd.impl = cfg_.attention_impl;
      launch_xattn_out(dd, s);
    });
    ops.back().label = std::string("xattn") + (xo_pre ? " attn1.to_out + residual +" : "") + "+ln q-proj " + std::to_string(C) + "->" + std::to_string(C) +
                       " + attention h=" + std::to_string(heads) + " d=64 Sq=" + std::to_string(S) + " Sk=" + std::to_string(L) +
                       " + to_out + residual " + b + ".attn2";
    ops.back().flop = (xo_pre ? 6.0 : 4.0) * h.M() * (double)C * C + 4.0 * h.B * (double)C * S * L;
    branch_done = true;
  } else if (can_fold_ln(h1, C, false) && xattn_fused_ok(C, heads, S, L)) {
    // norm2 -> to_q -> softmax(q k^T) v as ONE launch (xattn.hip): the q tile stays in registers
    LnFold f = fold_layernorm(b + ".norm2", {b + ".attn2.to_q"}, C, C, false);
    a2 = new_tensor(h1.B, h1.H, h1.W, C);
    XAttnDesc xd;
    xd.x = h1.p;
    xd.wq = f.w;
    xd.bias = f.bias;
    xd.colsum = f.colsum;
    xd.k = k2.p;
    xd.vt = vt2.p;
    xd.out = a2.p;
    xd.M = h1.M();
    xd.C = C;
    xd.S = S;
    xd.L = L;
    xd.ldv = ldvc;
    xd.heads = heads;
    ops.push_back([this, xd](hipStream_t s) {
      XAttnDesc dd = xd;
      dd.impl = cfg_.attention_impl;
      launch_xattn_fused(dd, s);
    });
    ops.back().label = "xattn+ln q-proj " + std::to_string(C) + "->" + std::to_string(C) + " + attention h=" + std::to_string(heads) +
                       " d=64 Sq=" + std::to_string(S) + " Sk=" + std::to_string(L) + " " + b + ".attn2";
    ops.back().flop = 2.0 * h1.M() * (double)C * C + 4.0 * h1.B * (double)C * S * L;
  } else {
    Tensor q2;
    if (can_fold_ln(h1, C, false)) {
      LnFold f = fold_layernorm(b + ".norm2", {b + ".attn2.to_q"}, C, C, false);
      ConvExtra ex;
      ex.ln_colsum = f.colsum;
      q2 = conv_w(ops, b + ".attn2.to_q", f.w, f.bias, h1, nullptr, C, 1, 1, 1, nullptr, nullptr, kOutHalf, 0, false, &ex);
    } else {
      Tensor n2 = layer_norm(ops, b + ".norm2", h1);
      q2 = conv(ops, b + ".attn2.to_q", n2, nullptr, C, 1, 1, 1, false, nullptr, nullptr);
    }
    a2 = attention(ops, q2, k2.p, vt2.p, heads, S, L, C, ldvc, C);
  }
  if (!branch_done) h2 = conv(ops, b + ".attn2.to_out.0", a2, nullptr, C, 1, 1, 1, true, nullptr, h1.p);
  // --- GEGLU feed-forward (norm3 folded the same way)
  Tensor g;
  if (can_fold_ln(h2, 8 * C, true)) {
    LnFold f = fold_layernorm(b + ".norm3", {b + ".ff.net.0.proj"}, C, 8 * C, true);
    ConvExtra ex;
    ex.ln_colsum = f.colsum;
    g = conv_w(ops, b + ".ff.net.0.proj", f.w, f.bias, h2, nullptr, 8 * C, 1, 1, 1, nullptr, nullptr, kOutGeglu, 0,
               false, &ex);
  } else {
    Tensor n3 = layer_norm(ops, b + ".norm3", h2);
    g = conv(ops, b + ".ff.net.0.proj", n3, nullptr, 8 * C, 1, 1, 1, true, nullptr, nullptr, kOutGeglu);
  }
  // the tail of the SpatialTransformer - ff.net.2 + residual -> proj_out + residual - as ONE launch at the 320-channel level
  // (ffn_proj_kernel: the intermediate never goes to HBM, the output's GroupNorm statistics for the next resnet come out of it);
  // five launches less per step, 5.402 -> 5.392 ms on the (slow) box it was measured on (profiles/r05_ffn_proj_ab_slow_box.log);
  // SD_FFN_PROJ=0 (with SD_TUNE): the two GEMM launches (A/B)
  static const int fp_mode = tune_env_int("SD_FFN_PROJ", 1);
  if (proj_out && tres && tail_done && fp_mode != 0 && !f32_ && g.C == 4 * C && ffn_proj_ok(C, 4 * C, h.M(), S)) {
    const half_t* w1 = upload_conv_weight(b + ".ff.net.2", C, 4 * C, 1, false);
    half_t* w1_t = arena_.alloc_n<half_t>((size_t)C * 4 * C);
    launch_xattn_out_retile_nk(w1, w1_t, C, 4 * C, stream_);
    const half_t* w2 = upload_conv_weight(*proj_out, C, C, 1, false);
    half_t* w2_t = arena_.alloc_n<half_t>((size_t)C * C);
    launch_xattn_out_retile_nk(w2, w2_t, C, C, stream_);
    Tensor out = new_tensor(h.B, h.H, h.W, C);
    FfnProjDesc d;
    d.g = g.p;
    d.w1_t = w1_t;
    d.b1 = upload_vec(b + ".ff.net.2.bias", C);
    d.res1 = h2.p;
    d.w2_t = w2_t;
    d.b2 = upload_vec(*proj_out + ".bias", C);
    d.res2 = tres->p;
    d.out = out.p;
    d.M = h.M();
    d.C = C;
    d.K1 = 4 * C;
    d.S = S;
    auto hook = std::make_shared<GnHook>();   // a GroupNorm built later over this tensor may ask for its statistics (as conv_w)
    out.gn = hook;
    hook->ops_pos = (int)ops.size();
    hook->ops_list = &ops;
    ops.push_back([d, hook](hipStream_t s) {
      FfnProjDesc dd = d;
      dd.gn_partial = hook->partial;
      dd.gn_groups = hook->groups;
      hook->produced(launch_ffn_proj(dd, s));
    });
    char buf[256];
    snprintf(buf, sizeof(buf), "gemm1x1 %d->%d @%dx%d M=%d K=%d %s.ff.net.2 + residual + proj_out + residual", 4 * C, C, h.H, h.W, h.M(), 4 * C, b.c_str());
    ops.back().label = buf;
    ops.back().flop = 2.0 * h.M() * (double)C * (4 * C) + 2.0 * h.M() * (double)C * C;
    *tail_done = true;
    return out;
  }
  return conv(ops, b + ".ff.net.2", g, nullptr, C, 1, 1, 1, true, nullptr, h2.p);
}

// unet.py:553-563; GroupNorm eps hard-coded 1e-6 (unet.py:528-531)
Tensor UNet::transformer(std::vector<Op>& ops, const std::string& p, const Tensor& x, int heads, int depth) {
  // norm -> proj_in as ONE launch where x's producer can leave the GroupNorm statistics (GnHook): the 1x1 GEMM folds them into a
  // per-channel scale / shift of its activation fragments (gemm_pipe_kernel GNF).  When the producer's plan wrote none (split-K)
  // the op falls back, at launch time, to the GroupNorm launch + the plain GEMM.
  // MEASURED AND NOT ADOPTED (LAB_NOTES.md r5): the folded GEMM (table build behind two barriers in its prologue, three table reads
  // and eight packed ops per activation fragment) takes 17.4 / 13.9 us where the plain one takes 9.2 / 10.0 - what the 7-8 us
  // GroupNorm launch cost; the step is unchanged on a fast box (4.40 ms both ways) and 0.1 ms SLOWER on a slow one (5.43 vs 5.33).
  // Off unless SD_GN_FOLD=1 (with SD_TUNE); the kernel stays tested at operator level (tests/test_round5_gpu.py).
  static const int fold_mode = tune_env_int("SD_GN_FOLD", 0);
  const int G = cfg_.norm_num_groups, C = x.C, HW = x.H * x.W;
  ConvDesc d;
  d.x0 = x.p;
  d.C0 = C;
  d.B = x.B;
  d.Hi = d.Ho = x.H;
  d.Wi = d.Wo = x.W;
  d.N = C;
  const bool fold = fold_mode != 0 && !f32_ && x.gn && !x.gn->partial && x.gn->n_twins == 0 && x.gn->ops_list == &ops && C % G == 0 && G <= 32 &&
                    C <= 2048 && HW % 64 == 0 && conv_fast_path_ok(d) && (size_t)x.M() * C * 2 < ((size_t)1 << 31);
  // norm -> proj_in -> norm1 -> to_q | to_k | to_v of block 0 as ONE launch at the 5-head level (xattn_out.hip gn_proj_qkv_kernel: rows are
  // independent once the GroupNorm statistics exist, so a 32-token workgroup carries its rows through both GEMMs in LDS).
  // Three launches of 10 + 9.5 + 15 us become one of 24.5; same box, back to back: 4.415 / 4.430 -> 4.344 / 4.368 ms, eight prompts
  // 18.73 -> 18.45 ms (profiles/r06_gn_qkv_ab.txt).  SD_GN_QKV=0 (with SD_TUNE): the three launches.
  static const int gq_mode = tune_env_int("SD_GN_QKV", SD_GN_QKV_DEFAULT);
  const int ldv0 = round_up(HW, 8);
  const bool gq = gq_mode != 0 && !fold && !f32_ && depth >= 1 && x.gn && !x.gn->partial && x.gn->n_twins == 0 && x.gn->ops_list == &ops &&
                  C % G == 0 && can_fold_ln(x, 3 * C, false) && gn_proj_qkv_ok(C, heads, HW, x.M(), ldv0, G) &&
                  (size_t)x.M() * C * 2 < ((size_t)1 << 31);
  Tensor h;
  PreQkv pre;
  if (gq) {
    const std::string b0 = p + ".transformer_blocks.0";
    float* partial = arena_.alloc_n<float>(groupnorm_scratch_floats(x.B, HW, G));
    std::shared_ptr<GnHook> hook = x.gn;
    hook->partial = partial;
    hook->groups = G;
    Tensor t0 = new_tensor(x.B, x.H, x.W, C);   // the normalised tensor of the fall-back (producer left no statistics)
    h = new_tensor(x.B, x.H, x.W, C);
    const half_t* wp = upload_conv_weight(p + ".proj_in", C, C, 1, false);
    half_t* wp_t = arena_.alloc_n<half_t>((size_t)C * C);
    launch_xattn_out_retile_nk(wp, wp_t, C, C, stream_);
    LnFold f = fold_layernorm(b0 + ".norm1", {b0 + ".attn1.to_q", b0 + ".attn1.to_k", b0 + ".attn1.to_v"}, C, C, false);
    half_t* wq_t = arena_.alloc_n<half_t>((size_t)3 * C * C);
    launch_xattn_out_retile_nk(f.w, wq_t, 3 * C, C, stream_);
    pre.qk = new_tensor(x.B, x.H, x.W, 2 * C);
    pre.vt = arena_.alloc_n<half_t>((size_t)x.B * C * ldv0);
    pre.vt_perm = attention8_shape_ok(C / heads, HW, HW) && HW % 16 == 0;
    pre.q_pre = pre.vt_perm;
    GnProjQkvDesc g;
    g.x = x.p;
    g.gn_partial = partial;
    g.gn_gamma = upload_vec(p + ".norm.weight", C);
    g.gn_beta = upload_vec(p + ".norm.bias", C);
    g.gn_groups = G;
    g.gn_eps = 1e-6f;
    g.wp_t = wp_t;
    g.p_bias = upload_vec(p + ".proj_in.bias", C);
    g.h = h.p;
    g.wqkv_t = wq_t;
    g.qkv_bias = f.bias;
    g.qkv_colsum = f.colsum;
    g.ln_eps = 1e-5f;
    g.qk = pre.qk.p;
    g.vt = pre.vt;
    g.M = x.M();
    g.C = C;
    g.S = HW;
    g.ldT = ldv0;
    g.vt_perm = pre.vt_perm;
    g.q_scale = pre.q_pre ? attention_q_prescale(C / heads) : 1.f;
    const int B = x.B;
    ops.push_back([g, hook, t0, B, HW, G, C](hipStream_t s) {
      GnProjQkvDesc gg = g;
      const int n = hook->consume();
      if (n >= 1 && n <= 128) {
        gg.gn_entries = n;
      } else {
        launch_groupnorm(g.x, C, nullptr, 0, const_cast<float*>(g.gn_partial), g.gn_gamma, g.gn_beta, t0.p, B, HW, G, g.gn_eps, 0, s, n);
        gg.x = t0.p;
        gg.gn_entries = 0;
      }
      launch_gn_proj_qkv(gg, s);
    });
    char buf[256];
    snprintf(buf, sizeof(buf), "gemm1x1+gn %d->%d @%dx%d M=%d K=%d %s.proj_in + norm1 + to_qkv %d->%d", C, C, x.H, x.W, x.M(), C, p.c_str(), C, 3 * C);
    ops.back().label = buf;
    ops.back().flop = 8.0 * x.M() * (double)C * C;
  } else if (fold) {
    float* partial = arena_.alloc_n<float>(groupnorm_scratch_floats(x.B, HW, G));
    const float* gamma = upload_vec(p + ".norm.weight", C);
    const float* beta = upload_vec(p + ".norm.bias", C);
    std::shared_ptr<GnHook> hook = x.gn;
    hook->partial = partial;
    hook->groups = G;
    Tensor t0 = new_tensor(x.B, x.H, x.W, C);   // the normalised tensor of the fallback path
    h = new_tensor(x.B, x.H, x.W, C);
    d.w = upload_conv_weight(p + ".proj_in", C, C, 1, false);
    d.bias = upload_vec(p + ".proj_in.bias", C);
    d.out = h.p;
    ws_need_ = std::max(ws_need_, conv_workspace_bytes(d));
    const int B = x.B;
    const half_t* xp = x.p;
    ops.push_back([this, d, hook, partial, gamma, beta, t0, xp, B, HW, G, C](hipStream_t s) {
      const int n = hook->consume();
      if (n >= 1 && n <= 128) {
        ConvDesc dd = d;
        dd.gnf_partial = partial;
        dd.gnf_gamma = gamma;
        dd.gnf_beta = beta;
        dd.gnf_eps = 1e-6f;
        dd.gnf_groups = G;
        dd.gnf_entries = n;
        launch_conv(dd, ws_conv_, s);
      } else {
        launch_groupnorm(xp, C, nullptr, 0, partial, gamma, beta, t0.p, B, HW, G, 1e-6f, 0, s, n);
        ConvDesc dd = d;
        dd.x0 = t0.p;
        launch_conv(dd, ws_conv_, s);
      }
    });
    char buf[256];
    snprintf(buf, sizeof(buf), "gemm1x1+gn %d->%d @%dx%d M=%d K=%d %s.proj_in #0,1,1,1,%d,%d,%d", C, C, x.H, x.W, x.M(), C, p.c_str(), C, C, x.M());
    ops.back().label = buf;
    ops.back().flop = 2.0 * x.M() * (double)C * C;
  } else {
    Tensor t0 = group_norm(ops, p + ".norm", x, nullptr, 1e-6f, false);
    h = conv(ops, p + ".proj_in", t0, nullptr, x.C, 1, 1, 1, true, nullptr, nullptr);
  }
  bool tail_done = false;
  const std::string proj_name = p + ".proj_out";
  for (int d = 0; d < depth; ++d) {
    const bool last = d == depth - 1;
    h = transformer_block(ops, p + ".transformer_blocks." + std::to_string(d), h, heads, last ? &proj_name : nullptr, last ? &x : nullptr,
                          last ? &tail_done : nullptr, (gq && d == 0) ? &pre : nullptr);
  }
  if (tail_done) return h;
  return conv(ops, p + ".proj_out", h, nullptr, x.C, 1, 1, 1, true, nullptr, x.p);
}

// down blocks (unet.py:336-350, :389-403) + mid block (unet.py:789-795)
void UNet::down_and_mid(std::vector<Op>& ops, Tensor& h, std::vector<Tensor>& skips) {
  const int n = cfg_.n_levels;
  skips.push_back(h);
  for (int i = 0; i < n; ++i) {
    const int cout = cfg_.block_out_channels[i];
    for (int j = 0; j < cfg_.layers_per_block; ++j) {
      const std::string p = "down_blocks." + std::to_string(i);
      h = resnet(ops, p + ".resnets." + std::to_string(j), h, nullptr, cout);
      if (cfg_.down_cross_attn[i])
        h = transformer(ops, p + ".attentions." + std::to_string(j), h, cfg_.attention_head_dim[i],
                        cfg_.transformer_layers_per_block[i]);
      skips.push_back(h);
    }
    if (i != n - 1) {   // Downsample2D: conv 3x3 stride 2 pad 1 (unet.py:503-510)
      h = conv(ops, "down_blocks." + std::to_string(i) + ".downsamplers.0.conv", h, nullptr, cout, 3, 2, 1, true,
               nullptr, nullptr);
      skips.push_back(h);
    }
  }
  const int c = cfg_.block_out_channels[n - 1];
  h = resnet(ops, "mid_block.resnets.0", h, nullptr, c);
  // the reference passes attention_head_dim[i] with the leaked loop index == last level (unet.py:929)
  h = transformer(ops, "mid_block.attentions.0", h, cfg_.attention_head_dim[n - 1],
                  cfg_.transformer_layers_per_block[n - 1]);
  h = resnet(ops, "mid_block.resnets.1", h, nullptr, c);
}

void UNet::finalize_temb() {
  // one [sum Cout][temb_dim] matrix for all time_emb_proj layers (unet.py:442-444, :477)
  const int tdim = cfg_.block_out_channels[0] * 4;
  std::vector<half_t> w((size_t)temb_used_ * tdim);
  std::vector<float> b(temb_used_);
  size_t row = 0;
  for (auto& [name, cout] : temb_layers_) {
    const HostTensor& tw = ws_->get(name + ".weight");
    const HostTensor& tb = ws_->get(name + ".bias");
    SD_REQUIRE(tw.numel() == (size_t)cout * tdim && (int)tb.numel() == cout, kInvalidArgument, "%s: bad shape",
               name.c_str());
    for (size_t i = 0; i < (size_t)cout * tdim; ++i) w[row * tdim + i] = (half_t)tw.data[i];
    for (int i = 0; i < cout; ++i) b[row + i] = tb.data[i];
    row += cout;
  }
  temb_w_all_ = arena_.alloc_n<half_t>(w.size());
  temb_b_all_ = arena_.alloc_n<float>(b.size());
  SD_HIP(hipMemcpy(temb_w_all_, w.data(), w.size() * sizeof(half_t), hipMemcpyHostToDevice));
  SD_HIP(hipMemcpy(temb_b_all_, b.data(), b.size() * sizeof(float), hipMemcpyHostToDevice));
}

void UNet::build_unet() {
  const int B = cfg_.batch, H = cfg_.height, W = cfg_.width, n = cfg_.n_levels;
  const int C0 = cfg_.block_out_channels[0], tdim = C0 * 4, L = cfg_.context_len;
  const bool xl = cfg_.addition_time_embed_dim > 0;
  const bool cn = cfg_.is_controlnet != 0;

  in_sample_ = arena_.alloc_n<half_t>((size_t)B * cfg_.in_channels * H * W);
  in_timestep_ = arena_.alloc_n<half_t>(B);
  in_ehs_ = arena_.alloc_n<half_t>((size_t)B * cfg_.cross_attention_dim * L);
  tbuf_ = arena_.alloc_n<float>(B);
  emb_ = arena_.alloc_n<float>((size_t)B * tdim);
  temb_all_ = arena_.alloc_n<float>((size_t)B * kTembCap);
  x_in_ = new_tensor(B, H, W, cfg_.in_channels);
  ctx_ = new_tensor(B, 1, L, cfg_.cross_attention_dim);

  // ---- input conversions (boundary layouts -> kernel layouts) ----
  {
    half_t* ts = in_timestep_;
    float* tb = tbuf_;
    half_t* smp = in_sample_;
    Tensor xin = x_in_;
    in_ops_.push_back([=](hipStream_t s) {
      launch_half_to_float(ts, tb, B, s);
      launch_nchw_to_nhwc(smp, 0, xin.p, B, xin.C, xin.H, xin.W, s);
    });
    in_ops_.back().label = "boundary: timestep f16->f32, sample NCHW->NHWC";
    half_t* ehs = in_ehs_;
    Tensor ctx = ctx_;
    ctx_ops_.push_back([=](hipStream_t s) { launch_bc1s_to_tokens(ehs, ctx.p, B, ctx.C, ctx.W, s); });
  }

  // ---- time embedding (unet.py:983-984; XL :1076-1088) ----
  {
    auto upload_freq = [&](int dim) {
      std::vector<float> f = timestep_freq_table(dim, cfg_.freq_shift);
      float* d = arena_.alloc_n<float>(f.size());
      SD_HIP(hipMemcpy(d, f.data(), f.size() * sizeof(float), hipMemcpyHostToDevice));
      return d;
    };
    const float* freq = upload_freq(C0);
    float* t_emb = arena_.alloc_n<float>((size_t)B * C0);
    float* e1 = arena_.alloc_n<float>((size_t)B * tdim);
    half_t* w1 = upload_conv_weight("time_embedding.linear_1", tdim, C0, 1, false);
    float* b1 = upload_vec("time_embedding.linear_1.bias", tdim);
    half_t* w2 = upload_conv_weight("time_embedding.linear_2", tdim, tdim, 1, false);
    float* b2 = upload_vec("time_embedding.linear_2.bias", tdim);
    float* tb = tbuf_;
    float* emb = emb_;
    SD_REQUIRE(cfg_.flip_sin_to_cos == 1, kUnsupported, "flip_sin_to_cos=False is not on the path");
    time_ops_.push_back([=](hipStream_t s) {
      launch_timestep_embedding(tb, freq, t_emb, B, C0, s);
      launch_gemv(w1, b1, t_emb, C0, e1, tdim, B, tdim, C0, 0, 1, 0, s);
      launch_gemv(w2, b2, e1, tdim, emb, tdim, B, tdim, tdim, 0, 0, 0, s);
    });
    time_ops_.back().label = "time embedding: sinusoid + 2 GEMV (time_embedding.linear_1/2)";
    tpath_ = TimePath{freq, w1, w2, b1, b2, C0, tdim, !xl};
    if (xl) {
      const int nt = cfg_.num_time_ids, adim = cfg_.addition_time_embed_dim;
      const int pin = cfg_.projection_class_embeddings_input_dim;
      const int text_dim = pin - nt * adim;
      SD_REQUIRE(text_dim > 0, kInvalidArgument, "projection_class_embeddings_input_dim too small");
      in_time_ids_ = arena_.alloc_n<half_t>((size_t)B * nt);
      in_text_embeds_ = arena_.alloc_n<half_t>((size_t)B * text_dim);
      float* tid_f = arena_.alloc_n<float>((size_t)B * nt);
      float* add_in = arena_.alloc_n<float>((size_t)B * pin);   // [text_embeds | time_embeds]
      float* te = arena_.alloc_n<float>((size_t)B * nt * adim);
      float* a1 = arena_.alloc_n<float>((size_t)B * tdim);
      half_t* aw1 = upload_conv_weight("add_embedding.linear_1", tdim, pin, 1, false);
      float* ab1 = upload_vec("add_embedding.linear_1.bias", tdim);
      half_t* aw2 = upload_conv_weight("add_embedding.linear_2", tdim, tdim, 1, false);
      float* ab2 = upload_vec("add_embedding.linear_2.bias", tdim);
      const float* afreq = upload_freq(adim);
      half_t* tids = in_time_ids_;
      half_t* txt = in_text_embeds_;
      time_ops_.push_back([=](hipStream_t s) {
        launch_half_to_float(tids, tid_f, (size_t)B * nt, s);
        launch_timestep_embedding(tid_f, afreq, te, B * nt, adim, s);   // time_ids.flatten() (unet.py:1079)
        for (int b = 0; b < B; ++b) {   // concat([text_embeds, time_embeds], dim=-1) (unet.py:1082)
          launch_half_to_float(txt + (size_t)b * text_dim, add_in + (size_t)b * pin, text_dim, s);
          SD_HIP(hipMemcpyAsync(add_in + (size_t)b * pin + text_dim, te + (size_t)b * nt * adim,
                                (size_t)nt * adim * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
        launch_gemv(aw1, ab1, add_in, pin, a1, tdim, B, tdim, pin, 0, 1, 0, s);
        launch_gemv(aw2, ab2, a1, tdim, emb, tdim, B, tdim, tdim, 0, 0, 1, s);   // emb += aug_emb (:1090)
      });
      time_ops_.back().label = "text_time add-embedding: sinusoid + concat + 2 GEMV";
    }
  }

  // ---- conv_in (unet.py:991) [+ ControlNet conditioning embedding, controlnet.py:211-215] ----
  Tensor h;
  if (cn) {
    in_cond_ = arena_.alloc_n<half_t>((size_t)B * 3 * H * 8 * W * 8);
    Tensor c = new_tensor(B, H * 8, W * 8, 3);
    {
      half_t* src = in_cond_;
      Tensor cc = c;
      cond_ops_.push_back([=](hipStream_t s) { launch_nchw_to_nhwc(src, 0, cc.p, B, 3, cc.H, cc.W, s); });
    }
    const std::string p = "controlnet_cond_embedding";
    int ch = (int)ws_->get(p + ".conv_in.weight").shape[0];
    Tensor e = conv(cond_ops_, p + ".conv_in", c, nullptr, ch, 3, 1, 1, true, nullptr, nullptr, kOutHalf, 0, true);
    for (int i = 0; ws_->has(p + ".blocks." + std::to_string(i) + ".weight"); ++i) {
      const int co = (int)ws_->get(p + ".blocks." + std::to_string(i) + ".weight").shape[0];
      e = conv(cond_ops_, p + ".blocks." + std::to_string(i), e, nullptr, co, 3, 1 + (i % 2), 1, true, nullptr,
               nullptr, kOutHalf, 0, true);
    }
    e = conv(cond_ops_, p + ".conv_out", e, nullptr, C0, 3, 1, 1, true, nullptr, nullptr);
    SD_REQUIRE(e.H == H && e.W == W, kInvalidArgument, "controlnet_cond must be 8x the latent size");
    h = conv(main_ops_, "conv_in", x_in_, nullptr, C0, 3, 1, 1, true, nullptr, e.p);   // sample += cond
  } else {
    h = conv(main_ops_, "conv_in", x_in_, nullptr, C0, 3, 1, 1, true, nullptr, nullptr);
  }

  std::vector<Tensor> skips;
  down_and_mid(main_ops_, h, skips);

  // residual shapes (controlnet.py:191-197): one per skip tensor + mid
  for (auto& t : skips) res_shapes_.push_back({t.B, t.C, t.H, t.W});
  res_shapes_.push_back({h.B, h.C, h.H, h.W});

  if (cn) {
    // zero-conv taps (controlnet.py:236-248)
    for (size_t i = 0; i < skips.size(); ++i)
      cn_out_.push_back(conv(main_ops_, "controlnet_down_blocks." + std::to_string(i), skips[i], nullptr, skips[i].C,
                             1, 1, 1, true, nullptr, nullptr));
    cn_out_.push_back(conv(main_ops_, "controlnet_mid_block", h, nullptr, h.C, 1, 1, 1, true, nullptr, nullptr));
    for (auto& t : cn_out_) {
      float* o = arena_.alloc_n<float>(t.numel());
      res_out_.push_back(o);
      Tensor tt = t;
      out_ops_.push_back([=](hipStream_t s) { launch_nhwc_to_nchw_f32(tt.p, o, tt.B, tt.C, tt.H, tt.W, s); });
    }
    finalize_temb();
  } else {
    if (cfg_.support_controlnet) {   // unet.py:1009-1022
      std::vector<Tensor> added;
      for (size_t i = 0; i < res_shapes_.size(); ++i) {
        const auto& sh = res_shapes_[i];
        half_t* src = arena_.alloc_n<half_t>((size_t)sh[0] * sh[1] * sh[2] * sh[3]);
        in_res_nchw_.push_back(src);
        Tensor r = new_tensor(sh[0], sh[2], sh[3], sh[1]);
        const Tensor base = (i + 1 == res_shapes_.size()) ? h : skips[i];
        Tensor sum = new_tensor(sh[0], sh[2], sh[3], sh[1]);
        main_ops_.push_back([=](hipStream_t s) {
          if (attached_.empty()) {   // residuals arrive through the host boundary like the reference (pipeline.py:519-529)
            launch_nchw_to_nhwc(src, 0, r.p, r.B, r.C, r.H, r.W, s);
            launch_add_half(base.p, r.p, sum.p, r.numel(), s);
          } else {                   // device hand-off: skip[i] + sum over ControlNets of residual[i], one launch
            const half_t* srcs[4] = {base.p, nullptr, nullptr, nullptr};
            for (size_t k = 0; k < attached_.size(); ++k) srcs[1 + k] = attached_[k]->cn_out_[i].p;
            launch_sum_half(srcs, 1 + (int)attached_.size(), sum.p, r.numel(), s);
          }
        });
        main_ops_.back().label = "controlnet residual add " + std::to_string(i);
        if (cn_join_pos_ < 0) cn_join_pos_ = (int)main_ops_.size() - 1;   // first consumer of the ControlNets' outputs
        added.push_back(sum);
      }
      for (size_t i = 0; i < skips.size(); ++i) skips[i] = added[i];
      h = added.back();
    }
    // ---- up blocks (unet.py:207-225, :266-279, :1025-1041) ----
    for (int i = 0; i < n; ++i) {
      const int lvl = n - 1 - i;
      const int cout = cfg_.block_out_channels[lvl];
      const std::string p = "up_blocks." + std::to_string(i);
      for (int j = 0; j < cfg_.layers_per_block + 1; ++j) {
        Tensor skip = skips.back();
        skips.pop_back();
        h = resnet(main_ops_, p + ".resnets." + std::to_string(j), h, &skip, cout);
        if (cfg_.up_cross_attn[i])
          h = transformer(main_ops_, p + ".attentions." + std::to_string(j), h, cfg_.attention_head_dim[lvl],
                          cfg_.transformer_layers_per_block[lvl]);
      }
      if (i != n - 1)   // Upsample2D: nearest x2 fused into the conv's gather (unet.py:498-500)
        h = conv(main_ops_, p + ".upsamplers.0.conv", h, nullptr, cout, 3, 1, 2, true, nullptr, nullptr);
    }
    SD_REQUIRE(skips.empty(), kInternal, "skip stack not empty");
    // ---- conv_norm_out -> SiLU -> conv_out (unet.py:1044-1046), fp32 NCHW at the boundary ----
    Tensor t = group_norm(main_ops_, "conv_norm_out", h, nullptr, cfg_.norm_eps, true);
    noise_pred_ = arena_.alloc_n<float>((size_t)B * cfg_.out_channels * H * W);
    ConvDesc d;
    d.x0 = t.p;
    d.C0 = t.C;
    d.w = upload_conv_weight("conv_out", cfg_.out_channels, t.C, 3, false);
    d.bias = upload_vec("conv_out.bias", cfg_.out_channels);
    d.B = B; d.Hi = H; d.Wi = W; d.Ho = H; d.Wo = W;
    d.ksize = 3; d.stride = 1; d.up = 1; d.N = cfg_.out_channels;
    float* np = noise_pred_;
    if (cfg_.out_channels <= 8 && t.C % 8 == 0) {
      main_ops_.push_back([=](hipStream_t s) { launch_conv_small_n(d, np, s); });
      main_ops_.back().label = "conv3x3 small-N conv_out -> fp32 NCHW";
      main_ops_.back().flop = 2.0 * B * H * W * (double)cfg_.out_channels * t.C * 9;
    } else {
      Tensor o = new_tensor(B, H, W, cfg_.out_channels);
      ConvDesc d2 = d;
      d2.out = o.p;
      main_ops_.push_back([=](hipStream_t s) {
        launch_conv_generic(d2, 0, s);
        launch_nhwc_to_nchw_f32(o.p, np, o.B, o.C, o.H, o.W, s);
      });
    }
    finalize_temb();
  }

  // batched time_emb_proj(SiLU(emb)) for every resnet: one GEMV per step
  {
    half_t* w = temb_w_all_;
    float* b = temb_b_all_;
    float* emb = emb_;
    float* out = temb_all_;
    const int N = temb_used_;
    time_ops_.push_back([=](hipStream_t s) { launch_gemv(w, b, emb, tdim, out, kTembCap, B, N, tdim, 1, 0, 0, s); });
    time_ops_.back().label = "time_emb_proj of all resnets: one batched GEMV N=" + std::to_string(N);
  }
  if (ws_need_ > 0) {
    ws_conv_.partial = reinterpret_cast<float*>(arena_.alloc(ws_need_));
    ws_conv_.partial_bytes = ws_need_;
  }
  // loop state (allocated lazily on first denoise_loop)
}


// ---------------------------------------------------------------------------------------------
// VAE decoder (diffusers AutoencoderKL.decode, wrapped by the reference in
// torch2coreml.py:584-594: image = decoder(post_quant_conv(z)); called from pipeline.py:313-320).
// Third-party arithmetic restated from the public architecture:
//   post_quant_conv 1x1 -> conv_in 3x3 -> mid [ResNet, 1-head self-attention, ResNet]
//   -> up blocks (3 ResNets each, nearest-x2 + conv3x3 after all but the last)
//   -> GroupNorm(32, 1e-6) -> SiLU -> conv_out 3x3.   All GroupNorm eps = 1e-6, no time embedding.
// Reuses the UNet config struct: block_out_channels = decoder channels in ENCODER order
// (SD: 128,256,512,512), layers_per_block = 2 (decoder uses +1), in_channels = latent channels,
// out_channels = 3, height/width = latent size.
// ---------------------------------------------------------------------------------------------
// AutoencoderKL mid-block attention: single head over H*W tokens, d = C (too wide for the streaming kernel's
// register budget): scores and P are materialised per image (S x S fp16) through the GEMM kernel.
Tensor UNet::vae_attention(std::vector<Op>& ops, const std::string& p, const Tensor& h) {
  const int B = h.B, H = h.H, W = h.W, C = h.C, S = H * W;
  Tensor t0 = group_norm(ops, p + ".group_norm", h, nullptr, 1e-6f, false);
  Tensor q = conv(ops, p + ".to_q", t0, nullptr, C, 1, 1, 1, true, nullptr, nullptr);
  Tensor k = conv(ops, p + ".to_k", t0, nullptr, C, 1, 1, 1, true, nullptr, nullptr);
  if (f32_) {   // the same three steps on the fp32 kernels: K tokens / V play the weight matrices
    Tensor v = conv(ops, p + ".to_v", t0, nullptr, C, 1, 1, 1, true, nullptr, nullptr);
    float* scores = arena_.alloc_n<float>((size_t)S * S);
    Tensor a = new_tensor(B, H, W, C);
    const float scale = 1.0f / std::sqrt((float)C);
    for (int b = 0; b < B; ++b) {
      ConvF32Desc d1;   // scores[q][k] = sum_c Q[q][c] K[k][c]
      d1.x = reinterpret_cast<const float*>(q.p) + (size_t)b * S * C;
      d1.w = reinterpret_cast<const float*>(k.p) + (size_t)b * S * C;
      d1.w_kind = 1;
      d1.out = scores;
      d1.B = 1; d1.Hi = 1; d1.Wi = S; d1.Cin = C; d1.Ho = 1; d1.Wo = S; d1.N = S;
      ConvF32Desc d2;   // out[q][c] = sum_k P[q][k] V[k][c]
      d2.x = scores;
      d2.w = reinterpret_cast<const float*>(v.p) + (size_t)b * S * C;
      d2.w_kind = 2;
      d2.out = reinterpret_cast<float*>(a.p) + (size_t)b * S * C;
      d2.B = 1; d2.Hi = 1; d2.Wi = S; d2.Cin = S; d2.Ho = 1; d2.Wo = S; d2.N = C;
      ops.push_back([d1, d2, scores, S, scale](hipStream_t s) {
        launch_conv_f32(d1, s);
        launch_row_softmax_f32(scores, S, S, scale, s);
        launch_conv_f32(d2, s);
      });
      ops.back().label = "VAE attention fp32: QK^T GEMM + row softmax + PV GEMM, S=" + std::to_string(S);
      ops.back().flop = 4.0 * (double)S * S * C;
    }
    return conv(ops, p + ".to_out.0", a, nullptr, C, 1, 1, 1, true, nullptr, h.p);
  }
  const int ldv = round_up(S, 8);
  Tensor vt = conv(ops, p + ".to_v", t0, nullptr, C, 1, 1, 1, true, nullptr, nullptr, kOutHalfT, ldv);
  SD_REQUIRE(C % 64 == 0 && S % 64 == 0, kUnsupported, "VAE attention needs C %% 64 == 0 and H*W %% 64 == 0");
  SD_REQUIRE(ldv == S, kUnsupported, "VAE attention needs H*W %% 8 == 0");
  half_t* scores = arena_.alloc_n<half_t>((size_t)S * S);
  Tensor a = new_tensor(B, H, W, C);
  const float scale = 1.0f / std::sqrt((float)C);
  for (int b = 0; b < B; ++b) {
    ConvDesc d1;   // scores[q][k] = sum_c Q[q][c] K[k][c]   (K tokens play the role of the weight matrix)
    d1.x0 = q.p + (size_t)b * S * C; d1.C0 = C; d1.w = k.p + (size_t)b * S * C;
    d1.out = scores; d1.B = 1; d1.Hi = 1; d1.Wi = S; d1.Ho = 1; d1.Wo = S; d1.N = S;
    ConvDesc d2;   // out[q][c] = sum_k P[q][k] V^T[c][k]
    d2.x0 = scores; d2.C0 = S; d2.w = vt.p + (size_t)b * C * ldv;
    d2.out = a.p + (size_t)b * S * C; d2.B = 1; d2.Hi = 1; d2.Wi = S; d2.Ho = 1; d2.Wo = S; d2.N = C;
    ws_need_ = std::max(ws_need_, std::max(conv_workspace_bytes(d1), conv_workspace_bytes(d2)));
    ops.push_back([this, d1, d2, scores, S, scale](hipStream_t s) {
      launch_conv(d1, ws_conv_, s);
      launch_row_softmax(scores, S, S, scale, s);
      launch_conv(d2, ws_conv_, s);
    });
    ops.back().label = "VAE attention: QK^T GEMM + row softmax + PV GEMM, S=" + std::to_string(S);
    ops.back().flop = 4.0 * (double)S * S * C;
  }
  return conv(ops, p + ".to_out.0", a, nullptr, C, 1, 1, 1, true, nullptr, h.p);
}

void UNet::build_vae_decoder() {
  const int B = cfg_.batch, H = cfg_.height, W = cfg_.width, n = cfg_.n_levels;
  const int Cz = cfg_.in_channels;
  cfg_.norm_eps = 1e-6f;
  in_z_ = arena_.alloc_n<float>((size_t)B * Cz * H * W);
  Tensor z = new_tensor(B, H, W, Cz);
  {
    float* src = in_z_;
    if (f32_) main_ops_.push_back([=](hipStream_t s) { launch_nchw_to_nhwc_f32(src, 1, reinterpret_cast<float*>(z.p), B, Cz, H, W, s); });
    else main_ops_.push_back([=](hipStream_t s) { launch_nchw_to_nhwc(src, 1, z.p, B, Cz, H, W, s); });
  }
  Tensor h = conv(main_ops_, "post_quant_conv", z, nullptr, Cz, 1, 1, 1, true, nullptr, nullptr);
  const int Ctop = cfg_.block_out_channels[n - 1];
  h = conv(main_ops_, "decoder.conv_in", h, nullptr, Ctop, 3, 1, 1, true, nullptr, nullptr);
  h = resnet(main_ops_, "decoder.mid_block.resnets.0", h, nullptr, Ctop, false);
  h = vae_attention(main_ops_, "decoder.mid_block.attentions.0", h);
  h = resnet(main_ops_, "decoder.mid_block.resnets.1", h, nullptr, Ctop, false);
  for (int i = 0; i < n; ++i) {
    const int cout = cfg_.block_out_channels[n - 1 - i];
    const std::string p = "decoder.up_blocks." + std::to_string(i);
    for (int j = 0; j < cfg_.layers_per_block + 1; ++j)
      h = resnet(main_ops_, p + ".resnets." + std::to_string(j), h, nullptr, cout, false);
    if (i != n - 1) h = conv(main_ops_, p + ".upsamplers.0.conv", h, nullptr, cout, 3, 1, 2, true, nullptr, nullptr);
  }
  Tensor t = group_norm(main_ops_, "decoder.conv_norm_out", h, nullptr, 1e-6f, true);
  {  // conv_out 3x3 -> 3 channels: one wavefront per pixel, written straight as fp32 NCHW
    SD_REQUIRE(cfg_.out_channels <= 8 && t.C % 8 == 0, kUnsupported, "VAE conv_out: %d -> %d channels", t.C,
               cfg_.out_channels);
    image_elems_ = (size_t)t.B * cfg_.out_channels * t.H * t.W;
    image_ = arena_.alloc_n<float>(image_elems_);
    ConvDesc d;
    d.x0 = t.p;
    d.C0 = t.C;
    d.w = upload_conv_weight("decoder.conv_out", cfg_.out_channels, t.C, 3, false);
    d.bias = upload_vec("decoder.conv_out.bias", cfg_.out_channels);
    d.B = t.B; d.Hi = t.H; d.Wi = t.W; d.Ho = t.H; d.Wo = t.W;
    d.ksize = 3; d.stride = 1; d.up = 1; d.N = cfg_.out_channels;
    float* dst = image_;
    if (f32_) {
      Tensor o = new_tensor(t.B, t.H, t.W, cfg_.out_channels);
      ConvF32Desc fd;
      fd.x = reinterpret_cast<const float*>(t.p); fd.w = d.w; fd.bias = d.bias; fd.out = reinterpret_cast<float*>(o.p);
      fd.B = t.B; fd.Hi = t.H; fd.Wi = t.W; fd.Cin = t.C; fd.Ho = t.H; fd.Wo = t.W; fd.N = cfg_.out_channels; fd.ksize = 3;
      main_ops_.push_back([=](hipStream_t s) {
        launch_conv_f32(fd, s);
        launch_nhwc_to_nchw_f32f32(fd.out, dst, fd.B, fd.N, fd.Ho, fd.Wo, s);
      });
    } else {
      main_ops_.push_back([=](hipStream_t s) { launch_conv_small_n(d, dst, s); });
    }
    main_ops_.back().label = "conv3x3 small-N decoder.conv_out -> fp32 NCHW";
    main_ops_.back().flop = 2.0 * t.B * t.H * t.W * (double)cfg_.out_channels * t.C * 9;
  }
  if (ws_need_ > 0) {
    ws_conv_.partial = reinterpret_cast<float*>(arena_.alloc(ws_need_));
    ws_conv_.partial_bytes = ws_need_;
  }
}

// pipeline.py:313-320 hands z = latents / scaling_factor (fp16 or fp32); returns image in [-1, 1]
void UNet::vae_decode(const void* z, int z_is_f32, float* image, int flags) {
  SD_HIP(hipSetDevice(device_));
  SD_REQUIRE(cfg_.is_vae_decoder == 1, kInvalidArgument, "handle is not a VAE decoder");
  const bool dev = (flags & SD_FLAG_DEVICE_PTRS) != 0;
  const size_t n = (size_t)cfg_.batch * cfg_.in_channels * cfg_.height * cfg_.width;
  if (z_is_f32) {
    SD_HIP(hipMemcpyAsync(in_z_, z, n * 4, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream_));
  } else {
    if (!z_half_) z_half_ = arena_.alloc_n<half_t>(n);
    SD_HIP(hipMemcpyAsync(z_half_, z, n * 2, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream_));
    launch_half_to_float(z_half_, in_z_, n, stream_);
  }
  run_vae_graph();
  SD_HIP(hipMemcpyAsync(image, image_, image_elems_ * sizeof(float),
                        dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, stream_));
  SD_HIP(hipStreamSynchronize(stream_));
  have_inputs_ = true;
}

void UNet::run_vae_graph() {
  if (cfg_.use_graph) {
    if (!graph_) {
      run_ops(main_ops_);
      SD_HIP(hipStreamSynchronize(stream_));
      graph_ = capture([&] { run_ops(main_ops_); });
    }
    SD_HIP(hipGraphLaunch(graph_, stream_));
  } else {
    run_ops(main_ops_);
  }
}

// ---------------------------------------------------------------------------------------------
// VAE encoder (diffusers AutoencoderKL.encode up to the moments; the reference wraps it as
// latent = quant_conv(encoder(x)), torch2coreml.py:739-749, and samples / scales in Encoder.swift:48-90).
// Third-party arithmetic restated from the public architecture:
//   conv_in 3x3 (3 -> C0) -> down blocks (layers_per_block ResNets, then conv3x3 stride 2 on F.pad(x, (0,1,0,1)) in
//   all but the last) -> mid [ResNet, 1-head self-attention, ResNet] -> GroupNorm(32, 1e-6) -> SiLU -> conv_out 3x3
//   (-> 2 * latent channels) -> quant_conv 1x1.   Config: block_out_channels in encoder order, in_channels = 3,
//   out_channels = 2 * latent channels, height / width = IMAGE size.  Output: moments (B, 2*Cz, H/8, W/8) f32.
// ---------------------------------------------------------------------------------------------
void UNet::build_vae_encoder() {
  const int B = cfg_.batch, H = cfg_.height, W = cfg_.width, n = cfg_.n_levels;
  const int Cm = cfg_.out_channels;
  cfg_.norm_eps = 1e-6f;
  SD_REQUIRE(cfg_.in_channels == 3 && Cm >= 2 && Cm <= 8, kUnsupported, "VAE encoder: %d -> %d channels", cfg_.in_channels, Cm);
  in_x_ = arena_.alloc((size_t)B * 3 * H * W * 4);
  Tensor x = new_tensor(B, H, W, 3);
  {
    void* src = in_x_;
    UNet* self = this;
    if (f32_) main_ops_.push_back([=](hipStream_t s) { launch_nchw_to_nhwc_f32(src, self->vae_in_f32_, reinterpret_cast<float*>(x.p), B, 3, H, W, s); });
    else main_ops_.push_back([=](hipStream_t s) { launch_nchw_to_nhwc(src, self->vae_in_f32_, x.p, B, 3, H, W, s); });
    main_ops_.back().label = "boundary: image NCHW -> NHWC";
  }
  Tensor h = conv(main_ops_, "encoder.conv_in", x, nullptr, cfg_.block_out_channels[0], 3, 1, 1, true, nullptr, nullptr);
  for (int i = 0; i < n; ++i) {
    const int cout = cfg_.block_out_channels[i];
    const std::string p = "encoder.down_blocks." + std::to_string(i);
    for (int j = 0; j < cfg_.layers_per_block; ++j)
      h = resnet(main_ops_, p + ".resnets." + std::to_string(j), h, nullptr, cout, false);
    if (i != n - 1)
      h = conv(main_ops_, p + ".downsamplers.0.conv", h, nullptr, cout, 3, 2, 1, true, nullptr, nullptr, kOutHalf, 0, false, 0);
  }
  const int Ctop = cfg_.block_out_channels[n - 1];
  h = resnet(main_ops_, "encoder.mid_block.resnets.0", h, nullptr, Ctop, false);
  h = vae_attention(main_ops_, "encoder.mid_block.attentions.0", h);
  h = resnet(main_ops_, "encoder.mid_block.resnets.1", h, nullptr, Ctop, false);
  Tensor t = group_norm(main_ops_, "encoder.conv_norm_out", h, nullptr, 1e-6f, true);
  SD_REQUIRE(t.C % 8 == 0, kUnsupported, "VAE encoder conv_out: %d input channels", t.C);
  Tensor m = new_tensor(t.B, t.H, t.W, Cm);
  {
    ConvDesc d;
    d.x0 = t.p;
    d.C0 = t.C;
    d.w = upload_conv_weight("encoder.conv_out", Cm, t.C, 3, false);
    d.bias = upload_vec("encoder.conv_out.bias", Cm);
    d.out = m.p;
    d.B = t.B; d.Hi = t.H; d.Wi = t.W; d.Ho = t.H; d.Wo = t.W;
    d.ksize = 3; d.stride = 1; d.up = 1; d.N = Cm;
    if (f32_) {
      ConvF32Desc fd;
      fd.x = reinterpret_cast<const float*>(t.p); fd.w = d.w; fd.bias = d.bias; fd.out = reinterpret_cast<float*>(m.p);
      fd.B = t.B; fd.Hi = t.H; fd.Wi = t.W; fd.Cin = t.C; fd.Ho = t.H; fd.Wo = t.W; fd.N = Cm; fd.ksize = 3;
      main_ops_.push_back([=](hipStream_t s) { launch_conv_f32(fd, s); });
    } else {
      main_ops_.push_back([=](hipStream_t s) { launch_conv_small_n(d, nullptr, s); });
    }
    main_ops_.back().label = "conv3x3 small-N encoder.conv_out";
  }
  Tensor q = conv(main_ops_, "quant_conv", m, nullptr, Cm, 1, 1, 1, true, nullptr, nullptr);
  image_elems_ = q.numel();
  image_ = arena_.alloc_n<float>(image_elems_);
  {
    float* dst = image_;
    if (f32_) main_ops_.push_back([=](hipStream_t s) { launch_nhwc_to_nchw_f32f32(reinterpret_cast<const float*>(q.p), dst, q.B, q.C, q.H, q.W, s); });
    else main_ops_.push_back([=](hipStream_t s) { launch_nhwc_to_nchw_f32(q.p, dst, q.B, q.C, q.H, q.W, s); });
    main_ops_.back().label = "boundary: moments NHWC -> NCHW fp32";
  }
  if (ws_need_ > 0) {
    ws_conv_.partial = reinterpret_cast<float*>(arena_.alloc(ws_need_));
    ws_conv_.partial_bytes = ws_need_;
  }
}

void UNet::vae_encode(const void* x, int x_is_f32, float* moments, int flags) {
  SD_HIP(hipSetDevice(device_));
  SD_REQUIRE(cfg_.is_vae_decoder == 2, kInvalidArgument, "handle is not a VAE encoder");
  const bool dev = (flags & SD_FLAG_DEVICE_PTRS) != 0;
  const size_t n = (size_t)cfg_.batch * 3 * cfg_.height * cfg_.width;
  if ((x_is_f32 != 0) != (vae_in_f32_ != 0)) {   // the captured boundary kernel bakes the input dtype in
    vae_in_f32_ = x_is_f32 ? 1 : 0;
    invalidate_graphs();
  }
  SD_HIP(hipMemcpyAsync(in_x_, x, n * (x_is_f32 ? 4 : 2), dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream_));
  run_vae_graph();
  SD_HIP(hipMemcpyAsync(moments, image_, image_elems_ * sizeof(float), dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
                        stream_));
  SD_HIP(hipStreamSynchronize(stream_));
  have_inputs_ = true;
}

void UNet::invalidate_graphs() {
  if (graph_) { (void)hipGraphExecDestroy(graph_); graph_ = nullptr; }
  if (loop_graph_) { (void)hipGraphExecDestroy(loop_graph_); loop_graph_ = nullptr; }
}

void UNet::set_attention(int impl) {
  SD_REQUIRE(impl >= 0 && impl <= 2, kInvalidArgument, "attention impl %d", impl);
  if (impl != cfg_.attention_impl) {
    cfg_.attention_impl = impl;
    invalidate_graphs();
  }
  for (UNet* cn : attached_) cn->set_attention(impl);   // their launches live inside this handle's graph
}

void UNet::run_ops(const std::vector<Op>& ops) { run_ops_on(ops, stream_); }
void UNet::run_ops_on(const std::vector<Op>& ops, hipStream_t s) {
  // SD_NAN_TRACE=1 (with SD_TUNE; eager launches only): after every op the whole arena is scanned for fp16 Inf / NaN patterns and the op
  // after which their number changes is named on stderr (fp32 regions give false positives - read the list with the op in mind).
  static const bool trace = tune_env_set("SD_NAN_TRACE");
  if (trace) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) {
      unsigned long long* d = nullptr;
      SD_HIP(hipMalloc(&d, sizeof(unsigned long long)));
      auto scan = [&]() {
        SD_HIP(hipMemsetAsync(d, 0, sizeof(unsigned long long), s));
        for (size_t i = 0; i < arena_.chunks().size(); ++i) launch_count_nonfinite_half(arena_.chunks()[i], arena_.chunk_bytes()[i], d, s);
        unsigned long long h = 0;
        SD_HIP(hipMemcpyAsync(&h, d, sizeof(h), hipMemcpyDeviceToHost, s));
        SD_HIP(hipStreamSynchronize(s));
        return h;
      };
      unsigned long long last = scan();
      fprintf(stderr, "nan-trace: %llu non-finite fp16 patterns in the arena before the list\n", last);
      for (auto& op : ops) {
        op(s);
        const unsigned long long now = scan();
        if (now != last) fprintf(stderr, "nan-trace: %llu -> %llu after  %s\n", last, now, op.label.c_str());
        last = now;
      }
      (void)hipFree(d);
      return;
    }
  }
  static const bool poison = tune_env_set("SD_POISON_LDS");   // debug: NaN patterns into every CU's LDS in front of every op
  for (auto& op : ops) {
    if (poison) launch_lds_poison(s);
    op(s);
  }
}

// Stream capture that cannot leave the stream in capture mode: an op that throws between Begin and
// End (SD_HIP / SD_REQUIRE inside a launch) ends and discards the capture before the error travels on.
hipGraphExec_t UNet::capture(const std::function<void()>& body) {
  SD_HIP(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
  hipGraph_t g = nullptr;
  try {
    body();
  } catch (...) {
    (void)hipStreamEndCapture(stream_, &g);
    if (g) (void)hipGraphDestroy(g);
    throw;
  }
  SD_HIP(hipStreamEndCapture(stream_, &g));
  hipGraphExec_t exec = nullptr;
  const hipError_t e = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
  (void)hipGraphDestroy(g);
  SD_REQUIRE(e == hipSuccess, kHipError, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
  return exec;
}

namespace {
// hipEvents that are destroyed on every exit path
struct EventList {
  std::vector<hipEvent_t> ev;
  explicit EventList(size_t n) {
    ev.reserve(n);
    for (size_t i = 0; i < n; ++i) {
      hipEvent_t e = nullptr;
      SD_HIP(hipEventCreate(&e));
      ev.push_back(e);
    }
  }
  ~EventList() {
    for (hipEvent_t e : ev) (void)hipEventDestroy(e);
  }
  hipEvent_t operator[](size_t i) const { return ev[i]; }
};
}  // namespace

void UNet::run_time_and_main() { run_main(true); }

void UNet::run_main(bool with_time) {
  const bool fork_time = with_time && side_ && !time_ops_.empty() && temb_join_pos_ >= 0;
  if (with_time && !fork_time) run_ops(time_ops_);
  if (fork_time) {
    // fork: the side stream sees everything queued so far (the timestep buffer), runs the embedding
    // MLP + the batched time_emb_proj GEMV, and the main stream waits for it only where the first
    // ResNet conv adds its slice.  Works eagerly and under stream capture (the side stream joins the
    // capture through the event and is joined back before the capture ends).
    SD_HIP(hipEventRecord(ev_fork_, stream_));
    SD_HIP(hipStreamWaitEvent(side_, ev_fork_, 0));
    for (auto& op : time_ops_) op(side_);
    SD_HIP(hipEventRecord(ev_join_, side_));
  }
  for (size_t i = 0; i < main_ops_.size(); ++i) {
    if (fork_time && (int)i == temb_join_pos_) SD_HIP(hipStreamWaitEvent(stream_, ev_join_, 0));
    if (cn_join_pending_ && (int)i == cn_join_pos_) {   // the attached ControlNets (run_attached) are done: their residuals may be read
      SD_HIP(hipStreamWaitEvent(stream_, ev_cn_join_, 0));
      cn_join_pending_ = false;
    }
    static const bool poison = tune_env_set("SD_POISON_LDS");
    if (poison) launch_lds_poison(stream_);
    main_ops_[i](stream_);
  }
  SD_REQUIRE(!cn_join_pending_, kInternal, "forked ControlNets were never joined");
}

// ---- device-resident ControlNet hand-off ---------------------------------------------------------
void UNet::attach_controlnets(const std::vector<UNet*>& cns) {
  SD_REQUIRE(!cfg_.is_controlnet && !cfg_.is_vae_decoder, kInvalidArgument, "attach_controlnets needs a UNet handle");
  SD_REQUIRE(cns.empty() || cfg_.support_controlnet, kInvalidArgument,
             "this UNet was built without support_controlnet (unet.py:1009-1022)");
  SD_REQUIRE(cns.size() <= 3, kUnsupported, "at most 3 ControlNets per UNet handle, got %zu", cns.size());
  for (UNet* cn : cns) {
    SD_REQUIRE(cn && cn->cfg_.is_controlnet, kInvalidArgument, "attach_controlnets: handle is not a ControlNet");
    SD_REQUIRE(cn->device_ == device_, kInvalidArgument, "ControlNet lives on device %d, UNet on %d", cn->device_, device_);
    SD_REQUIRE(cn->cfg_.batch == cfg_.batch && cn->cfg_.height == cfg_.height && cn->cfg_.width == cfg_.width &&
                   cn->cfg_.cross_attention_dim == cfg_.cross_attention_dim && cn->cfg_.context_len == cfg_.context_len,
               kInvalidArgument, "ControlNet static shapes differ from the UNet's");
    SD_REQUIRE(cn->res_shapes_ == res_shapes_, kInvalidArgument, "ControlNet residual shapes differ from the UNet's skips");
  }
  SD_HIP(hipSetDevice(device_));
  SD_HIP(hipStreamSynchronize(stream_));
  // a ControlNet that leaves this handle forgets its conditioning image: re-attaching it later without a fresh
  // sd_controlnet_set_cond must fail loudly instead of running with the previous generation's image
  for (UNet* old : attached_)
    if (std::find(cns.begin(), cns.end(), old) == cns.end()) old->have_cond_ = false;
  attached_ = cns;
  for (UNet* cn : attached_) cn->set_attention(cfg_.attention_impl);
  invalidate_graphs();     // the captured launches bake in which residual sources are read
  have_ctx_ = false;       // the ControlNets' hoisted cross-attention K/V follow this handle's prompt
}

// controlnet.py:211-215: the conditioning embedding depends only on the image -> once per generation
void UNet::set_controlnet_cond(const void* cond, int flags) {
  SD_HIP(hipSetDevice(device_));
  SD_REQUIRE(cfg_.is_controlnet && in_cond_, kInvalidArgument, "set_controlnet_cond needs a ControlNet handle");
  SD_REQUIRE(cond != nullptr, kInvalidArgument, "missing input 'controlnet_cond'");
  const size_t n = (size_t)cfg_.batch * 3 * cfg_.height * 8 * cfg_.width * 8;
  const bool dev = (flags & SD_FLAG_DEVICE_PTRS) != 0;
  SD_HIP(hipMemcpyAsync(in_cond_, cond, n * 2, dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, stream_));
  run_ops(cond_ops_);
  SD_HIP(hipStreamSynchronize(stream_));
  have_cond_ = true;
  if (!dev) {
    const uint16_t* src = reinterpret_cast<const uint16_t*>(cond);
    last_cond_.assign(src, src + n);
  } else {
    last_cond_.clear();
  }
}

void UNet::set_context_from(const half_t* ehs_dev, hipStream_t s) {
  const size_t n = (size_t)cfg_.batch * cfg_.cross_attention_dim * cfg_.context_len;
  SD_HIP(hipMemcpyAsync(in_ehs_, ehs_dev, n * 2, hipMemcpyDeviceToDevice, s));
  run_ops_on(ctx_ops_, s);
  have_ctx_ = true;
  last_ehs_.clear();
}

void UNet::run_as_controlnet(hipStream_t s, const half_t* x_nhwc, const float* tbuf) {
  SD_REQUIRE(have_cond_, kInvalidArgument, "ControlNet has no conditioning image: call sd_controlnet_set_cond first");
  SD_HIP(hipMemcpyAsync(x_in_.p, x_nhwc, x_in_.numel() * sizeof(half_t), hipMemcpyDeviceToDevice, s));
  SD_HIP(hipMemcpyAsync(tbuf_, tbuf, (size_t)cfg_.batch * sizeof(float), hipMemcpyDeviceToDevice, s));
  run_ops_on(time_ops_, s);
  run_ops_on(main_ops_, s);
}

void UNet::run_attached() {
  if (attached_.empty()) return;
  if (!cn_stream_ || cn_join_pos_ < 0) {   // serial (SD_CN_CONCURRENT=0)
    for (UNet* cn : attached_) cn->run_as_controlnet(stream_, x_in_.p, tbuf_);
    return;
  }
  //  a stale flag from a run that threw between the fork and the join must not survive; every attached handle is
  // checked BEFORE the fork so that the usual error - no conditioning image - cannot leave the side stream forked
  cn_join_pending_ = false;
  for (UNet* cn : attached_)
    SD_REQUIRE(cn->have_cond_, kInvalidArgument, "ControlNet has no conditioning image: call sd_controlnet_set_cond first");
  // fork behind the sample / timestep hand-over (everything queued on stream_ so far); run_main joins in front of the first residual add
  SD_HIP(hipEventRecord(ev_cn_fork_, stream_));
  SD_HIP(hipStreamWaitEvent(cn_stream_, ev_cn_fork_, 0));
  try {
    for (UNet* cn : attached_) cn->run_as_controlnet(cn_stream_, x_in_.p, tbuf_);
  } catch (...) {
    // join what was forked (eagerly: the main stream waits for the side stream; under capture: the side stream re-joins the
    // capture so that hipStreamEndCapture can discard it), then let the error travel on
    (void)hipEventRecord(ev_cn_join_, cn_stream_);
    (void)hipStreamWaitEvent(stream_, ev_cn_join_, 0);
    throw;
  }
  SD_HIP(hipEventRecord(ev_cn_join_, cn_stream_));
  cn_join_pending_ = true;
}

void UNet::upload_inputs(const sd_unet_io& io, bool loop_mode) {
  const int B = cfg_.batch, L = cfg_.context_len;
  const bool dev = (io.flags & SD_FLAG_DEVICE_PTRS) != 0;
  const hipMemcpyKind kind = dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  auto copy_in = [&](void* dst, const void* src, size_t bytes, const char* what) {
    SD_REQUIRE(src != nullptr, kInvalidArgument, "missing input '%s'", what);
    SD_HIP(hipMemcpyAsync(dst, src, bytes, kind, stream_));
  };
  if (!loop_mode) {
    copy_in(in_sample_, io.sample, (size_t)B * cfg_.in_channels * cfg_.height * cfg_.width * 2, "sample");
    copy_in(in_timestep_, io.timestep, (size_t)B * 2, "timestep");
  }
  if (in_time_ids_) {
    copy_in(in_time_ids_, io.time_ids, (size_t)B * cfg_.num_time_ids * 2, "time_ids");
    const int text_dim = cfg_.projection_class_embeddings_input_dim - cfg_.num_time_ids * cfg_.addition_time_embed_dim;
    copy_in(in_text_embeds_, io.text_embeds, (size_t)B * text_dim * 2, "text_embeds");
  }
  // encoder_hidden_states: re-project K/V only when the prompt embedding changed
  {
    const size_t n = (size_t)B * cfg_.cross_attention_dim * L;
    SD_REQUIRE(io.encoder_hidden_states != nullptr, kInvalidArgument, "missing input 'encoder_hidden_states'");
    bool changed = true;
    if (!dev) {
      const uint16_t* src = reinterpret_cast<const uint16_t*>(io.encoder_hidden_states);
      if (have_ctx_ && last_ehs_.size() == n && std::memcmp(last_ehs_.data(), src, n * 2) == 0) changed = false;
      if (changed) last_ehs_.assign(src, src + n);
    }
    if (changed) {
      SD_HIP(hipMemcpyAsync(in_ehs_, io.encoder_hidden_states, n * 2, kind, stream_));
      run_ops(ctx_ops_);
      have_ctx_ = true;
      for (UNet* cn : attached_) cn->set_context_from(in_ehs_, stream_);   // same prompt (pipeline.py:266-268)
    }
  }
  if (in_cond_) {
    const size_t n = (size_t)B * 3 * cfg_.height * 8 * cfg_.width * 8;
    SD_REQUIRE(io.controlnet_cond != nullptr, kInvalidArgument, "missing input 'controlnet_cond'");
    bool changed = true;
    if (!dev) {
      const uint16_t* src = reinterpret_cast<const uint16_t*>(io.controlnet_cond);
      if (have_cond_ && last_cond_.size() == n && std::memcmp(last_cond_.data(), src, n * 2) == 0) changed = false;
      if (changed) last_cond_.assign(src, src + n);
    }
    if (changed) {
      SD_HIP(hipMemcpyAsync(in_cond_, io.controlnet_cond, n * 2, kind, stream_));
      run_ops(cond_ops_);
      have_cond_ = true;
    }
  }
  if (!in_res_nchw_.empty() && attached_.empty()) {
    SD_REQUIRE(io.additional_residuals && io.num_additional_residuals == (int)in_res_nchw_.size(), kInvalidArgument,
               "expected %zu additional_residual inputs, got %d", in_res_nchw_.size(), io.num_additional_residuals);
    for (size_t i = 0; i < in_res_nchw_.size(); ++i) {
      const auto& sh = res_shapes_[i];
      copy_in(in_res_nchw_[i], io.additional_residuals[i], (size_t)sh[0] * sh[1] * sh[2] * sh[3] * 2,
              "additional_residual");
    }
  }
  have_inputs_ = true;
}

// one forward: boundary conversions, [attached ControlNets,] time embedding, the network, [ControlNet
// boundary outputs]
void UNet::run_forward_ops() {
  run_ops(in_ops_);
  run_attached();
  run_time_and_main();
  run_ops(out_ops_);
}

void UNet::ensure_graph() {
  if (graph_ || !cfg_.use_graph) return;
  // first run eagerly (sets kernel attributes, warms code objects), then capture
  run_forward_ops();
  SD_HIP(hipStreamSynchronize(stream_));
  graph_ = capture([&] { run_forward_ops(); });
}

void UNet::forward(const sd_unet_io& io) {
  SD_HIP(hipSetDevice(device_));
  SD_REQUIRE(!cfg_.is_vae_decoder, kInvalidArgument, "handle is a VAE decoder: use sd_vae_decode");
  upload_inputs(io, false);
  if (cfg_.use_graph) {
    ensure_graph();
    SD_HIP(hipGraphLaunch(graph_, stream_));
  } else {
    run_forward_ops();
  }
  const bool dev = (io.flags & SD_FLAG_DEVICE_PTRS) != 0;
  const hipMemcpyKind kind = dev ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
  if (cfg_.is_controlnet) {
    SD_REQUIRE(io.residual_outputs != nullptr, kInvalidArgument, "missing residual_outputs");
    for (size_t i = 0; i < res_out_.size(); ++i)
      SD_HIP(hipMemcpyAsync(io.residual_outputs[i], res_out_[i], cn_out_[i].numel() * sizeof(float), kind, stream_));
  } else {
    SD_REQUIRE(io.noise_pred != nullptr, kInvalidArgument, "missing output 'noise_pred'");
    SD_HIP(hipMemcpyAsync(io.noise_pred, noise_pred_,
                          (size_t)cfg_.batch * cfg_.out_channels * cfg_.height * cfg_.width * sizeof(float), kind,
                          stream_));
  }
  SD_HIP(hipStreamSynchronize(stream_));
}

float UNet::time_forward(int warmup, int iters) {
  SD_HIP(hipSetDevice(device_));
  SD_REQUIRE(have_inputs_, kInvalidArgument, "time_forward: call sd_unet_forward once first");
  SD_REQUIRE(iters >= 1, kInvalidArgument, "iters must be >= 1");
  auto once = [&]() {
    if (cfg_.is_vae_decoder) {
      if (cfg_.use_graph && graph_)
        SD_HIP(hipGraphLaunch(graph_, stream_));
      else
        run_ops(main_ops_);
    } else if (cfg_.use_graph) {
      ensure_graph();
      SD_HIP(hipGraphLaunch(graph_, stream_));
    } else {
      run_forward_ops();
    }
  };
  for (int i = 0; i < warmup; ++i) once();
  EventList ev(2);
  SD_HIP(hipEventRecord(ev[0], stream_));
  for (int i = 0; i < iters; ++i) once();
  SD_HIP(hipEventRecord(ev[1], stream_));
  SD_HIP(hipEventSynchronize(ev[1]));
  float ms = 0.f;
  SD_HIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
  return ms / (float)iters;
}

std::vector<OpTime> UNet::profile(int iters) {
  SD_HIP(hipSetDevice(device_));
  SD_REQUIRE(have_inputs_, kInvalidArgument, "profile: call sd_unet_forward / sd_vae_decode once first");
  SD_REQUIRE(iters >= 1 && iters <= 100, kInvalidArgument, "iters must be in 1..100");
  std::vector<const Op*> seq;
  for (const auto* list : {&in_ops_, &time_ops_, &main_ops_, &out_ops_})
    for (const Op& op : *list) seq.push_back(&op);
  const size_t n = seq.size();
  std::vector<std::vector<float>> t(n, std::vector<float>((size_t)iters));
  for (const Op* op : seq) (*op)(stream_);   // warm (kernel attributes, code objects)
  SD_HIP(hipStreamSynchronize(stream_));
  EventList ev(n + 1);
  for (int it = 0; it < iters; ++it) {
    SD_HIP(hipEventRecord(ev[0], stream_));
    for (size_t i = 0; i < n; ++i) {
      (*seq[i])(stream_);
      SD_HIP(hipEventRecord(ev[i + 1], stream_));
    }
    SD_HIP(hipEventSynchronize(ev[n]));
    for (size_t i = 0; i < n; ++i) SD_HIP(hipEventElapsedTime(&t[i][(size_t)it], ev[i], ev[i + 1]));
  }
  std::vector<OpTime> out;
  for (size_t i = 0; i < n; ++i) {
    std::sort(t[i].begin(), t[i].end());
    out.push_back({seq[i]->label.empty() ? std::string("op") : seq[i]->label, seq[i]->flop, t[i][(size_t)iters / 2]});
  }
  return out;
}

// pipeline.py:500-573 on the device: per step {duplicate latents + fp16 cast, [ControlNets,] UNet, CFG
// combine, scheduler update}; the host only replays one graph per step.
void UNet::denoise_loop(const sd_unet_io& io, float* latents, int n_images, int n_steps, const float* timesteps,
                        const float* coef, const float* sample_scale, int history, float guidance, float* history_io,
                        float* ms_per_step) {
  SD_HIP(hipSetDevice(device_));
  SD_REQUIRE(!cfg_.is_controlnet && !cfg_.is_vae_decoder, kInvalidArgument, "denoise_loop needs a UNet handle");
  const int cfgmul = guidance > 1.0f ? 2 : 1;   // pipeline.py:443
  SD_REQUIRE(cfgmul * n_images == cfg_.batch, kInvalidArgument,
             "UNet batch %d != %d images x %d (guidance_scale %s 1)", cfg_.batch, n_images, cfgmul,
             cfgmul == 2 ? ">" : "<=");
  SD_REQUIRE(cfg_.in_channels == cfg_.out_channels, kInvalidArgument, "loop needs in_channels == out_channels");
  SD_REQUIRE(history >= 0 && history <= 3 && n_steps >= 1, kInvalidArgument, "bad history (%d, max 3) / n_steps (%d)",
             history, n_steps);
  SD_REQUIRE(in_res_nchw_.empty() || !attached_.empty(), kInvalidArgument,
             "a UNet built with support_controlnet needs attached ControlNets for the device-resident loop "
             "(sd_unet_attach_controlnets); host residuals only travel through sd_unet_forward");
  const int C = cfg_.in_channels, H = cfg_.height, W = cfg_.width;
  const size_t lat_n = (size_t)n_images * C * H * W;
  const size_t hist_n = (size_t)3 * cfg_.batch * C * H * W;
  if (!latents_) {
    latents_ = arena_.alloc_n<float>((size_t)cfg_.batch * C * H * W);
    eps_hist_ = arena_.alloc_n<float>(hist_n);
    step_ = arena_.alloc_n<int>(2);   // [0]: step counter, [1]: arrival ticket of cfg_sched_step_kernel
  }
  if (tab_cap_ < n_steps) {
    tab_cap_ = std::max(n_steps, 1024);
    tab_timesteps_ = arena_.alloc_n<float>(tab_cap_);
    tab_coef_ = arena_.alloc_n<float>((size_t)tab_cap_ * 8);
    tab_scale_ = arena_.alloc_n<float>(tab_cap_);
    if (loop_graph_) { (void)hipGraphExecDestroy(loop_graph_); loop_graph_ = nullptr; }
  }
  upload_inputs(io, true);
  SD_HIP(hipMemcpyAsync(latents_, latents, lat_n * sizeof(float), hipMemcpyHostToDevice, stream_));
  SD_HIP(hipMemcpyAsync(tab_timesteps_, timesteps, (size_t)n_steps * sizeof(float), hipMemcpyHostToDevice, stream_));
  SD_HIP(hipMemcpyAsync(tab_coef_, coef, (size_t)n_steps * 8 * sizeof(float), hipMemcpyHostToDevice, stream_));
  SD_HIP(hipMemsetAsync(step_, 0, 2 * sizeof(int), stream_));
  SD_HIP(hipMemsetAsync(eps_hist_, 0, hist_n * sizeof(float), stream_));
  if (history_io && history > 0)   // slot j: [n_images][C][H][W], continuing a loop another handle began (refiner swap)
    for (int j = 0; j < history; ++j)
      SD_HIP(hipMemcpyAsync(eps_hist_ + (size_t)j * lat_n, history_io + (size_t)j * lat_n, lat_n * sizeof(float),
                            hipMemcpyHostToDevice, stream_));
  if (sample_scale)
    SD_HIP(hipMemcpyAsync(tab_scale_, sample_scale, (size_t)n_steps * sizeof(float), hipMemcpyHostToDevice, stream_));
  // Time path out of the step (unet.py:703-728, :454-456): without added conditioning the embedding MLP and the batched
  // time_emb_proj depend on the timestep only, so their outputs for ALL steps of this call are computed here in one batched
  // pass (rows = steps x batch through the same GEMV kernels) and each step copies its rows inside loop_prep - four
  // dependent launches (~65 us) less per step for ~0.5 ms once per call.  Recomputed on every call: nothing is cached
  // across generations.  SDXL's text_time conditioning enters the same MLP: it keeps the in-step path.
  static const bool hoist_off = tune_env_set("SD_NO_TEMB_TABLE");   // A/B switch
  const bool hoist = !hoist_off && tpath_.ok && temb_w_all_ && temb_used_ > 0 && temb_used_ % 4 == 0;
  if (hoist) {
    const int Bt = cfg_.batch, rows = n_steps * Bt;
    if (temb_tab_cap_ < n_steps) {
      temb_tab_cap_ = std::max(n_steps, 64);
      const size_t r = (size_t)temb_tab_cap_ * Bt;
      temb_tab_ = arena_.alloc_n<float>(r * kTembCap);
      tt_tab_ = arena_.alloc_n<float>(r);
      tsin_tab_ = arena_.alloc_n<float>(r * tpath_.c0);
      e1_tab_ = arena_.alloc_n<float>(r * tpath_.tdim);
      emb_tab_ = arena_.alloc_n<float>(r * tpath_.tdim);
      if (loop_graph_) { (void)hipGraphExecDestroy(loop_graph_); loop_graph_ = nullptr; }   // the table address is baked in
    }
    // every call recomputes the table (no caching across generations): one batched pass over all steps' timesteps
    std::vector<float> tt((size_t)rows);
    for (int i = 0; i < n_steps; ++i)
      for (int b = 0; b < Bt; ++b) tt[(size_t)i * Bt + b] = timesteps[i];
    SD_HIP(hipMemcpyAsync(tt_tab_, tt.data(), tt.size() * sizeof(float), hipMemcpyHostToDevice, stream_));
    SD_HIP(hipStreamSynchronize(stream_));   // tt is a local
    const int c0 = tpath_.c0, td = tpath_.tdim;
    launch_timestep_embedding(tt_tab_, tpath_.freq, tsin_tab_, rows, c0, stream_);
    launch_gemv(tpath_.w1, tpath_.b1, tsin_tab_, c0, e1_tab_, td, rows, td, c0, 0, 1, 0, stream_);
    launch_gemv(tpath_.w2, tpath_.b2, e1_tab_, td, emb_tab_, td, rows, td, td, 0, 0, 0, stream_);
    launch_gemv(temb_w_all_, temb_b_all_, emb_tab_, td, temb_tab_, kTembCap, rows, temb_used_, td, 1, 0, 0, stream_);
  }
  LoopTables tab{tab_timesteps_, tab_coef_, step_, sample_scale ? tab_scale_ : nullptr,
                 reinterpret_cast<unsigned*>(step_ + 1), hoist ? temb_tab_ : nullptr, temb_all_, cfg_.batch, kTembCap, temb_used_};
  if (io.step_noise) {   // ancestral samplers: the host's pre-drawn, pre-scaled noise of every step
    const size_t need = (size_t)n_steps * lat_n;
    if (noise_cap_ < need) {
      // the bump arena never frees: size the table with headroom (like the coefficient tables) so that a later call with a few
      // more steps re-uses it instead of abandoning the old block (and the captured loop graph with it)
      const size_t cap = std::max(need + need / 2, (size_t)64 * lat_n);
      noise_tab_ = arena_.alloc_n<float>(cap);
      noise_cap_ = cap;
      if (loop_graph_) { (void)hipGraphExecDestroy(loop_graph_); loop_graph_ = nullptr; }   // the address is baked in
    }
    SD_HIP(hipMemcpyAsync(noise_tab_, io.step_noise, need * sizeof(float), hipMemcpyHostToDevice, stream_));
    tab.noise_tab = noise_tab_;
  }
  auto step_ops = [&]() {
    launch_loop_prep(latents_, x_in_.p, tbuf_, tab, n_images, C, H, W, cfgmul, stream_);
    run_attached();
    run_main(!hoist);
    launch_cfg_sched_step(noise_pred_, latents_, eps_hist_, tab, guidance, n_images, C * H * W, cfgmul, history,
                          stream_);
  };
  const int key = n_images * 128 + (io.step_noise ? 64 : 0) + (hoist ? 32 : 0) + history * 4 + (cfgmul - 1) * 2 + (sample_scale ? 1 : 0);
  std::unique_ptr<EventList> ev;
  if (ms_per_step) ev = std::make_unique<EventList>((size_t)n_steps + 1);
  if (cfg_.use_graph) {
    // guidance is baked into the captured kernel arguments -> key the graph on its bit pattern too
    int gbits;
    std::memcpy(&gbits, &guidance, 4);
    const int full_key = key ^ (gbits * 31);
    if (!loop_graph_ || loop_graph_key_ != full_key) {
      if (loop_graph_) { (void)hipGraphExecDestroy(loop_graph_); loop_graph_ = nullptr; }
      if (!graph_) {   // make sure every kernel has been launched eagerly once (attributes, code objects)
        launch_loop_prep(latents_, x_in_.p, tbuf_, tab, n_images, C, H, W, cfgmul, stream_);
        run_attached();
        run_time_and_main();
        SD_HIP(hipStreamSynchronize(stream_));
      }
      loop_graph_ = capture(step_ops);
      loop_graph_key_ = full_key;
    }
  }
  for (int i = 0; i < n_steps; ++i) {
    if (ms_per_step) SD_HIP(hipEventRecord((*ev)[i], stream_));
    if (cfg_.use_graph)
      SD_HIP(hipGraphLaunch(loop_graph_, stream_));
    else
      step_ops();
  }
  if (ms_per_step) SD_HIP(hipEventRecord((*ev)[n_steps], stream_));
  SD_HIP(hipMemcpyAsync(latents, latents_, lat_n * sizeof(float), hipMemcpyDeviceToHost, stream_));
  if (history_io && history > 0)
    for (int j = 0; j < history; ++j)
      SD_HIP(hipMemcpyAsync(history_io + (size_t)j * lat_n, eps_hist_ + (size_t)j * lat_n, lat_n * sizeof(float),
                            hipMemcpyDeviceToHost, stream_));
  SD_HIP(hipStreamSynchronize(stream_));
  if (ms_per_step)
    for (int i = 0; i < n_steps; ++i) SD_HIP(hipEventElapsedTime(&ms_per_step[i], (*ev)[i], (*ev)[i + 1]));
}

}  // namespace sd
