// Host-side builder/executor of the UNet (and ControlNet) launch graph.
// Spec: python_coreml_stable_diffusion/unet.py:798-1152, controlnet.py:49-250.
#pragma once
#include <functional>

#include "../../include/sd_mi355x.h"
#include "kernels.h"
#include "weights.h"

namespace sd {

// Link between a conv / GEMM op and the GroupNorm that consumes its output: the GroupNorm (built later) asks the
// producer to leave per-tile (sum, sumsq) partials of the tensor in `partial`; at launch time the producer reports how
// many entries per (sample, group) its plan wrote (0: none - split-K, ragged tiles - the GroupNorm runs its own pass).
// `entries` is written by the producer's launch closure (produced()) and taken by the GroupNorm's (consume()): the GroupNorm op
// sits behind its producer in the launch list (UNet::group_norm asserts the positions at build time: ops_pos of the producer <
// its own), and at LAUNCH time a consumer that runs without its producer having run since the last consumption - an op timed on
// its own, a list walked out of order - gets 0 entries, i.e. runs its own statistics pass: stale partials are never folded.
//
// Twins (round 5): at the 8x8 / 16x16 levels (Ho * Wo <= 256) the producer can leave through fp32 slabs and the group-organised
// combine of wstream.hip, which holds whole (sample, group) slices and writes the GroupNorm(+SiLU) the consumer asks for next to
// the raw tensor - the GroupNorm launch disappears.  A GroupNorm built later registers up to two twins here (a tensor can be
// normalised on its own by the next resnet AND as the second source of an up-block concat); the producer's closure reads them
// at launch time.
struct GnHook {
  float* partial = nullptr;
  int groups = 0;
  int entries = 0;
  bool fresh = false;            // produced() since the last consume()
  void produced(int n) {
    entries = n;
    fresh = true;
  }
  int consume() {
    const int n = fresh ? entries : 0;
    fresh = false;
    return n;
  }
  int ops_pos = -1;              // index of the producing op in its launch list
  const void* ops_list = nullptr;
  bool twin_capable = false;     // the producer can run the slab + reduce_twin path
  ConvDesc desc;                 // the producer's conv (workspace sizing when a twin is attached)
  GnTwin twin[2];
  int n_twins = 0;
};

struct Tensor {
  half_t* p = nullptr;
  int B = 0, H = 0, W = 0, C = 0;
  std::shared_ptr<GnHook> gn;   // set on conv / GEMM outputs
  int M() const { return B * H * W; }
  size_t numel() const { return (size_t)B * H * W * C; }
};

// Self-attention operands of a transformer block written ahead of it (UNet::transformer's one-launch head, xattn_out.hip gn_proj_qkv_kernel)
struct PreQkv {
  Tensor qk;               // [M][2C]: q | k
  half_t* vt = nullptr;    // [B][C][round_up(S, 8)]
  bool vt_perm = false;    // V^T in attention8's key order
  bool q_pre = false;      // queries carry d^-0.5 * log2(e)
};

// One entry of a handle's launch list: the launch closure plus what the per-op profile reports about it.
struct Op {
  std::function<void(hipStream_t)> fn;
  std::string label;   // "<kind> <shape> <checkpoint name>"
  double flop = 0;     // algorithmic FLOP (2 per MAC) of MFMA ops, 0 for bandwidth ops
  Op() = default;
  template <class F, class = std::enable_if_t<!std::is_same<std::decay_t<F>, Op>::value>>
  Op(F&& f) : fn(std::forward<F>(f)) {}
  void operator()(hipStream_t s) const { fn(s); }
};

struct OpTime {
  std::string label;
  double flop;
  float ms;
};

class UNet {
 public:
  UNet(const sd_unet_config& cfg, const WeightStore& ws, int device);
  ~UNet();

  void forward(const sd_unet_io& io);
  float time_forward(int warmup, int iters);
  void drop_graphs() { invalidate_graphs(); }   // measurement hook: the next forward re-captures (sd_tune_set_plan_table)
  // HIP-event time of every op of one forward, in launch order (eager launches, cold caches between
  // dependent kernels exactly as inside the graph); median over `iters` passes
  std::vector<OpTime> profile(int iters);
  void denoise_loop(const sd_unet_io& io, float* latents, int n_images, int n_steps, const float* timesteps,
                    const float* coef, const float* sample_scale, int history, float guidance, float* history_io,
                    float* ms_per_step);
  // device-resident ControlNet hand-off (pipeline.py:259-284, unet.py:1009-1022): this UNet reads the
  // residual tensors of the attached ControlNet handles straight from HBM and sums them on the device
  void attach_controlnets(const std::vector<UNet*>& cns);
  void set_controlnet_cond(const void* cond, int flags);
  void set_attention(int impl);
  void vae_decode(const void* z, int z_is_f32, float* image, int flags);
  void vae_encode(const void* x, int x_is_f32, float* moments, int flags);
  int num_residuals() const { return (int)res_shapes_.size(); }
  size_t device_bytes() const { return arena_.bytes(); }
  const sd_unet_config& config() const { return cfg_; }

 private:
  // ---- build ----
  void build_unet();
  void build_vae_decoder();
  void build_vae_encoder();
  void run_vae_graph();
  Tensor new_tensor(int B, int H, int W, int C);
  half_t* upload_conv_weight(const std::string& name, int cout, int cin, int k, bool geglu);
  float* upload_vec(const std::string& name, int n, bool geglu = false);
  Tensor conv(std::vector<Op>& ops, const std::string& name, const Tensor& x, const Tensor* x2, int cout, int k,
              int stride, int up, bool bias, const float* temb, const half_t* res, int out_mode = kOutHalf,
              int ldT = 0, bool silu_out = false, int pad = -1);
  // optional extras of conv_w: LayerNorm fold (ln_colsum) and the fused q|k|v split (n_trans > 0:
  // columns >= n_trans leave token-transposed in *vt, [B][cout - n_trans][ldT])
  struct ConvExtra {
    const float* ln_colsum = nullptr;
    int n_trans = 0;
    half_t* vt = nullptr;   // out
    bool vt_perm = false;   // V^T in attention8's key order (AttnDesc::vt_perm)
    float q_scale = 1.f;    // fused q|k|v: the first q_cols columns leave pre-scaled for attention8 (ConvDesc::q_scale)
    int q_cols = 0;
  };
  Tensor conv_w(std::vector<Op>& ops, const std::string& name, const half_t* w, const float* bias, const Tensor& x,
                const Tensor* x2, int cout, int k, int stride, int up, const float* temb, const half_t* res,
                int out_mode, int ldT, bool silu_out, ConvExtra* ex = nullptr, int pad = -1);
  // LayerNorm `ln` folded into the bias-free/biased 1x1 projections `names` (stacked along Cout) that
  // consume it: returns w' = W*gamma (fp16), colsum of w', bias' = b + W.beta
  struct LnFold {
    half_t* w;
    float* colsum;
    float* bias;
  };
  LnFold fold_layernorm(const std::string& ln, const std::vector<std::string>& names, int cin, int cout_each,
                        bool geglu);
  bool can_fold_ln(const Tensor& x, int cout, bool geglu) const;
  Tensor conv_stacked(std::vector<Op>& ops, const std::vector<std::string>& names, const Tensor& x, int cout_each);
  // side (round 5): an independent 1x1 GEMM launched in the SAME grid as the GroupNorm's apply / single-launch kernel
  // (launch_groupnorm); side_label / side_flop describe it in the per-op profile
  Tensor group_norm(std::vector<Op>& ops, const std::string& name, const Tensor& x, const Tensor* x2, float eps,
                    bool silu, const ConvDesc* side = nullptr, const std::string& side_label = std::string(), double side_flop = 0);
  Tensor layer_norm(std::vector<Op>& ops, const std::string& name, const Tensor& x);
  Tensor resnet(std::vector<Op>& ops, const std::string& p, const Tensor& x, const Tensor* x2, int cout,
                bool has_temb = true);
  Tensor transformer(std::vector<Op>& ops, const std::string& p, const Tensor& x, int heads, int depth);
  // proj_out / tres (last block of a SpatialTransformer only): the transformer's proj_out and its residual - where the tail
  // ff.net.2 + residual -> proj_out + residual runs as ONE launch (xattn_out.hip ffn_proj_kernel) *tail_done is set and the returned
  // tensor is the transformer's output.  pre: the block's self-attention operands where the transformer's head launch already wrote them
  Tensor transformer_block(std::vector<Op>& ops, const std::string& b, const Tensor& h, int heads, const std::string* proj_out = nullptr,
                           const Tensor* tres = nullptr, bool* tail_done = nullptr, const PreQkv* pre = nullptr);
  Tensor vae_attention(std::vector<Op>& ops, const std::string& p, const Tensor& h);
  Tensor attention(std::vector<Op>& ops, const Tensor& q, const half_t* k, const half_t* vt, int heads, int Sq,
                   int Sk, int ldk, int ldv, int ldq, bool vt_perm = false, bool q_prescaled = false);
  // scratch of attention8's balanced form (AttnDesc::sk_part / sk_cnt), shared by this handle's self-attention launches
  float* sk_part_ = nullptr;
  size_t sk_part_bytes_ = 0;
  unsigned* sk_cnt_ = nullptr;
  int sk_cnt_n_ = 0;
  void down_and_mid(std::vector<Op>& ops, Tensor& h, std::vector<Tensor>& skips);
  const float* register_temb(const std::string& name, int cout);
  void finalize_temb();
  void upload_inputs(const sd_unet_io& io, bool loop_mode);
  void run_ops(const std::vector<Op>& ops);
  void run_ops_on(const std::vector<Op>& ops, hipStream_t s);
  // ControlNet handle driven by a UNet handle: sample / timesteps come from the UNet's device buffers,
  // everything runs on the UNet's stream (and inside its HIP graph)
  void run_as_controlnet(hipStream_t s, const half_t* x_nhwc, const float* tbuf);
  void set_context_from(const half_t* ehs_dev, hipStream_t s);
  void run_attached();
  void invalidate_graphs();
  hipGraphExec_t capture(const std::function<void()>& body);
  void run_forward_ops();
  void run_time_and_main();   // time_ops_ (optionally on the forked side stream) + main_ops_
  void run_main(bool with_time);   // main_ops_ [behind time_ops_] with the joins of the forked time path / ControlNets in place
  void ensure_graph();

  sd_unet_config cfg_;
  const WeightStore* ws_ = nullptr;   // only valid during construction
  int device_ = 0;
  bool f32_ = false;                  // VAE handle with cfg.compute_fp32: fp32 activations on the vae_f32.hip kernels
  // build time: the GroupNorm(+SiLU) in front of the next conv_w (UNet::gn_loader_conv): applied in the 3x3 kernel's halo loader when
  // the producer of the raw tensor left its statistics, else by a GroupNorm launch inside the same op (ConvDesc::gnf_* on ksize 3)
  struct GnIn {
    std::shared_ptr<GnHook> hook;   // producer of the raw input
    float* partial = nullptr;
    const float* gamma = nullptr;
    const float* beta = nullptr;
    half_t* y = nullptr;            // normalised tensor of the fallback path
    float eps = 1e-5f;
    int groups = 32, silu = 1;
    std::string norm_name;
  };
  std::shared_ptr<GnIn> pending_gn_in_;
  // norm -> SiLU -> 3x3 conv as ONE op; false when the shape / the build does not admit it (then the caller builds the two ops)
  bool gn_loader_conv(std::vector<Op>& ops, const std::string& norm, const std::string& cv, const Tensor& x, int cout, const float* temb,
                      const half_t* res, Tensor* out);
  bool gn_loader_ok(const std::vector<Op>& ops, const Tensor& x, int cout) const;
  bool w_f32_pending_ = false;        // build time: the weight pointer handed to conv_w holds fp32 values (UNet::conv on an fp32 handle)
  hipStream_t stream_ = nullptr;
  // SD_SIDE_TIME=1 (experiment, off by default): the time-embedding chain (depends only on the timestep)
  // runs on a forked stream beside conv_in / the first GroupNorm and joins before the first consumer of
  // its output.  Measured: the fork/join costs the captured step +0.22 ms while hiding 65 us (LAB_NOTES.md, round 4).
  hipStream_t side_ = nullptr;
  hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
  int temb_join_pos_ = -1;
  // Attached ControlNets run CONCURRENTLY with this UNet's time path and down / mid blocks (pipeline.py:519-529 evaluates them
  // back to back; nothing in the UNet depends on them before the residual adds of unet.py:1009-1022): they are forked onto
  // cn_stream_ behind the sample / timestep hand-over and joined in front of the first residual add (main_ops_[cn_join_pos_]) -
  // eagerly and inside the captured step graphs alike.  SD_CN_CONCURRENT=0 keeps the serial order (A/B).
  hipStream_t cn_stream_ = nullptr;
  hipEvent_t ev_cn_fork_ = nullptr, ev_cn_join_ = nullptr;
  int cn_join_pos_ = -1;
  bool cn_join_pending_ = false;
  Arena arena_;

  // static-shape input/output device buffers
  half_t* in_sample_ = nullptr;     // NCHW f16
  half_t* in_timestep_ = nullptr;   // (B,) f16
  half_t* in_ehs_ = nullptr;        // BC1S f16
  half_t* in_time_ids_ = nullptr;
  half_t* in_text_embeds_ = nullptr;
  half_t* in_cond_ = nullptr;       // ControlNet conditioning image NCHW f16
  std::vector<half_t*> in_res_nchw_;   // support_controlnet: residual inputs (NCHW f16)
  std::vector<Tensor> res_nhwc_;       // ... converted to NHWC
  std::vector<std::vector<int>> res_shapes_;   // (B,C,H,W) of each residual
  float* noise_pred_ = nullptr;     // NCHW f32
  float* in_z_ = nullptr;           // VAE: latent input NCHW f32
  half_t* z_half_ = nullptr;
  float* image_ = nullptr;          // VAE: decoded image NCHW f32 (encoder: the moments NCHW f32)
  void* in_x_ = nullptr;            // VAE encoder: input image NCHW (f16 or f32)
  int vae_in_f32_ = 0;
  size_t image_elems_ = 0;
  std::vector<float*> res_out_;     // ControlNet outputs NCHW f32
  std::vector<Tensor> cn_out_;      // ControlNet outputs NHWC f16

  Tensor x_in_;                     // NHWC sample
  float* tbuf_ = nullptr;           // (B,) f32 timesteps
  float* emb_ = nullptr;            // (B, temb_dim)
  float* temb_all_ = nullptr;       // (B, kTembCap)
  // device loop: time-embedding rows of every step of the current schedule (unconditioned time path only), and the
  // timesteps they were computed for
  float* temb_tab_ = nullptr;       // [steps][B][kTembCap]
  int temb_tab_cap_ = 0;            // in steps
  float *tt_tab_ = nullptr, *tsin_tab_ = nullptr, *e1_tab_ = nullptr, *emb_tab_ = nullptr;   // [steps * B] x {1, C0, tdim, tdim}
  struct TimePath {                 // the unconditioned time path's weights (unet.py:703-728), kept for the table pass
    const float* freq = nullptr;
    const half_t *w1 = nullptr, *w2 = nullptr;
    const float *b1 = nullptr, *b2 = nullptr;
    int c0 = 0, tdim = 0;
    bool ok = false;                // false: text_time conditioning enters the MLP (SDXL) -> in-step path only
  } tpath_;
  int temb_used_ = 0;
  std::vector<std::pair<std::string, int>> temb_layers_;
  half_t* temb_w_all_ = nullptr;
  float* temb_b_all_ = nullptr;
  Tensor ctx_;                      // encoder_hidden_states as tokens [B][L][Cctx]
  ConvWorkspace ws_conv_;
  size_t ws_need_ = 0;

  std::vector<Op> in_ops_, time_ops_, ctx_ops_, cond_ops_, main_ops_;
  std::vector<Op> out_ops_;          // ControlNet: fp16 NHWC residuals -> fp32 NCHW for the host boundary only
  std::vector<UNet*> attached_;      // UNet with support_controlnet: ControlNets whose outputs it consumes on device
  std::vector<uint16_t> last_ehs_, last_cond_;
  bool have_ctx_ = false, have_cond_ = false, have_inputs_ = false;
  hipGraphExec_t graph_ = nullptr, loop_graph_ = nullptr;
  int loop_graph_key_ = -1;

  // denoise-loop state
  float* latents_ = nullptr;
  float* eps_hist_ = nullptr;
  float* tab_timesteps_ = nullptr;
  float* tab_coef_ = nullptr;
  float* tab_scale_ = nullptr;
  int* step_ = nullptr;
  int tab_cap_ = 0;
  float* noise_tab_ = nullptr;      // ancestral samplers: per-step noise of the current call
  size_t noise_cap_ = 0;            // in floats
};

}  // namespace sd

struct sd_unet {
  std::unique_ptr<sd::UNet> impl;
};
