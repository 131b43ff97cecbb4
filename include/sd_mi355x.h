/* libsdmi355 - C ABI of the MI355X-native Stable Diffusion hot path.
 *
 * This is the drop-in boundary for the model-runner seam of apple/ml-stable-diffusion:
 *   python_coreml_stable_diffusion/coreml_model.py:36-120   CoreMLModel(**np.ndarray) -> dict
 *   swift/StableDiffusion/pipeline/Unet.swift:90-144        Unet.predictNoise
 *   swift/StableDiffusion/pipeline/ManagedMLModel.swift:11-127 (load / perform / unload)
 * Each entry point below names the reference interface it replaces.  Plain pointers and sizes
 * only; no torch / numpy types.  All functions return 0 on success or a negative sd_status and
 * leave a message retrievable with sd_last_error() (thread-local).  A handle is NOT thread-safe
 * (one HIP stream per handle, like the one serial DispatchQueue per model of
 * ManagedMLModel.swift:24-69); different handles may be driven from different threads.
 *
 * Tensor conventions at the boundary are the reference's: NCHW images, BC1S sequences, fp16
 * inputs, fp32 outputs (python_coreml_stable_diffusion/torch2coreml.py:135, :857-863).
 */
#ifndef SD_MI355X_H
#define SD_MI355X_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum sd_status {
  SD_OK = 0,
  SD_ERR_INVALID_ARGUMENT = -1, /* wrong shape/dtype/flag      -> TypeError / ValueError (coreml_model.py:97-116) */
  SD_ERR_NOT_FOUND = -2,        /* missing file / weight key   -> FileNotFoundError (coreml_model.py:176-178)     */
  SD_ERR_HIP = -3,              /* HIP runtime failure         -> RuntimeError                                     */
  SD_ERR_UNSUPPORTED = -4,      /* config outside the path     -> NotImplementedError (unet.py:835-880)            */
  SD_ERR_INTERNAL = -5
} sd_status;

typedef enum sd_dtype { SD_F16 = 0, SD_F32 = 1 } sd_dtype;

/* unet.py:33-36 AttentionImplementations; a per-handle field here instead of the reference's
 * module global unet.py:39 (conversion-time flag torch2coreml.py:1678-1685). */
typedef enum sd_attention_impl {
  SD_ATTN_ORIGINAL = 0,
  SD_ATTN_SPLIT_EINSUM = 1,
  SD_ATTN_SPLIT_EINSUM_V2 = 2
} sd_attention_impl;

const char* sd_last_error(void);
const char* sd_version(void);
/* number of visible HIP devices (<0: error).  The library never falls back to the CPU. */
int sd_device_count(void);

/* ------------------------------------------------------------------------------------------
 * Checkpoints.  Replaces `load_state_dict` of torch2coreml.py:917-918 + the two load hooks of
 * unet.py:121-146: tensors are handed over with their diffusers key names AS-IS (Linear weights
 * may be 2-D or 4-D; LayerNorm bias is the checkpoint's, not the reference's rewritten b/w).
 * ------------------------------------------------------------------------------------------ */
typedef struct sd_weights sd_weights;
int sd_weights_create(sd_weights** out);
/* copies `data` (host memory); shape has ndim entries */
int sd_weights_add(sd_weights* w, const char* name, const void* data, sd_dtype dtype, const int64_t* shape,
                   int ndim);
/* parse a .safetensors file (F16 / F32 / BF16 tensors); `prefix` (may be NULL) is stripped */
int sd_weights_load_safetensors(sd_weights* w, const char* path, const char* prefix);
int sd_weights_count(const sd_weights* w);
void sd_weights_destroy(sd_weights* w);

/* ------------------------------------------------------------------------------------------
 * UNet / ControlNet.  Config mirrors UNet2DConditionModel.__init__ (unet.py:801-833) with the
 * static shapes Core ML bakes in at conversion time (torch2coreml.py:827-863).
 * ------------------------------------------------------------------------------------------ */
#define SD_MAX_LEVELS 6

typedef struct sd_unet_config {
  int32_t batch;                 /* UNet batch (2 = classifier-free guidance for one image)          */
  int32_t in_channels, out_channels;
  int32_t height, width;         /* latent size (image / 8)                                          */
  int32_t n_levels;
  int32_t block_out_channels[SD_MAX_LEVELS];
  int32_t down_cross_attn[SD_MAX_LEVELS];   /* 1: CrossAttnDownBlock2D, 0: DownBlock2D               */
  int32_t up_cross_attn[SD_MAX_LEVELS];     /* 1: CrossAttnUpBlock2D,  0: UpBlock2D (up order)       */
  int32_t layers_per_block;
  int32_t attention_head_dim[SD_MAX_LEVELS];           /* = NUMBER of heads (unet.py:194-198, :912)  */
  int32_t transformer_layers_per_block[SD_MAX_LEVELS];
  int32_t cross_attention_dim;
  int32_t context_len;           /* 77                                                               */
  int32_t norm_num_groups;       /* 32                                                               */
  float norm_eps;                /* 1e-5                                                             */
  int32_t flip_sin_to_cos;       /* 1                                                                */
  float freq_shift;              /* 0                                                                */
  int32_t addition_time_embed_dim;                 /* 0: none; SDXL 256 (text_time)                  */
  int32_t projection_class_embeddings_input_dim;   /* SDXL 2816 / refiner 2560                       */
  int32_t num_time_ids;          /* SDXL base 6, refiner 5                                           */
  int32_t support_controlnet;    /* UNet: consume additional_residual_0..N (unet.py:1009-1022)       */
  int32_t is_controlnet;         /* build controlnet.py:49-250 instead of the UNet                   */
  int32_t is_vae_decoder;        /* build the AutoencoderKL decoder (sd_vae_decoder_create)          */
  int32_t attention_impl;        /* sd_attention_impl                                                */
  int32_t use_graph;             /* 1: capture the forward into a HIP graph and replay it            */
  int32_t compute_fp32;          /* VAE handles only: 1 = fp32 activations and arithmetic, the reference's
                                  * float32 VAE of SDXL (torch2coreml.py:570-578 decoder, :726-733 encoder: "z" / "x"
                                  * declared float32, pipeline.py:315 reads the dtype off the model); 0 = fp16 storage */
} sd_unet_config;

typedef struct sd_unet sd_unet;

/* replaces CoreMLModel.__init__ / _load_mlpackage (coreml_model.py:40-95, :155-203) and
 * ManagedMLModel.loadResources (ManagedMLModel.swift:40-47) */
int sd_unet_create(const sd_unet_config* cfg, const sd_weights* w, int device, sd_unet** out);
/* ManagedMLModel.unloadResources (ManagedMLModel.swift:49-52) */
void sd_unet_destroy(sd_unet* u);
int sd_unet_set_attention(sd_unet* u, int impl);
int sd_unet_num_residuals(const sd_unet* u); /* controlnet.py:191-197 */
/* bytes of HBM held by the handle (weights + activations) */
size_t sd_unet_device_bytes(const sd_unet* u);

#define SD_FLAG_DEVICE_PTRS 1 /* all data pointers are device pointers on the handle's GPU */

typedef struct sd_unet_io {
  const void* sample;                /* (B, in_channels, H, W) f16                                   */
  const void* timestep;              /* (B,) f16                                                     */
  const void* encoder_hidden_states; /* (B, cross_attention_dim, 1, context_len) f16 BC1S            */
  const void* time_ids;              /* SDXL: (B, num_time_ids) f16, else NULL                       */
  const void* text_embeds;           /* SDXL: (B, proj_dim - num_time_ids*addition_time_embed_dim) f16 */
  const void* controlnet_cond;       /* ControlNet: (B, 3, 8H, 8W) f16                               */
  const void* const* additional_residuals; /* UNet w/ support_controlnet: N pointers, f16 NCHW        */
  int32_t num_additional_residuals;
  void* noise_pred;                  /* UNet out: (B, out_channels, H, W) f32                        */
  void* const* residual_outputs;     /* ControlNet out: N pointers, f32 NCHW                         */
  int32_t flags;
  const float* step_noise;           /* sd_unet_denoise_loop only, optional (HOST pointer): (n_steps, n_images, C, H, W) f32
                                      * added to the latents at the end of step i - the fresh noise of the ancestral
                                      * samplers (EulerAncestralDiscrete, pipeline.py:592-604), already scaled by the
                                      * step's sigma_up, drawn by the host's seeded generator like the reference does */
} sd_unet_io;

/* replaces CoreMLModel.__call__ -> MLModel.predict (coreml_model.py:118-120; call site
 * pipeline.py:531-536) and Unet.predictNoise (Unet.swift:90-144).  Synchronous. */
int sd_unet_forward(sd_unet* u, const sd_unet_io* io);

/* Timing on the handle's own stream with HIP events: runs `iters` forwards on the inputs of
 * the last sd_unet_forward call and returns the mean milliseconds per forward. */
int sd_unet_time_forward(sd_unet* u, int warmup, int iters, float* ms_per_iter);

/* Measurement hook: HIP-event time of every launch-list entry ("op": one kernel, or a kernel plus its
 * split-K reduce) of one forward on the inputs of the last call, in launch order, eager launches (the
 * same dependent-kernel sequence the HIP graph replays), median over `iters` passes.  Fills up to `cap`
 * entries of ms / flop (algorithmic FLOP of MFMA ops, 0 for bandwidth ops) / labels (cap x label_bytes
 * chars, NUL-terminated) and always sets *n_ops to the number of ops.  No reference counterpart: it is
 * what bench.py derives the in-sequence roofline of the dominant kernel family from. */
int sd_unet_profile(sd_unet* u, int iters, int cap, float* ms, double* flop, char* labels, int label_bytes,
                    int* n_ops);

/* ---- debug ABI (tuning / race-guard tools): both entry points below return SD_ERR_UNSUPPORTED unless the process
 * runs with SD_TUNE=1 in its environment; nothing in the product path calls them. ----
 * Tuning hook of the same kind: while tile != 0, every convolution / GEMM whose constraints admit it runs with
 * this plan (tile 1-6, LDS-DMA ring code 0-5, split-K; csrc/igemm.hip) instead of the table's, so ONE profiled
 * forward measures a candidate on every layer shape in sequence (tools/tune_plans.py).  tile = 0 switches it off.
 * Process-global; never set in production. */
int sd_tune_set_candidate(int tile, int staging, int splitk);
/* Measurement hook: replace the run-time plan table (consulted before the compiled-in one; same rows as
 * csrc/tuned_convs.inc, one "{kind, ksize, stride, up, Ctot, N, M, tile, staging, splitk}" per line; NULL = empty) and, when
 * `u` is given, drop its captured HIP graphs so that the next forward re-captures with the new plans.  tools/tune_e2e.py
 * uses it to judge a plan by the graph-replay time of the WHOLE step instead of a per-kernel timing.  Process-global. */
int sd_tune_set_plan_table(const char* rows, sd_unet* u, int* n_plans);

/* Device-resident denoising loop (pipeline.py:500-573 with latents, CFG combine and scheduler
 * update never leaving HBM; Swift twin StableDiffusionPipeline.swift:233-333 incl. imageCount > 1).
 * latents: (n_images, C, H, W) f32 host in/out; the UNet batch must be cfg * n_images with
 * cfg = (guidance_scale > 1 ? 2 : 1) (pipeline.py:443), batch order [uncond..., cond...] (:245).
 * `timesteps` (n_steps) and `coef` (n_steps x 8) come from the host-side scheduler tables; every
 * scheduler on the path (DDIM, PNDM/PLMS Scheduler.swift:137-344, DPM-Solver++ 2M
 * DPMSolverMultistepScheduler.swift:27-273) is the linear multistep rule
 *     eps = u + g*(c - u);  m = a*x + b*eps;  x <- cx*x + cm*m + sum_{j<history} ch[j]*hist[j]
 * with coef row = [cx, cm, ch0, ch1, ch2, a, b, flags]; hist[j] is the m of j+1 pushes ago and
 * flags != 0 keeps this step's m out of the history (PLMS warm-up).  history <= 3.
 * `sample_scale` (may be NULL): n_steps factors of scheduler.scale_model_input (pipeline.py:504-505;
 * 1/sqrt(sigma^2+1) for the sigma-space Euler / LMS schedulers), applied to the UNet input only.
 * `history_io` (may be NULL): (history, n_images, C, H, W) f32 host buffer read before the first
 * step and written after the last, so a loop can continue on another handle (SDXL base -> refiner
 * hand-off, StableDiffusionXLPipeline.swift:205-225).  The encoder_hidden_states / SDXL extras are
 * taken from `io`; ControlNet residuals come from the attached ControlNet handles (below).
 * ms_per_step (may be NULL) receives HIP-event time per loop iteration. */
int sd_unet_denoise_loop(sd_unet* u, const sd_unet_io* io, float* latents, int n_images, int n_steps,
                         const float* timesteps, const float* coef, const float* sample_scale, int history,
                         float guidance_scale, float* history_io, float* ms_per_step);

/* ControlNet residuals on the device (replaces the host round trip of pipeline.py:259-284 / :519-529
 * and ControlNet.swift:64-118): `u` (built with support_controlnet) runs the n <= 3 attached
 * ControlNet handles on its own stream before every forward - same sample / timestep /
 * encoder_hidden_states - reads their 13 fp16 residual tensors straight from HBM, sums them over
 * the ControlNets (pipeline.py:269-282) and adds them to its skip / mid tensors (unet.py:1009-1022).
 * With ControlNets attached, sd_unet_forward ignores io->additional_residuals and
 * sd_unet_denoise_loop runs ControlNet + UNet every step.  n = 0 detaches.  The ControlNet handles
 * must outlive the attachment and share the UNet's device and static shapes. */
int sd_unet_attach_controlnets(sd_unet* u, sd_unet* const* controlnets, int n);
/* the conditioning image of an attached ControlNet, (B, 3, 8H, 8W) f16 (pipeline.py:346-357); its
 * embedding (controlnet.py:211-215) is computed once here, not every step */
int sd_controlnet_set_cond(sd_unet* controlnet, const void* controlnet_cond, int flags);

/* ------------------------------------------------------------------------------------------
 * VAE decoder: image = decoder(post_quant_conv(z)) in [-1, 1] (torch2coreml.py:584-594; call
 * site pipeline.py:313-320 `vae_decoder(z=latents / 0.18215)["image"]`; Decoder.swift).  The
 * config reuses sd_unet_config: block_out_channels = the VAE's (128,256,512,512), layers_per_block
 * = 2, in_channels = 4, out_channels = 3, height/width = latent size, norm_num_groups = 32.
 * z: (B, 4, h, w) f16 or f32 (z_dtype); image: (B, 3, 8h, 8w) f32.
 * ------------------------------------------------------------------------------------------ */
int sd_vae_decoder_create(const sd_unet_config* cfg, const sd_weights* w, int device, sd_unet** out);
int sd_vae_decode(sd_unet* vae, const void* z, sd_dtype z_dtype, float* image, int flags);
/* VAE encoder: latent = quant_conv(encoder(x)) (torch2coreml.py:739-749 `vae_encoder`, output "latent";
 * Encoder.swift:48-90 samples mean + std * noise from it and multiplies by the scale factor).  Config reuses
 * sd_unet_config: block_out_channels = the VAE's (128,256,512,512), layers_per_block = 2, in_channels = 3,
 * out_channels = 2 * latent channels (8), height/width = IMAGE size.  x: (B, 3, H, W) f16 or f32 in [-1, 1];
 * moments: (B, 8, H/8, W/8) f32 = [mean | logvar]. */
int sd_vae_encoder_create(const sd_unet_config* cfg, const sd_weights* w, int device, sd_unet** out);
int sd_vae_encode(sd_unet* vae, const void* x, sd_dtype x_dtype, float* moments, int flags);

/* ------------------------------------------------------------------------------------------
 * CLIP text encoder(s): transformers' CLIPTextModel / CLIPTextModelWithProjection as the reference
 * wraps them for conversion (torch2coreml.py:379-441, causal mask -1e4 :363-377) and calls them
 * (pipeline.py:151-175, `text_encoder(input_ids=...)`; TextEncoder.swift / TextEncoderXL.swift).
 * One prompt per call: input_ids (1, max_position_embeddings) int32.  Outputs (each may be NULL), f32:
 *   last_hidden_state (1, S, D)  final_layer_norm(last layer)          - SD 1.x / 2.x context
 *   hidden_embeds     (1, S, D)  hidden_states[-2], no final LN        - SDXL context (:416-428)
 *   pooled            (1, projection_dim) text_projection(final_LN(last)[eos_index]) when
 *                     projection_dim > 0 ("text_embeds"), else (1, D) pooler_output = final_LN(last)[eos_index]
 * Weights use the transformers key names (text_model.embeddings.*, text_model.encoder.layers.N.*,
 * text_model.final_layer_norm.*, text_projection.weight).
 * ------------------------------------------------------------------------------------------ */
typedef struct sd_text_encoder_config {
  int32_t vocab_size, hidden_size, intermediate_size, num_hidden_layers, num_attention_heads;
  int32_t max_position_embeddings; /* 77 */
  int32_t hidden_act;              /* 0: quick_gelu (CLIP ViT-L), 1: gelu (OpenCLIP H / bigG) */
  int32_t projection_dim;          /* 0: CLIPTextModel, > 0: CLIPTextModelWithProjection */
  float layer_norm_eps;            /* 1e-5 */
  int32_t use_graph;
} sd_text_encoder_config;
typedef struct sd_text_encoder sd_text_encoder;
int sd_text_encoder_create(const sd_text_encoder_config* cfg, const sd_weights* w, int device, sd_text_encoder** out);
void sd_text_encoder_destroy(sd_text_encoder* t);
size_t sd_text_encoder_device_bytes(const sd_text_encoder* t);
int sd_text_encoder_encode(sd_text_encoder* t, const int32_t* input_ids, int eos_index, float* last_hidden_state,
                           float* hidden_embeds, float* pooled);

/* ------------------------------------------------------------------------------------------
 * Operator-level entry points (what the parity tests and micro-benchmarks bind).  Host
 * pointers unless SD_FLAG_DEVICE_PTRS; each call is synchronous on an internal stream.
 * `ms` (may be NULL) returns HIP-event kernel time averaged over `iters` (>=1) launches.
 * ------------------------------------------------------------------------------------------ */
/* attention.py:24-168.  q (B, h*d, 1, Sq), k/v (B, h*d, 1, Sk) f16 BC1S -> out (B, h*d, 1, Sq) f16.
 * variant (A/B testing): 0 = default dispatch, 1 = never the software-pipelined d = 64 kernel, 2 = q arrives multiplied by
 * d^-0.5 * log2(e) (how the UNet's fused q|k|v GEMM hands its queries to that kernel: scaled in fp32, rounded once),
 * 100 + u = that kernel's balanced grid with u (query tile, key tile) units per workgroup (0: the launch's own split). */
int sd_op_attention(int impl, const void* q, const void* k, const void* v, void* out, int B, int heads, int d,
                    int Sq, int Sk, int variant, int iters, float* ms);
/* layer_norm.py:51-80.  x (B, C, 1, S) f16, weight/bias (C) f32 -> out (B, C, 1, S) f16 */
int sd_op_layernorm(const void* x, const float* weight, const float* bias, void* out, int B, int C, int S, float eps,
                    int iters, float* ms);
/* torch.nn.GroupNorm (+ optional SiLU) as used by unet.py:430-451,:472-481,:528-531.  NCHW f16 */
int sd_op_groupnorm(const void* x, const float* weight, const float* bias, void* out, int B, int C, int H, int W,
                    int groups, float eps, int silu, int iters, float* ms);
/* The two independent first steps of a ResnetBlock2D with a channel change (unet.py:470-489): norm1 (+ SiLU) over the channel concat
 * (x0 | x1) (x1 may be NULL; the up blocks' torch.cat of unet.py:213-216) and conv_shortcut, a 1x1 conv over the same concat.
 * side = 1: both in ONE launch (the GroupNorm's blocks and the GEMM's tiles share a grid); side = 0: two launches.  The results are
 * bit-identical.  x0 (B,C0,H,W), x1 (B,C1,H,W) f16 NCHW, gn_weight / gn_bias (C0+C1) f32, w (N, C0+C1) f16, bias (N) f32 or NULL
 * -> out_gn (B, C0+C1, H, W), out_sc (B, N, H, W) f16 NCHW. */
int sd_op_groupnorm_shortcut(const void* x0, const void* x1, const float* gn_weight, const float* gn_bias, const void* w, const float* bias,
                             void* out_gn, void* out_sc, int B, int C0, int C1, int H, int W, int N, int groups, float eps, int silu,
                             int side, int iters, float* ms);
/* nn.Conv2d as used by unet.py (k in {1,3}, stride in {1,2}, padding k/2), optional nearest x2
 * upsample before the conv (unet.py:498-500), optional residual add.  x (B,Cin,H,W) f16 NCHW,
 * w (Cout,Cin,k,k) f16, bias (Cout) f32 or NULL, res (B,Cout,Ho,Wo) f16 or NULL -> out f16 NCHW.
 * tile/splitk: 0 = heuristic (tuning hooks). force_generic: 1 = direct non-MFMA kernel. */
int sd_op_conv2d(const void* x, const void* w, const float* bias, const void* res, void* out, int B, int Cin, int H,
                 int W, int Cout, int ksize, int stride, int upsample, int tile, int splitk, int force_generic,
                 int iters, float* ms);
/* The same conv followed by torch.nn.GroupNorm (+ SiLU) of its output (unet.py:470-489 conv -> norm -> SiLU; stride 1, no
 * upsample): with producer_stats = 1 the GroupNorm statistics come out of the conv kernel's own epilogue (one launch less per
 * GroupNorm), with 0 from the GroupNorm's own statistics pass.  *entries (may be NULL) returns the number of partial
 * (sum, sumsq) entries per (sample, group) the conv wrote - 0 when its plan cannot produce them.  conv_out (may be NULL) receives
 * the conv result, out the normalised tensor; both (B, Cout, H, W) f16 NCHW. */
int sd_op_conv2d_groupnorm(const void* x, const void* w, const float* bias, const void* res, const float* gn_weight,
                           const float* gn_bias, void* conv_out, void* out, int B, int Cin, int H, int W, int Cout, int ksize,
                           int groups, float eps, int silu, int tile, int producer_stats, int* entries, int iters, float* ms);
/* conv -> torch.nn.GroupNorm (no SiLU) -> 1x1 projection: a resnet's last conv followed by SpatialTransformer.norm + proj_in
 * (unet.py:528-531 eps 1e-6, :553-556).  fold = 1: the GroupNorm is applied inside the projection GEMM (statistics from the
 * conv's epilogue, per-channel mean / scale / shift applied to the GEMM's activation fragments: no GroupNorm launch);
 * fold = 0: GroupNorm launch + plain GEMM.  *entries (may be NULL) returns the number of statistics entries the fold consumed
 * (0: it fell back to the GroupNorm launch because the conv's plan wrote none).  proj_w (Nproj, Cout) f16, proj_bias (Nproj)
 * f32 or NULL; conv_out (may be NULL) (B, Cout, H, W), out (B, Nproj, H, W) f16 NCHW. */
int sd_op_conv2d_groupnorm_proj(const void* x, const void* w, const float* bias, const void* res, const float* gn_weight,
                                const float* gn_bias, const void* proj_w, const float* proj_bias, void* conv_out, void* out, int B,
                                int Cin, int H, int W, int Cout, int ksize, int Nproj, int groups, float eps, int fold, int tile,
                                int* entries, int iters, float* ms);
/* conv -> torch.nn.GroupNorm (+ SiLU) -> 3x3 conv (stride 1, padding 1): a ResnetBlock2D's norm2 -> SiLU -> conv2 behind conv1, or
 * norm1 -> SiLU -> conv1 behind the previous block (unet.py:470-489).  fold = 1: the GroupNorm (+ SiLU) is applied in the HALO
 * LOADER of the second conv (statistics from the first conv's epilogue; every wave normalises the halo pieces it fetched, in LDS;
 * pixels outside the image stay zero - the conv pads the NORMALISED tensor): no GroupNorm launch, no round trip of the normalised
 * tensor; fold = 0: GroupNorm launch + plain conv.  *entries (may be NULL): statistics entries the loader consumed (0: it fell
 * back).  w2 (N2, Cout, 3, 3) f16, bias2 (N2) f32 or NULL, res2 (B, N2, H, W) f16 or NULL; staging2: ring code of the second conv
 * (0 = plan table).  conv_out (may be NULL) (B, Cout, H, W), out (B, N2, H, W) f16 NCHW. */
int sd_op_conv2d_groupnorm_conv3x3(const void* x, const void* w, const float* bias, const void* res, const float* gn_weight,
                                   const float* gn_bias, const void* w2, const float* bias2, const void* res2, void* conv_out, void* out,
                                   int B, int Cin, int H, int W, int Cout, int ksize, int N2, int groups, float eps, int silu, int fold,
                                   int tile, int staging2, int* entries, int iters, float* ms);
/* The tail of a SpatialTransformer (unet.py:591 FeedForward.net.2 + residual, :561-563 proj_out + residual):
 * out = res2 + proj_out(res1 + ff.net.2(g) + b1) + b2.  fused = 1: ONE launch (32 tokens x 5 waves per workgroup, C = 320,
 * S % 32 == 0; the intermediate never goes to HBM), fused = 0: the two 1x1 GEMMs.  g (B, 4C, 1, S), res1 / res2 / out (B, C, 1, S) f16,
 * w1 (C, 4C), w2 (C, C) f16, b1 / b2 (C) f32.  gn_sums (may be NULL): (B, groups, 2) f32 - the GroupNorm statistics (sum, sum of
 * squares per sample and group) the launch left for a consuming GroupNorm, folded; NaN when it left none. */
int sd_op_ffn_out_proj(const void* g, const void* w1, const float* b1, const void* res1, const void* w2, const float* b2, const void* res2,
                       void* out, float* gn_sums, int B, int C, int S, int groups, int fused, int iters, float* ms);
/* Cross-attention front half as one launch (unet.py:87-118 inside :586-591): out = softmax(to_q(LayerNormANE(x)) k^T / 8) v
 * per head, head dim 64, Sk <= 96 (the prompt), any Sq >= 1 (ragged last token tile).  V^T columns [Sk, round_up(Sk, 8)) must be
 * zero (this entry point zero-fills them; the masked probabilities there are 0 but 0 * inf would be NaN).  x (B, heads*64, 1, Sq), k / v (B, heads*64, 1, Sk) f16 BC1S,
 * ln_weight / ln_bias (C) f32 as in the checkpoint (x_hat * w + b), wq (C, C) f16 -> out (B, C, 1, Sq) f16.  nst: LDS-DMA
 * ring depth 2-4 (0 = heuristic). */
int sd_op_cross_attention_fused(const void* x, const float* ln_weight, const float* ln_bias, const void* wq, const void* k,
                                const void* v, void* out, int B, int heads, int Sq, int Sk, float eps, int nst, int iters,
                                float* ms);
/* The whole cross-attention branch of a BasicTransformerBlock (unet.py:586-591 around :87-118):
 * out = x + to_out(softmax(to_q(LayerNormANE(x)) k^T / 8) v) + bo.  fused = 1: ONE launch (xattn_out.hip: 32 tokens x all heads per
 * workgroup; 5 or 10 heads of 64, Sq % 32 == 0), fused = 0: the fused q-projection + attention launch followed by the to_out GEMM
 * with its residual epilogue.  Layouts as sd_op_cross_attention_fused; wo (C, C) f16, bo (C) f32.
 * a1 != NULL (with wo1, bo1): the SELF-attention's output projection in front (unet.py:588): the branch runs on
 * h1 = x + to_out1(a1) + bo1 (a1 the self-attention's output, x the block input) and out = h1 + to_out(...) + bo; the one-launch
 * form (5 heads only) never stores h1. */
int sd_op_cross_attention_block(const void* x, const float* ln_weight, const float* ln_bias, const void* wq, const void* k, const void* v,
                                const void* wo, const float* bo, const void* a1, const void* wo1, const float* bo1, void* out, int B, int heads,
                                int Sq, int Sk, float eps, int fused, int iters, float* ms);
/* GEGLU feed-forward first half (unet.py:609-617): x (M, C) f16, w (8C', C) f16, bias (8C'/.. ) */
int sd_op_geglu(const void* x, const void* w, const float* bias, void* out, int M, int C, int N2, int iters, float* ms);
/* The same projection with the LayerNorm in front of it folded into the GEMM, as the UNet runs it (unet.py:583-591 norm3 ->
 * :609-617; host-side fold of UNet::fold_layernorm): x (M, C) f16 UN-normalised, ln_weight / ln_bias (C) f32 or both NULL,
 * w (N2, C) f16, bias (N2) f32 or NULL -> out (M, N2 / 2) f16.  kernel: 0 = the library's plan, 1 = the tiled GEMM kernels,
 * 2 = the weight-stationary kernel (wsgemm.hip: C = 320, N2 % 256 == 0, M >= 2048; other shapes -> SD_ERR_INVALID_ARGUMENT). */
int sd_op_geglu_ln(const void* x, const float* ln_weight, const float* ln_bias, const void* w, const float* bias, void* out, int M, int C,
                   int N2, float eps, int kernel, int iters, float* ms);
/* Fused q|k|v projection of self-attention with norm1 folded in (unet.py:583-586 -> :74-84 as ONE GEMM): x (B * HW, C) f16,
 * ln_weight / ln_bias (C) f32, w (3C, C) f16 = [Wq | Wk | Wv] -> out_qk (B * HW, 2C) f16 with the queries multiplied by q_scale
 * before the rounding, out_vt (B, C, HW) f16 = V^T (vt_perm != 0: the two middle 4-token blocks of every 16 tokens swapped, the key
 * order of the d = 64 attention kernel).  kernel: 0 = the library's plan, 1 = the tiled kernels, 3 = bvgemm.hip, 4 + v = its variant v + 1. */
int sd_op_qkv_ln(const void* x, const float* ln_weight, const float* ln_bias, const void* w, void* out_qk, void* out_vt, int B, int HW, int C,
                 float eps, float q_scale, int vt_perm, int kernel, int iters, float* ms);
/* The head of a SpatialTransformer (unet.py:553-556 norm -> proj_in; :583-586 norm1 -> :74-84 fused to_q | to_k | to_v) behind a 1x1 conv
 * x = conv_w . x_in that leaves the GroupNorm statistics of x in its epilogue, as the resnet conv in front of it does in the UNet.
 * fused = 1: GroupNorm apply, proj_in, LayerNorm and the q|k|v projection in ONE launch (2 / 3: forced to 64 / 32 tokens per workgroup;
 * 64 needs H * W % 64 == 0); 0: the three launches it replaces.
 * x_in (B, C, H, W) f16; conv_w / proj_w (C, C) f16; gn_*, proj_bias, ln_* (C) f32; wqkv (3C, C) f16 -> out_h (B * H * W, C) = proj_in's
 * output, out_qk (B * H * W, 2C) with pre-scaled queries, out_vt (B, C, H * W) (vt_perm as sd_op_qkv_ln), all f16.  C = 320 only.
 * *entries: the producer's statistics entries per (sample, group) folded by the fused launch (0: fall-back GroupNorm launch). */
int sd_op_gn_proj_qkv(const void* x_in, const void* conv_w, const float* gn_weight, const float* gn_bias, const void* proj_w,
                      const float* proj_bias, const float* ln_weight, const float* ln_bias, const void* wqkv, void* out_h, void* out_qk,
                      void* out_vt, int B, int H, int W, int C, int groups, float gn_eps, float ln_eps, float q_scale, int vt_perm, int fused,
                      int* entries, int iters, float* ms);
/* unet.py:703-728 */
int sd_op_timestep_embedding(const float* t, float* out, int n, int dim, int flip_sin_to_cos, float freq_shift);
/* numpy legacy stream: np.random.seed(seed); np.random.randn(n) (pipeline.py:331,:726;
 * NumPyRandomSource.swift:28-102).  Host-side, bit-exact. */
int sd_numpy_randn(uint32_t seed, double* out, size_t n);
/* the other two seed-exact sources of the reference (RandomSource.swift): torch's CPU generator
 * (`torch.manual_seed(seed); torch.randn(n)`, TorchRandomSource.swift:116-150) and torch's CUDA generator
 * (Philox4x32-10, NvRandomSource.swift:25-80; `offset` = number of arrays drawn before).  Host-side. */
int sd_torch_randn(uint32_t seed, double* out, size_t n);
int sd_philox_randn(uint64_t seed, uint32_t offset, double* out, size_t n);
/* Box calibration for bench.py (no reference counterpart: measurement infrastructure).  out9[0] = device copy GB/s (1 GiB,
 * read + written bytes), out9[1] = dense v_mfma_f32_32x32x16_f16 register loop TFLOP/s, out9[2] = us per launch of a captured
 * graph of 323 empty launches (the step's launch count), out9[3] = us per launch of a 323-launch chain of short dependent
 * kernels on cold operands, out9[4] = us per launch of a 323-launch chain in which every workgroup reads 16 KB that a workgroup
 * on ANOTHER XCD wrote in the previous launch (8 MB handed over per launch), out9[5] / out9[6] = ns per dependent load from
 * never-touched HBM lines / from a 2-MB table resident in the caches, out9[7] = us per launch of a 323-launch chain of SMALL
 * grids (64 workgroups: 4 MB read, a reduction behind one barrier, 4 MB written - the shape of the launches that a slow box of the
 * pool runs 1.3-2 x slower - and which reads the same on them), out9[8] = COLD CODE: us per launch of a 320-launch chain that walks 32
 * different kernels of ~30 KB of code each minus the same chain repeating one of them - 0.8 us on the fast boxes of the pool, 11 us
 * on the slow ones: the figure `value_normalised` is built on.  Nine floats.  Allocates and frees 2 GiB of device memory; synchronous. */
int sd_calibrate(int device, float* out9);
/* MFMA fragment layout self-check used by the build/smoke tests (returns 0 when the hardware
 * layout matches what the kernels assume). */
int sd_selftest_mfma(void);

#ifdef __cplusplus
}
#endif
#endif /* SD_MI355X_H */
