// CLIP text encoder (transformers CLIPTextModel / CLIPTextModelWithProjection as the reference wraps
// them, torch2coreml.py:379-441; call site pipeline.py:151-175) as a static launch list of the gfx950
// kernels: embedding gather -> L x [LN1 -> q|k|v GEMM -> causal attention -> out_proj + residual ->
// LN2 -> fc1 -> activation -> fc2 + residual] -> final LN -> pooled row [-> text_projection].
// One prompt (1 x 77 tokens) per call like the converted Core ML model; runs once per prompt, not per step.
#include <cmath>
#include <cstring>

#include "../../include/sd_mi355x.h"
#include "kernels.h"
#include "weights.h"

namespace sd {

void launch_clip_embed(const int* ids, const half_t* tok, const half_t* pos, half_t* x, int S, int D, int vocab, hipStream_t s);
void launch_clip_attention(const half_t* qkv, half_t* out, int S, int D, int heads, hipStream_t s);
void launch_clip_act(half_t* x, size_t n, int act, hipStream_t s);

class TextEncoder {
 public:
  TextEncoder(const sd_text_encoder_config& cfg, const WeightStore& ws, int device);
  ~TextEncoder();
  void encode(const int32_t* input_ids, int eos_index, float* last_hidden_state, float* hidden_embeds, float* pooled);
  size_t device_bytes() const { return arena_.bytes(); }

 private:
  half_t* upload_matrix(const WeightStore& ws, const std::string& name, int rows, int cols);
  half_t* upload_stacked(const WeightStore& ws, const std::vector<std::string>& names, int rows_each, int cols);
  float* upload_vec(const WeightStore& ws, const std::vector<std::string>& names, int n_each);
  void gemm(const half_t* x, const half_t* w, const float* bias, const half_t* res, half_t* out, int N, int K);
  void run();

  sd_text_encoder_config cfg_;
  int device_ = 0;
  hipStream_t stream_ = nullptr;
  hipGraphExec_t graph_ = nullptr;
  Arena arena_;
  std::vector<std::function<void(hipStream_t)>> ops_;
  ConvWorkspace ws_conv_;
  size_t ws_need_ = 0;
  int* ids_ = nullptr;
  half_t* final_ = nullptr;      // final_layer_norm(last hidden)       [S][D]
  half_t* penult_ = nullptr;     // hidden_states[-2]                   [S][D]
  float* final_f32_ = nullptr;
  float* penult_f32_ = nullptr;
  float* pooled_ = nullptr;
  half_t* proj_w_ = nullptr;
};

TextEncoder::TextEncoder(const sd_text_encoder_config& cfg, const WeightStore& ws, int device) : cfg_(cfg), device_(device) {
  const int D = cfg.hidden_size, S = cfg.max_position_embeddings, L = cfg.num_hidden_layers, H = cfg.num_attention_heads;
  const int I = cfg.intermediate_size;
  SD_REQUIRE(D > 0 && S > 0 && L >= 1 && H > 0 && I > 0 && cfg.vocab_size > 0, kInvalidArgument, "bad text-encoder config");
  SD_REQUIRE(D % 64 == 0 && I % 64 == 0 && D % H == 0, kUnsupported, "text encoder: hidden %d / intermediate %d / heads %d", D,
             I, H);
  SD_REQUIRE(cfg.hidden_act == 0 || cfg.hidden_act == 1, kUnsupported, "hidden_act %d (0 quick_gelu, 1 gelu)", cfg.hidden_act);
  int ndev = 0;
  SD_HIP(hipGetDeviceCount(&ndev));
  SD_REQUIRE(ndev > 0, kHipError, "no HIP device visible: libsdmi355 has no CPU fallback");
  SD_REQUIRE(device >= 0 && device < ndev, kInvalidArgument, "device %d out of range (%d visible)", device, ndev);
  SD_HIP(hipSetDevice(device));
  SD_HIP(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking));
  const float eps = cfg.layer_norm_eps > 0 ? cfg.layer_norm_eps : 1e-5f;
  const std::string tm = ws.has("text_model.embeddings.token_embedding.weight") ? "text_model." : "";
  half_t* tok = upload_matrix(ws, tm + "embeddings.token_embedding.weight", cfg.vocab_size, D);
  half_t* pos = upload_matrix(ws, tm + "embeddings.position_embedding.weight", S, D);
  ids_ = arena_.alloc_n<int>(S);
  half_t* x = arena_.alloc_n<half_t>((size_t)S * D);
  {
    int* ids = ids_;
    const int vocab = cfg.vocab_size;
    ops_.push_back([=](hipStream_t s) { launch_clip_embed(ids, tok, pos, x, S, D, vocab, s); });
  }
  for (int l = 0; l < L; ++l) {
    const std::string p = tm + "encoder.layers." + std::to_string(l);
    if (l == L - 1) penult_ = x;                                  // hidden_states[-2] = input of the last layer
    const float* g1 = upload_vec(ws, {p + ".layer_norm1.weight"}, D);
    const float* b1 = upload_vec(ws, {p + ".layer_norm1.bias"}, D);
    half_t* n1 = arena_.alloc_n<half_t>((size_t)S * D);
    {
      const half_t* xi = x;
      ops_.push_back([=](hipStream_t s) { launch_layernorm(xi, g1, b1, n1, S, D, eps, s); });
    }
    // one stacked q|k|v projection (CLIPAttention q_proj / k_proj / v_proj, all with bias); the d^-0.5 scale
    // transformers applies to q is applied to the scores inside the attention kernel (same product)
    half_t* wqkv = upload_stacked(ws, {p + ".self_attn.q_proj.weight", p + ".self_attn.k_proj.weight", p + ".self_attn.v_proj.weight"}, D, D);
    float* bqkv = upload_vec(ws, {p + ".self_attn.q_proj.bias", p + ".self_attn.k_proj.bias", p + ".self_attn.v_proj.bias"}, D);
    half_t* qkv = arena_.alloc_n<half_t>((size_t)S * 3 * D);
    gemm(n1, wqkv, bqkv, nullptr, qkv, 3 * D, D);
    half_t* att = arena_.alloc_n<half_t>((size_t)S * D);
    ops_.push_back([=](hipStream_t s) { launch_clip_attention(qkv, att, S, D, H, s); });
    half_t* wo = upload_matrix(ws, p + ".self_attn.out_proj.weight", D, D);
    float* bo = upload_vec(ws, {p + ".self_attn.out_proj.bias"}, D);
    half_t* x1 = arena_.alloc_n<half_t>((size_t)S * D);
    gemm(att, wo, bo, x, x1, D, D);                                 // + residual
    const float* g2 = upload_vec(ws, {p + ".layer_norm2.weight"}, D);
    const float* b2 = upload_vec(ws, {p + ".layer_norm2.bias"}, D);
    half_t* n2 = arena_.alloc_n<half_t>((size_t)S * D);
    ops_.push_back([=](hipStream_t s) { launch_layernorm(x1, g2, b2, n2, S, D, eps, s); });
    half_t* w1 = upload_matrix(ws, p + ".mlp.fc1.weight", I, D);
    float* bb1 = upload_vec(ws, {p + ".mlp.fc1.bias"}, I);
    half_t* hmid = arena_.alloc_n<half_t>((size_t)S * I);
    gemm(n2, w1, bb1, nullptr, hmid, I, D);
    const int act = cfg.hidden_act;
    ops_.push_back([=](hipStream_t s) { launch_clip_act(hmid, (size_t)S * I, act, s); });
    half_t* w2 = upload_matrix(ws, p + ".mlp.fc2.weight", D, I);
    float* bb2 = upload_vec(ws, {p + ".mlp.fc2.bias"}, D);
    half_t* x2 = arena_.alloc_n<half_t>((size_t)S * D);
    gemm(hmid, w2, bb2, x1, x2, D, I);                              // + residual
    x = x2;
  }
  if (L == 1) penult_ = x;   // degenerate single-layer model: keep the pointer valid (hidden_states[-2] = embeddings)
  {
    const float* gf = upload_vec(ws, {tm + "final_layer_norm.weight"}, D);
    const float* bf = upload_vec(ws, {tm + "final_layer_norm.bias"}, D);
    final_ = arena_.alloc_n<half_t>((size_t)S * D);
    final_f32_ = arena_.alloc_n<float>((size_t)S * D);
    penult_f32_ = arena_.alloc_n<float>((size_t)S * D);
    const half_t* xl = x;
    half_t* fin = final_;
    float* f32 = final_f32_;
    float* p32 = penult_f32_;
    const half_t* pen = penult_;
    ops_.push_back([=](hipStream_t s) {
      launch_layernorm(xl, gf, bf, fin, S, D, eps, s);
      launch_half_to_float(fin, f32, (size_t)S * D, s);
      launch_half_to_float(pen, p32, (size_t)S * D, s);
    });
  }
  if (cfg.projection_dim > 0) {
    SD_REQUIRE(D % 8 == 0 && D <= 3072, kUnsupported, "text_projection: hidden size %d", D);
    proj_w_ = upload_matrix(ws, "text_projection.weight", cfg.projection_dim, D);
    pooled_ = arena_.alloc_n<float>(cfg.projection_dim);
  }
  if (ws_need_ > 0) {
    ws_conv_.partial = reinterpret_cast<float*>(arena_.alloc(ws_need_));
    ws_conv_.partial_bytes = ws_need_;
  }
  SD_HIP(hipStreamSynchronize(stream_));
}

TextEncoder::~TextEncoder() {
  (void)hipSetDevice(device_);
  if (graph_) (void)hipGraphExecDestroy(graph_);
  if (stream_) {
    (void)hipStreamSynchronize(stream_);
    (void)hipStreamDestroy(stream_);
  }
}

half_t* TextEncoder::upload_matrix(const WeightStore& ws, const std::string& name, int rows, int cols) {
  return upload_stacked(ws, {name}, rows, cols);
}

half_t* TextEncoder::upload_stacked(const WeightStore& ws, const std::vector<std::string>& names, int rows_each, int cols) {
  std::vector<half_t> host((size_t)names.size() * rows_each * cols);
  for (size_t i = 0; i < names.size(); ++i) {
    const HostTensor& t = ws.get(names[i]);
    SD_REQUIRE(t.numel() == (size_t)rows_each * cols, kInvalidArgument, "%s has %zu elements, expected %d x %d",
               names[i].c_str(), t.numel(), rows_each, cols);
    for (size_t j = 0; j < t.numel(); ++j) host[i * (size_t)rows_each * cols + j] = (half_t)t.data[j];
  }
  half_t* d = arena_.alloc_n<half_t>(host.size());
  SD_HIP(hipMemcpy(d, host.data(), host.size() * sizeof(half_t), hipMemcpyHostToDevice));
  return d;
}

float* TextEncoder::upload_vec(const WeightStore& ws, const std::vector<std::string>& names, int n_each) {
  std::vector<float> host;
  for (const auto& n : names) {
    const HostTensor& t = ws.get(n);
    SD_REQUIRE((int)t.numel() == n_each, kInvalidArgument, "%s has %zu elements, expected %d", n.c_str(), t.numel(), n_each);
    host.insert(host.end(), t.data.begin(), t.data.end());
  }
  float* d = arena_.alloc_n<float>(host.size());
  SD_HIP(hipMemcpy(d, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
  return d;
}

// out[S][N] = x[S][K] . w[N][K]^T + bias (+ res): the UNet's 1x1 implicit-GEMM kernel (M = 77 ragged rows)
void TextEncoder::gemm(const half_t* x, const half_t* w, const float* bias, const half_t* res, half_t* out, int N, int K) {
  ConvDesc d;
  d.x0 = x;
  d.C0 = K;
  d.w = w;
  d.bias = bias;
  d.res = res;
  d.out = out;
  d.B = 1;
  d.Hi = 1;
  d.Wi = cfg_.max_position_embeddings;
  d.Ho = 1;
  d.Wo = cfg_.max_position_embeddings;
  d.N = N;
  SD_REQUIRE(conv_fast_path_ok(d), kUnsupported, "text encoder GEMM %d x %d not MFMA-tileable", N, K);
  ws_need_ = std::max(ws_need_, conv_workspace_bytes(d));
  ops_.push_back([this, d](hipStream_t s) { launch_conv(d, ws_conv_, s); });
}

void TextEncoder::run() {
  for (auto& op : ops_) op(stream_);
}

void TextEncoder::encode(const int32_t* input_ids, int eos_index, float* last_hidden_state, float* hidden_embeds,
                         float* pooled) {
  SD_HIP(hipSetDevice(device_));
  const int S = cfg_.max_position_embeddings, D = cfg_.hidden_size;
  SD_REQUIRE(input_ids != nullptr, kInvalidArgument, "missing input 'input_ids'");
  SD_REQUIRE(eos_index >= 0 && eos_index < S, kInvalidArgument, "eos_index %d outside the %d-token prompt", eos_index, S);
  SD_HIP(hipMemcpyAsync(ids_, input_ids, (size_t)S * sizeof(int), hipMemcpyHostToDevice, stream_));
  if (cfg_.use_graph) {
    if (!graph_) {
      run();   // eager first: kernel attributes, code objects
      SD_HIP(hipStreamSynchronize(stream_));
      SD_HIP(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
      hipGraph_t g = nullptr;
      try {
        run();
      } catch (...) {
        (void)hipStreamEndCapture(stream_, &g);
        if (g) (void)hipGraphDestroy(g);
        throw;
      }
      SD_HIP(hipStreamEndCapture(stream_, &g));
      const hipError_t e = hipGraphInstantiate(&graph_, g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      SD_REQUIRE(e == hipSuccess, kHipError, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
    }
    SD_HIP(hipGraphLaunch(graph_, stream_));
  } else {
    run();
  }
  if (last_hidden_state)
    SD_HIP(hipMemcpyAsync(last_hidden_state, final_f32_, (size_t)S * D * sizeof(float), hipMemcpyDeviceToHost, stream_));
  if (hidden_embeds)
    SD_HIP(hipMemcpyAsync(hidden_embeds, penult_f32_, (size_t)S * D * sizeof(float), hipMemcpyDeviceToHost, stream_));
  if (pooled) {
    // pooler_output = final_layer_norm(last)[eos]; text_embeds = text_projection(pooler_output) (no bias)
    const float* row = final_f32_ + (size_t)eos_index * D;
    if (cfg_.projection_dim > 0) {
      launch_gemv(proj_w_, nullptr, row, D, pooled_, cfg_.projection_dim, 1, cfg_.projection_dim, D, 0, 0, 0, stream_);
      SD_HIP(hipMemcpyAsync(pooled, pooled_, (size_t)cfg_.projection_dim * sizeof(float), hipMemcpyDeviceToHost, stream_));
    } else {
      SD_HIP(hipMemcpyAsync(pooled, row, (size_t)D * sizeof(float), hipMemcpyDeviceToHost, stream_));
    }
  }
  SD_HIP(hipStreamSynchronize(stream_));
}

}  // namespace sd

struct sd_text_encoder {
  std::unique_ptr<sd::TextEncoder> impl;
};

namespace sd {
extern thread_local std::string g_last_error;
}

namespace {
template <typename F>
int guarded_te(F&& f) {
  try {
    f();
    sd::g_last_error.clear();
    return sd::kOk;
  } catch (const sd::Error& e) {
    sd::g_last_error = e.what();
    return e.code;
  } catch (const std::exception& e) {
    sd::g_last_error = e.what();
    return sd::kInternal;
  }
}
}  // namespace

extern "C" {

int sd_text_encoder_create(const sd_text_encoder_config* cfg, const sd_weights* w, int device, sd_text_encoder** out) {
  return guarded_te([&] {
    SD_REQUIRE(cfg && w && out, sd::kInvalidArgument, "NULL argument");
    auto h = std::make_unique<sd_text_encoder>();
    h->impl = std::make_unique<sd::TextEncoder>(*cfg, w->store, device);
    *out = h.release();
  });
}
void sd_text_encoder_destroy(sd_text_encoder* t) { delete t; }
size_t sd_text_encoder_device_bytes(const sd_text_encoder* t) { return t ? t->impl->device_bytes() : 0; }
int sd_text_encoder_encode(sd_text_encoder* t, const int32_t* input_ids, int eos_index, float* last_hidden_state,
                           float* hidden_embeds, float* pooled) {
  return guarded_te([&] {
    SD_REQUIRE(t && input_ids, sd::kInvalidArgument, "NULL argument");
    t->impl->encode(input_ids, eos_index, last_hidden_state, hidden_embeds, pooled);
  });
}

}  // extern "C"
