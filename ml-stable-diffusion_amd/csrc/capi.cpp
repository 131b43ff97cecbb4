// extern "C" surface of libsdmi355 (include/sd_mi355x.h).  Exceptions never cross the ABI:
// every entry point converts sd::Error into a status code + thread-local message.
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "unet.h"

namespace sd {
extern thread_local std::string g_last_error;
int selftest_mfma();   // selftest.hip

namespace {

template <typename F>
int guarded(F&& f) {
  try {
    f();
    g_last_error.clear();
    return kOk;
  } catch (const Error& e) {
    g_last_error = e.what();
    return e.code;
  } catch (const std::bad_alloc&) {
    g_last_error = "out of host memory";
    return kInternal;
  } catch (const std::exception& e) {
    g_last_error = e.what();
    return kInternal;
  }
}

void require_device() {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    fail(kHipError, "no HIP device visible (%s): libsdmi355 has no CPU fallback", hipGetErrorString(e));
}

// scoped device scratch for the operator-level entry points
struct Scratch {
  std::vector<void*> ptrs;
  hipStream_t stream = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  Scratch() {
    require_device();
    SD_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    SD_HIP(hipEventCreate(&e0));
    SD_HIP(hipEventCreate(&e1));
  }
  ~Scratch() {
    if (stream) (void)hipStreamSynchronize(stream);
    for (void* p : ptrs) (void)hipFree(p);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (stream) (void)hipStreamDestroy(stream);
  }
  template <typename T>
  T* dev(size_t n, const T* host = nullptr) {
    void* p = nullptr;
    SD_HIP(hipMalloc(&p, std::max<size_t>(n, 1) * sizeof(T)));
    ptrs.push_back(p);
    if (host)
      SD_HIP(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
    else
      SD_HIP(hipMemset(p, 0, std::max<size_t>(n, 1) * sizeof(T)));
    return reinterpret_cast<T*>(p);
  }
  template <typename F>
  void timed(int iters, float* ms, F&& launch) {
    if (iters < 1) iters = 1;
    static const bool poison = tune_env_set("SD_POISON_LDS");   // debug: NaN patterns into every CU's LDS in front of the launches
    if (poison) {   // (the result the caller reads is the last launch's)
      launch();     // sets kernel attributes
      launch_lds_poison(stream);
      launch();
      SD_HIP(hipStreamSynchronize(stream));
      if (ms) *ms = 0.f;
      return;
    }
    launch();   // warm (also sets kernel attributes)
    SD_HIP(hipStreamSynchronize(stream));
    // SD_BENCH_COLD=1: every timed launch starts with cold caches like a kernel inside the UNet step does (its
    // weights were last touched 1.7 GB of traffic ago): a 512-MiB fill between launches evicts the L2s and the
    // 256-MiB Infinity Cache; each launch gets its own event pair.  Default: back-to-back launches, operands warm.
    static const bool cold = tune_env_set("SD_BENCH_COLD");
    if (cold) {
      const size_t flush_bytes = (size_t)512 << 20;
      void* flush = dev<char>(flush_bytes);
      float total = 0.f;
      for (int i = 0; i < iters; ++i) {
        SD_HIP(hipMemsetAsync(flush, i & 0xff, flush_bytes, stream));
        SD_HIP(hipEventRecord(e0, stream));
        launch();
        SD_HIP(hipEventRecord(e1, stream));
        SD_HIP(hipEventSynchronize(e1));
        float t = 0.f;
        SD_HIP(hipEventElapsedTime(&t, e0, e1));
        total += t;
      }
      if (ms) *ms = total / (float)iters;
      return;
    }
    SD_HIP(hipEventRecord(e0, stream));
    for (int i = 0; i < iters; ++i) launch();
    SD_HIP(hipEventRecord(e1, stream));
    SD_HIP(hipEventSynchronize(e1));
    float t = 0.f;
    SD_HIP(hipEventElapsedTime(&t, e0, e1));
    if (ms) *ms = t / (float)iters;
  }
};

// (B, C, H, W) -> (B, H, W, C) on the host
std::vector<half_t> nchw_to_nhwc(const half_t* src, int B, int C, int H, int W) {
  std::vector<half_t> out((size_t)B * C * H * W);
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int p = 0; p < H * W; ++p) out[((size_t)b * H * W + p) * C + c] = src[((size_t)b * C + c) * H * W + p];
  return out;
}
void nhwc_to_nchw(const half_t* src, half_t* dst, int B, int C, int H, int W) {
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int p = 0; p < H * W; ++p) dst[((size_t)b * C + c) * H * W + p] = src[((size_t)b * H * W + p) * C + c];
}

}  // namespace
}  // namespace sd

using namespace sd;

extern "C" {

const char* sd_last_error(void) { return g_last_error.c_str(); }
const char* sd_version(void) { return "libsdmi355 0.1 (gfx950)"; }

int sd_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    g_last_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
    return kHipError;
  }
  return n;
}

int sd_weights_create(sd_weights** out) {
  return guarded([&] {
    SD_REQUIRE(out != nullptr, kInvalidArgument, "out is NULL");
    *out = new sd_weights();
  });
}
int sd_weights_add(sd_weights* w, const char* name, const void* data, sd_dtype dtype, const int64_t* shape, int ndim) {
  return guarded([&] {
    SD_REQUIRE(w && name && data && shape, kInvalidArgument, "NULL argument");
    w->store.add(name, data, (int)dtype, shape, ndim);
  });
}
int sd_weights_load_safetensors(sd_weights* w, const char* path, const char* prefix) {
  return guarded([&] {
    SD_REQUIRE(w && path, kInvalidArgument, "NULL argument");
    w->store.load_safetensors(path, prefix ? prefix : "");
  });
}
int sd_weights_count(const sd_weights* w) { return w ? (int)w->store.size() : 0; }
void sd_weights_destroy(sd_weights* w) { delete w; }

int sd_unet_create(const sd_unet_config* cfg, const sd_weights* w, int device, sd_unet** out) {
  return guarded([&] {
    SD_REQUIRE(cfg && w && out, kInvalidArgument, "NULL argument");
    require_device();
    auto h = std::make_unique<sd_unet>();
    h->impl = std::make_unique<UNet>(*cfg, w->store, device);
    *out = h.release();
  });
}
void sd_unet_destroy(sd_unet* u) { delete u; }
int sd_unet_set_attention(sd_unet* u, int impl) {
  return guarded([&] {
    SD_REQUIRE(u, kInvalidArgument, "NULL handle");
    u->impl->set_attention(impl);
  });
}
// Debug ABI: process-global plan overrides exist for the tuning / race-guard tools only and are refused unless the
// process runs with SD_TUNE=1 (the product path has no global mutable state).
static void require_tune_env() {
  static const bool on = getenv("SD_TUNE") != nullptr;
  SD_REQUIRE(on, kUnsupported, "sd_tune_* is a debug ABI: set SD_TUNE=1 in the environment to enable it");
}

int sd_tune_set_plan_table(const char* rows, sd_unet* u, int* n_plans) {
  return guarded([&] {
    require_tune_env();
    const int n = conv_plan_table_set(rows);
    if (n_plans) *n_plans = n;
    if (u) u->impl->drop_graphs();   // the captured launches bake the old plans in
  });
}
int sd_unet_num_residuals(const sd_unet* u) { return u ? u->impl->num_residuals() : 0; }
size_t sd_unet_device_bytes(const sd_unet* u) { return u ? u->impl->device_bytes() : 0; }

int sd_unet_forward(sd_unet* u, const sd_unet_io* io) {
  return guarded([&] {
    SD_REQUIRE(u && io, kInvalidArgument, "NULL argument");
    u->impl->forward(*io);
  });
}
int sd_unet_time_forward(sd_unet* u, int warmup, int iters, float* ms_per_iter) {
  return guarded([&] {
    SD_REQUIRE(u && ms_per_iter, kInvalidArgument, "NULL argument");
    *ms_per_iter = u->impl->time_forward(warmup, iters);
  });
}
int sd_unet_denoise_loop(sd_unet* u, const sd_unet_io* io, float* latents, int n_images, int n_steps,
                         const float* timesteps, const float* coef, const float* sample_scale, int history,
                         float guidance_scale, float* history_io, float* ms_per_step) {
  return guarded([&] {
    SD_REQUIRE(u && io && latents && timesteps && coef, kInvalidArgument, "NULL argument");
    u->impl->denoise_loop(*io, latents, n_images, n_steps, timesteps, coef, sample_scale, history, guidance_scale,
                          history_io, ms_per_step);
  });
}

int sd_tune_set_candidate(int tile, int staging, int splitk) {
  return guarded([&] {
    require_tune_env();
    SD_REQUIRE(tile >= 0 && tile <= 9 && staging >= 0 && staging <= 8 && splitk >= 0 && splitk <= 64, kInvalidArgument,
               "tune candidate (tile %d, staging %d, splitk %d)", tile, staging, splitk);
    conv_tune_set_candidate(tile, staging, splitk);
  });
}

int sd_unet_profile(sd_unet* u, int iters, int cap, float* ms, double* flop, char* labels, int label_bytes, int* n_ops) {
  return guarded([&] {
    SD_REQUIRE(u && n_ops && cap >= 0 && (cap == 0 || (ms && flop && labels && label_bytes > 1)), kInvalidArgument,
               "NULL argument");
    const std::vector<OpTime> t = u->impl->profile(iters);
    *n_ops = (int)t.size();
    for (int i = 0; i < (int)t.size() && i < cap; ++i) {
      ms[i] = t[i].ms;
      flop[i] = t[i].flop;
      std::snprintf(labels + (size_t)i * label_bytes, (size_t)label_bytes, "%s", t[i].label.c_str());
    }
  });
}

int sd_unet_attach_controlnets(sd_unet* u, sd_unet* const* controlnets, int n) {
  return guarded([&] {
    SD_REQUIRE(u && n >= 0 && (n == 0 || controlnets), kInvalidArgument, "NULL argument");
    std::vector<UNet*> v;
    for (int i = 0; i < n; ++i) {
      SD_REQUIRE(controlnets[i], kInvalidArgument, "controlnets[%d] is NULL", i);
      v.push_back(controlnets[i]->impl.get());
    }
    u->impl->attach_controlnets(v);
  });
}

int sd_controlnet_set_cond(sd_unet* cn, const void* controlnet_cond, int flags) {
  return guarded([&] {
    SD_REQUIRE(cn && controlnet_cond, kInvalidArgument, "NULL argument");
    cn->impl->set_controlnet_cond(controlnet_cond, flags);
  });
}

int sd_vae_decoder_create(const sd_unet_config* cfg, const sd_weights* w, int device, sd_unet** out) {
  return guarded([&] {
    SD_REQUIRE(cfg && w && out, kInvalidArgument, "NULL argument");
    require_device();
    sd_unet_config c = *cfg;
    c.is_vae_decoder = 1;
    c.is_controlnet = 0;
    auto h = std::make_unique<sd_unet>();
    h->impl = std::make_unique<UNet>(c, w->store, device);
    *out = h.release();
  });
}
int sd_vae_encoder_create(const sd_unet_config* cfg, const sd_weights* w, int device, sd_unet** out) {
  return guarded([&] {
    SD_REQUIRE(cfg && w && out, kInvalidArgument, "NULL argument");
    require_device();
    sd_unet_config c = *cfg;
    c.is_vae_decoder = 2;
    c.is_controlnet = 0;
    auto h = std::make_unique<sd_unet>();
    h->impl = std::make_unique<UNet>(c, w->store, device);
    *out = h.release();
  });
}
int sd_vae_encode(sd_unet* vae, const void* x, sd_dtype x_dtype, float* moments, int flags) {
  return guarded([&] {
    SD_REQUIRE(vae && x && moments, kInvalidArgument, "NULL argument");
    vae->impl->vae_encode(x, x_dtype == SD_F32, moments, flags);
  });
}
int sd_vae_decode(sd_unet* vae, const void* z, sd_dtype z_dtype, float* image, int flags) {
  return guarded([&] {
    SD_REQUIRE(vae && z && image, kInvalidArgument, "NULL argument");
    vae->impl->vae_decode(z, z_dtype == SD_F32, image, flags);
  });
}

// ------------------------------- operator-level entry points -------------------------------

int sd_op_attention(int impl, const void* q, const void* k, const void* v, void* out, int B, int heads, int d,
                    int Sq, int Sk, int variant, int iters, float* ms) {
  return guarded([&] {
    SD_REQUIRE(q && k && v && out, kInvalidArgument, "NULL argument");
    SD_REQUIRE(impl >= 0 && impl <= 2, kInvalidArgument, "unknown attention implementation %d", impl);
    SD_REQUIRE(B > 0 && heads > 0 && d > 0 && Sq > 0 && Sk > 0, kInvalidArgument, "empty attention problem");
    Scratch sc;
    const int C = heads * d;
    const int ldv = (Sk + 7) / 8 * 8;
    // the software-pipelined d = 64 kernel reads V^T with the middle 4-key blocks of every 16 keys swapped (AttnDesc::vt_perm)
    const bool perm = variant != 1 && attention8_shape_ok(d, Sq, Sk);
    auto vpos = [perm](int s) { const int o = s & 15; return perm && o >= 4 && o < 12 ? s + (o < 8 ? 4 : -4) : s; };
    const half_t* qh = reinterpret_cast<const half_t*>(q);
    const half_t* kh = reinterpret_cast<const half_t*>(k);
    const half_t* vh = reinterpret_cast<const half_t*>(v);
    // BC1S (B, C, 1, S) -> token-major [B][S][C]; V stays channel-major, rows zero-padded to ldv
    std::vector<half_t> qt((size_t)B * Sq * C), kt((size_t)B * Sk * C), vt((size_t)B * C * ldv, (half_t)0);
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < C; ++c) {
        for (int s = 0; s < Sq; ++s) qt[((size_t)b * Sq + s) * C + c] = qh[((size_t)b * C + c) * Sq + s];
        for (int s = 0; s < Sk; ++s) {
          kt[((size_t)b * Sk + s) * C + c] = kh[((size_t)b * C + c) * Sk + s];
          vt[((size_t)b * C + c) * ldv + vpos(s)] = vh[((size_t)b * C + c) * Sk + s];
        }
      }
    AttnDesc a;
    a.q = sc.dev<half_t>(qt.size(), qt.data());
    a.k = sc.dev<half_t>(kt.size(), kt.data());
    a.vt = sc.dev<half_t>(vt.size(), vt.data());
    half_t* o = sc.dev<half_t>((size_t)B * Sq * C);
    a.out = o;
    a.B = B; a.heads = heads; a.d = d; a.Sq = Sq; a.Sk = Sk;
    a.ldq = C; a.ldk = C; a.ldv = ldv; a.ldo = C;
    a.impl = impl;
    a.variant = variant;
    a.vt_perm = perm ? 1 : 0;
    if (variant == 2) {   // the caller multiplied d^-0.5 * log2(e) into q before rounding it to fp16 (what the UNet's q|k|v GEMM does)
      SD_REQUIRE(perm, kInvalidArgument, "attention variant 2 (pre-scaled q) needs attention8's shape (d %d Sq %d Sk %d)", d, Sq, Sk);
      a.q_prescaled = 1;
    }
    if (variant >= 100) {   // attention8's balanced form, variant - 100 units per workgroup (0: the launch's own split)
      SD_REQUIRE(perm, kInvalidArgument, "attention variant %d (balanced form) needs attention8's shape (d %d Sq %d Sk %d)", variant, d, Sq, Sk);
      a.variant = 0;
      a.sk_force = 1;
      a.sk_upw = variant - 100;
    }
    {
      size_t pb = 0;
      int nc = 0;
      if (perm && attention8_sk_scratch(a, &pb, &nc)) {
        a.sk_part = reinterpret_cast<float*>(sc.dev<char>(pb));
        a.sk_part_bytes = pb;
        a.sk_cnt = sc.dev<unsigned>(nc);
        a.sk_cnt_n = nc;
        SD_HIP(hipMemset(a.sk_cnt, 0, (size_t)nc * sizeof(unsigned)));
      } else {
        SD_REQUIRE(variant < 100, kInvalidArgument, "attention variant %d: the balanced form cannot run this shape", variant);
      }
    }
    sc.timed(iters, ms, [&] { launch_attention(a, sc.stream); });
    std::vector<half_t> ot((size_t)B * Sq * C);
    SD_HIP(hipMemcpy(ot.data(), o, ot.size() * 2, hipMemcpyDeviceToHost));
    half_t* oh = reinterpret_cast<half_t*>(out);
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < C; ++c)
        for (int s = 0; s < Sq; ++s) oh[((size_t)b * C + c) * Sq + s] = ot[((size_t)b * Sq + s) * C + c];
  });
}

int sd_op_layernorm(const void* x, const float* weight, const float* bias, void* out, int B, int C, int S, float eps,
                    int iters, float* ms) {
  return guarded([&] {
    SD_REQUIRE(x && weight && bias && out, kInvalidArgument, "NULL argument");
    Scratch sc;
    // BC1S == NCHW with H=1, W=S
    std::vector<half_t> xt = nchw_to_nhwc(reinterpret_cast<const half_t*>(x), B, C, 1, S);
    half_t* dx = sc.dev<half_t>(xt.size(), xt.data());
    half_t* dy = sc.dev<half_t>(xt.size());
    float* dw = sc.dev<float>(C, weight);
    float* db = sc.dev<float>(C, bias);
    sc.timed(iters, ms, [&] { launch_layernorm(dx, dw, db, dy, B * S, C, eps, sc.stream); });
    SD_HIP(hipMemcpy(xt.data(), dy, xt.size() * 2, hipMemcpyDeviceToHost));
    nhwc_to_nchw(xt.data(), reinterpret_cast<half_t*>(out), B, C, 1, S);
  });
}

int sd_op_groupnorm(const void* x, const float* weight, const float* bias, void* out, int B, int C, int H, int W,
                    int groups, float eps, int silu, int iters, float* ms) {
  return guarded([&] {
    SD_REQUIRE(x && weight && bias && out, kInvalidArgument, "NULL argument");
    Scratch sc;
    std::vector<half_t> xt = nchw_to_nhwc(reinterpret_cast<const half_t*>(x), B, C, H, W);
    half_t* dx = sc.dev<half_t>(xt.size(), xt.data());
    half_t* dy = sc.dev<half_t>(xt.size());
    float* dw = sc.dev<float>(C, weight);
    float* db = sc.dev<float>(C, bias);
    float* partial = sc.dev<float>(groupnorm_scratch_floats(B, H * W, groups));
    sc.timed(iters, ms, [&] {
      launch_groupnorm(dx, C, nullptr, 0, partial, dw, db, dy, B, H * W, groups, eps, silu, sc.stream);
    });
    SD_HIP(hipMemcpy(xt.data(), dy, xt.size() * 2, hipMemcpyDeviceToHost));
    nhwc_to_nchw(xt.data(), reinterpret_cast<half_t*>(out), B, C, H, W);
  });
}

int sd_op_groupnorm_shortcut(const void* x0, const void* x1, const float* gn_weight, const float* gn_bias, const void* w, const float* bias,
                             void* out_gn, void* out_sc, int B, int C0, int C1, int H, int W, int N, int groups, float eps, int silu,
                             int side, int iters, float* ms) {
  return guarded([&] {
    SD_REQUIRE(x0 && gn_weight && gn_bias && w && out_gn && out_sc, kInvalidArgument, "NULL argument");
    if (!x1) C1 = 0;
    const int C = C0 + C1;
    Scratch sc;
    std::vector<half_t> x0t = nchw_to_nhwc(reinterpret_cast<const half_t*>(x0), B, C0, H, W);
    half_t* d0 = sc.dev<half_t>(x0t.size(), x0t.data());
    half_t* d1 = nullptr;
    if (x1) {
      std::vector<half_t> x1t = nchw_to_nhwc(reinterpret_cast<const half_t*>(x1), B, C1, H, W);
      d1 = sc.dev<half_t>(x1t.size(), x1t.data());
    }
    const size_t gn_n = (size_t)B * H * W * C, sc_n = (size_t)B * H * W * N;
    half_t* dy = sc.dev<half_t>(gn_n);
    half_t* ds = sc.dev<half_t>(sc_n);
    float* dgw = sc.dev<float>(C, gn_weight);
    float* dgb = sc.dev<float>(C, gn_bias);
    float* partial = sc.dev<float>(groupnorm_scratch_floats(B, H * W, groups));
    ConvDesc d;   // conv_shortcut: a 1x1 conv over the channel concat (x0 | x1), weights [N][C] as they are
    d.x0 = d0;
    d.C0 = C0;
    d.x1 = d1;
    d.C1 = C1;
    d.w = sc.dev<half_t>((size_t)N * C, reinterpret_cast<const half_t*>(w));
    d.bias = bias ? sc.dev<float>(N, bias) : nullptr;
    d.out = ds;
    d.B = B; d.Hi = H; d.Wi = W; d.Ho = H; d.Wo = W;
    d.N = N;
    SD_REQUIRE(conv_fast_path_ok(d), kInvalidArgument, "groupnorm_shortcut: C0=%d C1=%d N=%d not MFMA-tileable", C0, C1, N);
    SD_REQUIRE(!side || gn_side_gemm_ok(d), kUnsupported, "groupnorm_shortcut: the GEMM cannot ride in the GroupNorm launch (M=%d)", B * H * W);
    ConvWorkspace ws;
    ws.partial_bytes = conv_workspace_bytes(d);
    if (ws.partial_bytes) ws.partial = reinterpret_cast<float*>(sc.dev<char>(ws.partial_bytes));
    sc.timed(iters, ms, [&] {
      if (side) {
        launch_groupnorm(d0, C0, d1, C1, partial, dgw, dgb, dy, B, H * W, groups, eps, silu, sc.stream, 0, &d);
      } else {
        launch_groupnorm(d0, C0, d1, C1, partial, dgw, dgb, dy, B, H * W, groups, eps, silu, sc.stream);
        launch_conv(d, ws, sc.stream);
      }
    });
    std::vector<half_t> ot(gn_n);
    SD_HIP(hipMemcpy(ot.data(), dy, gn_n * 2, hipMemcpyDeviceToHost));
    nhwc_to_nchw(ot.data(), reinterpret_cast<half_t*>(out_gn), B, C, H, W);
    ot.resize(sc_n);
    SD_HIP(hipMemcpy(ot.data(), ds, sc_n * 2, hipMemcpyDeviceToHost));
    nhwc_to_nchw(ot.data(), reinterpret_cast<half_t*>(out_sc), B, N, H, W);
  });
}

int sd_op_conv2d(const void* x, const void* w, const float* bias, const void* res, void* out, int B, int Cin, int H,
                 int W, int Cout, int ksize, int stride, int upsample, int tile, int splitk, int force_generic,
                 int iters, float* ms) {
  return guarded([&] {
    SD_REQUIRE(x && w && out, kInvalidArgument, "NULL argument");
    SD_REQUIRE((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2) && (upsample == 0 || upsample == 1),
               kInvalidArgument, "conv2d: ksize %d stride %d upsample %d not on the path", ksize, stride, upsample);
    Scratch sc;
    const int up = upsample ? 2 : 1, pad = ksize / 2;
    const int Ho = (H * up + 2 * pad - ksize) / stride + 1, Wo = (W * up + 2 * pad - ksize) / stride + 1;
    std::vector<half_t> xt = nchw_to_nhwc(reinterpret_cast<const half_t*>(x), B, Cin, H, W);
    const half_t* wh = reinterpret_cast<const half_t*>(w);
    const int kk = ksize * ksize;
    std::vector<half_t> wt((size_t)Cout * Cin * kk);
    for (int o = 0; o < Cout; ++o)
      for (int c = 0; c < Cin; ++c)
        for (int t = 0; t < kk; ++t) wt[((size_t)o * kk + t) * Cin + c] = wh[((size_t)o * Cin + c) * kk + t];
    ConvDesc d;
    d.x0 = sc.dev<half_t>(xt.size(), xt.data());
    d.C0 = Cin;
    d.w = sc.dev<half_t>(wt.size(), wt.data());
    d.bias = bias ? sc.dev<float>(Cout, bias) : nullptr;
    std::vector<half_t> rt;
    if (res) {
      rt = nchw_to_nhwc(reinterpret_cast<const half_t*>(res), B, Cout, Ho, Wo);
      d.res = sc.dev<half_t>(rt.size(), rt.data());
    }
    const size_t on = (size_t)B * Ho * Wo * Cout;
    half_t* dout = sc.dev<half_t>(on);
    d.out = dout;
    d.B = B; d.Hi = H; d.Wi = W; d.Ho = Ho; d.Wo = Wo;
    d.ksize = ksize; d.stride = stride; d.up = up; d.N = Cout;
    d.tile = tile % 10;          // tile / 10 selects the staging variant of the same tile (A/B testing)
    d.staging = tile / 10;
    if (tile >= 110 && tile <= 116) {   // 110 / 111 / 112 / 113: plan tile 11 (bvgemm.hip) by grid size / variants 1-3 (launch_bvgemm)
      d.tile = 11;
      d.staging = tile - 110;
    }
    d.splitk = splitk;
    d.debug = force_generic >= 2 ? force_generic - 1 : 0;   // 2: loads only, 3: compute only (ablation)
    if (d.debug & 4) d.prof = sc.dev<long long>(8);
    const bool fast = force_generic != 1 && conv_fast_path_ok(d);
    ConvWorkspace ws;
    if (fast && d.tile == 9) {   // plan tile 9: the weight-streaming kernel reads the fragment-major copy of the weights
      SD_REQUIRE(wstream_shape_ok(d), kInvalidArgument, "conv2d: shape not eligible for plan tile 9 (wstream.hip)");
      half_t* wtd = sc.dev<half_t>(wstream_tiled_halves(Cout, Cin, ksize));
      launch_wstream_retile(d.w, wtd, Cout, Cin, ksize, sc.stream);
      d.w_tiled = wtd;
    }
    if (fast && d.tile == 11) {
      SD_REQUIRE(bvgemm_shape_ok(d), kInvalidArgument, "conv2d: shape not eligible for plan tile 11 (bvgemm.hip)");
      half_t* wtd = sc.dev<half_t>(bvgemm_tiled_halves(Cout, Cin));
      launch_bvgemm_retile(d.w, wtd, Cout, Cin, false, sc.stream);
      d.w_bv = wtd;
    }
    if (fast && d.tile != 11) {
      ws.partial_bytes = conv_workspace_bytes(d);
      if (ws.partial_bytes) ws.partial = reinterpret_cast<float*>(sc.dev<char>(ws.partial_bytes));
    }
    // N <= 8 (conv_out of the UNet / the VAE): the small-N kernels the handles use, unless the direct kernel was asked for
    const bool small_n = !fast && force_generic == 0 && Cout <= 8 && Cin % 8 == 0 && ksize == 3 && stride == 1 && !res;
    sc.timed(iters, ms, [&] {
      if (fast)
        launch_conv(d, ws, sc.stream);
      else if (small_n)
        launch_conv_small_n(d, nullptr, sc.stream);
      else
        launch_conv_generic(d, 0, sc.stream);
    });
    if (d.prof) {
      long long t[8];
      SD_HIP(hipMemcpy(t, d.prof, sizeof(t), hipMemcpyDeviceToHost));
      fprintf(stderr, "[sd prof] block0: prologue %lld, k-loop %lld, epilogue %lld shader cycles; total %lld cycles = %lld ticks of the 100 MHz wall clock\n",
              t[1] - t[0], t[2] - t[1], t[3] - t[2], t[3] - t[0], t[4]);
    }
    std::vector<half_t> ot(on);
    SD_HIP(hipMemcpy(ot.data(), dout, on * 2, hipMemcpyDeviceToHost));
    nhwc_to_nchw(ot.data(), reinterpret_cast<half_t*>(out), B, Cout, Ho, Wo);
  });
}

int sd_op_conv2d_groupnorm(const void* x, const void* w, const float* bias, const void* res, const float* gn_weight,
                           const float* gn_bias, void* conv_out, void* out, int B, int Cin, int H, int W, int Cout, int ksize,
                           int groups, float eps, int silu, int tile, int producer_stats, int* entries, int iters, float* ms) {
  return guarded([&] {
    SD_REQUIRE(x && w && gn_weight && gn_bias && out, kInvalidArgument, "NULL argument");
    SD_REQUIRE(ksize == 1 || ksize == 3, kInvalidArgument, "conv2d_groupnorm: ksize %d", ksize);
    Scratch sc;
    std::vector<half_t> xt = nchw_to_nhwc(reinterpret_cast<const half_t*>(x), B, Cin, H, W);
    const half_t* wh = reinterpret_cast<const half_t*>(w);
    const int kk = ksize * ksize;
    std::vector<half_t> wt((size_t)Cout * Cin * kk);
    for (int o = 0; o < Cout; ++o)
      for (int c = 0; c < Cin; ++c)
        for (int t = 0; t < kk; ++t) wt[((size_t)o * kk + t) * Cin + c] = wh[((size_t)o * Cin + c) * kk + t];
    ConvDesc d;
    d.x0 = sc.dev<half_t>(xt.size(), xt.data());
    d.C0 = Cin;
    d.w = sc.dev<half_t>(wt.size(), wt.data());
    d.bias = bias ? sc.dev<float>(Cout, bias) : nullptr;
    std::vector<half_t> rt;
    if (res) {
      rt = nchw_to_nhwc(reinterpret_cast<const half_t*>(res), B, Cout, H, W);
      d.res = sc.dev<half_t>(rt.size(), rt.data());
    }
    const size_t on = (size_t)B * H * W * Cout;
    half_t* dconv = sc.dev<half_t>(on);
    half_t* dy = sc.dev<half_t>(on);
    d.out = dconv;
    d.B = B; d.Hi = H; d.Wi = W; d.Ho = H; d.Wo = W;
    d.ksize = ksize; d.stride = 1; d.up = 1; d.N = Cout;
    d.tile = tile % 10;
    d.staging = tile / 10;
    d.splitk = 1;
    const bool fast = conv_fast_path_ok(d);   // else: conv_in's 4-channel MFMA kernel / the direct kernels
    float* partial = sc.dev<float>(groupnorm_scratch_floats(B, H * W, groups));
    // poison the partial buffer: the fold must only read what the producer wrote
    {
      std::vector<float> poison(groupnorm_scratch_floats(B, H * W, groups), 1.0e30f);
      SD_HIP(hipMemcpy(partial, poison.data(), poison.size() * sizeof(float), hipMemcpyHostToDevice));
    }
    if (producer_stats == 1) {
      d.gn_partial = partial;
      d.gn_groups = groups;
    }
    float* dgw = sc.dev<float>(Cout, gn_weight);
    float* dgb = sc.dev<float>(Cout, gn_bias);
    ConvWorkspace ws;
    if (fast && d.tile == 9) {   // the weight-streaming kernel reads the fragment-major copy of the weights (wstream.hip)
      SD_REQUIRE(wstream_shape_ok(d), kInvalidArgument, "conv2d_groupnorm: shape not eligible for plan tile 9");
      half_t* wtd = sc.dev<half_t>(wstream_tiled_halves(Cout, Cin, ksize));
      launch_wstream_retile(d.w, wtd, Cout, Cin, ksize, sc.stream);
      d.w_tiled = wtd;
    }
    if (producer_stats == 2) {   // the GroupNorm as a twin of the conv's slab combine: no GroupNorm launch
      SD_REQUIRE(fast, kInvalidArgument, "conv2d_groupnorm: GroupNorm twins need the MFMA path");
      d.n_twins = 1;
      d.twin[0].y = dy;
      d.twin[0].ld = Cout;
      d.twin[0].c_off = 0;
      d.twin[0].cpg = Cout / groups;
      d.twin[0].gamma = dgw;
      d.twin[0].beta = dgb;
      d.twin[0].eps = eps;
      d.twin[0].silu = silu;
      SD_REQUIRE(Cout % groups == 0 && reduce_twin_ok(H * W, Cout, 1, d.twin), kInvalidArgument,
                 "conv2d_groupnorm: shape not eligible for a GroupNorm twin (HW=%d C=%d groups=%d)", H * W, Cout, groups);
    }
    if (fast) {
      ws.partial_bytes = conv_workspace_bytes(d);
      if (ws.partial_bytes) ws.partial = reinterpret_cast<float*>(sc.dev<char>(ws.partial_bytes));
    }
    int n_entries = 0;
    sc.timed(iters, ms, [&] {
      n_entries = fast ? launch_conv(d, ws, sc.stream) : launch_conv_generic(d, 0, sc.stream);
      if (producer_stats != 2)
        launch_groupnorm(dconv, Cout, nullptr, 0, partial, dgw, dgb, dy, B, H * W, groups, eps, silu, sc.stream, n_entries);
    });
    if (entries) *entries = n_entries;
    std::vector<half_t> ot(on);
    if (conv_out) {
      SD_HIP(hipMemcpy(ot.data(), dconv, on * 2, hipMemcpyDeviceToHost));
      nhwc_to_nchw(ot.data(), reinterpret_cast<half_t*>(conv_out), B, Cout, H, W);
    }
    SD_HIP(hipMemcpy(ot.data(), dy, on * 2, hipMemcpyDeviceToHost));
    nhwc_to_nchw(ot.data(), reinterpret_cast<half_t*>(out), B, Cout, H, W);
  });
}

int sd_op_conv2d_groupnorm_proj(const void* x, const void* w, const float* bias, const void* res, const float* gn_weight,
                                const float* gn_bias, const void* proj_w, const float* proj_bias, void* conv_out, void* out, int B,
                                int Cin, int H, int W, int Cout, int ksize, int Nproj, int groups, float eps, int fold, int tile,
                                int* entries, int iters, float* ms) {
  return guarded([&] {
    SD_REQUIRE(x && w && gn_weight && gn_bias && proj_w && out, kInvalidArgument, "NULL argument");
    SD_REQUIRE(ksize == 1 || ksize == 3, kInvalidArgument, "conv2d_groupnorm_proj: ksize %d", ksize);
    Scratch sc;
    std::vector<half_t> xt = nchw_to_nhwc(reinterpret_cast<const half_t*>(x), B, Cin, H, W);
    const half_t* wh = reinterpret_cast<const half_t*>(w);
    const int kk = ksize * ksize;
    std::vector<half_t> wt((size_t)Cout * Cin * kk);
    for (int o = 0; o < Cout; ++o)
      for (int c = 0; c < Cin; ++c)
        for (int t = 0; t < kk; ++t) wt[((size_t)o * kk + t) * Cin + c] = wh[((size_t)o * Cin + c) * kk + t];
    ConvDesc d;
    d.x0 = sc.dev<half_t>(xt.size(), xt.data());
    d.C0 = Cin;
    d.w = sc.dev<half_t>(wt.size(), wt.data());
    d.bias = bias ? sc.dev<float>(Cout, bias) : nullptr;
    std::vector<half_t> rt;
    if (res) {
      rt = nchw_to_nhwc(reinterpret_cast<const half_t*>(res), B, Cout, H, W);
      d.res = sc.dev<half_t>(rt.size(), rt.data());
    }
    const size_t on = (size_t)B * H * W * Cout, pn = (size_t)B * H * W * Nproj;
    half_t* dconv = sc.dev<half_t>(on);
    half_t* dnorm = sc.dev<half_t>(on);
    half_t* dy = sc.dev<half_t>(pn);
    d.out = dconv;
    d.B = B; d.Hi = H; d.Wi = W; d.Ho = H; d.Wo = W;
    d.ksize = ksize; d.stride = 1; d.up = 1; d.N = Cout;
    d.tile = tile % 10;
    d.staging = tile / 10;
    d.splitk = 1;
    SD_REQUIRE(conv_fast_path_ok(d), kInvalidArgument, "conv2d_groupnorm_proj: the producer must run on the MFMA path");
    const size_t pf = groupnorm_scratch_floats(B, H * W, groups);
    float* partial = sc.dev<float>(pf);
    {   // poison: the fold must only read what the producer wrote
      std::vector<float> poison(pf, 1.0e30f);
      SD_HIP(hipMemcpy(partial, poison.data(), pf * sizeof(float), hipMemcpyHostToDevice));
    }
    d.gn_partial = partial;
    d.gn_groups = groups;
    float* dgw = sc.dev<float>(Cout, gn_weight);
    float* dgb = sc.dev<float>(Cout, gn_bias);
    ConvDesc pd;   // the 1x1 projection over the conv's output
    pd.C0 = Cout;
    pd.w = sc.dev<half_t>((size_t)Nproj * Cout, reinterpret_cast<const half_t*>(proj_w));
    pd.bias = proj_bias ? sc.dev<float>(Nproj, proj_bias) : nullptr;
    pd.out = dy;
    pd.B = B; pd.Hi = H; pd.Wi = W; pd.Ho = H; pd.Wo = W;
    pd.N = Nproj;
    SD_REQUIRE(conv_fast_path_ok(pd), kInvalidArgument, "conv2d_groupnorm_proj: the projection must run on the MFMA path");
    ConvWorkspace ws;
    ws.partial_bytes = std::max(conv_workspace_bytes(d), conv_workspace_bytes(pd));
    if (ws.partial_bytes) ws.partial = reinterpret_cast<float*>(sc.dev<char>(ws.partial_bytes));
    int n_entries = 0;
    sc.timed(iters, ms, [&] {
      n_entries = launch_conv(d, ws, sc.stream);
      ConvDesc pp = pd;
      if (fold && n_entries >= 1 && n_entries <= 128) {
        pp.x0 = dconv;
        pp.gnf_partial = partial;
        pp.gnf_gamma = dgw;
        pp.gnf_beta = dgb;
        pp.gnf_eps = eps;
        pp.gnf_groups = groups;
        pp.gnf_entries = n_entries;
      } else {
        launch_groupnorm(dconv, Cout, nullptr, 0, partial, dgw, dgb, dnorm, B, H * W, groups, eps, 0, sc.stream, n_entries);
        pp.x0 = dnorm;
      }
      launch_conv(pp, ws, sc.stream);
    });
    if (entries) *entries = (fold && n_entries >= 1 && n_entries <= 128) ? n_entries : 0;
    if (conv_out) {
      std::vector<half_t> ot(on);
      SD_HIP(hipMemcpy(ot.data(), dconv, on * 2, hipMemcpyDeviceToHost));
      nhwc_to_nchw(ot.data(), reinterpret_cast<half_t*>(conv_out), B, Cout, H, W);
    }
    std::vector<half_t> ot(pn);
    SD_HIP(hipMemcpy(ot.data(), dy, pn * 2, hipMemcpyDeviceToHost));
    nhwc_to_nchw(ot.data(), reinterpret_cast<half_t*>(out), B, Nproj, H, W);
  });
}

int sd_op_conv2d_groupnorm_conv3x3(const void* x, const void* w, const float* bias, const void* res, const float* gn_weight,
                                   const float* gn_bias, const void* w2, const float* bias2, const void* res2, void* conv_out, void* out,
                                   int B, int Cin, int H, int W, int Cout, int ksize, int N2, int groups, float eps, int silu, int fold,
                                   int tile, int staging2, int* entries, int iters, float* ms) {
  return guarded([&] {
    SD_REQUIRE(x && w && gn_weight && gn_bias && w2 && out, kInvalidArgument, "NULL argument");
    SD_REQUIRE(ksize == 1 || ksize == 3, kInvalidArgument, "conv2d_groupnorm_conv3x3: ksize %d", ksize);
    Scratch sc;
    std::vector<half_t> xt = nchw_to_nhwc(reinterpret_cast<const half_t*>(x), B, Cin, H, W);
    auto retile = [](const half_t* wh, int co, int ci, int kk) {   // [Cout][Cin][ky][kx] -> [Cout][ky][kx][Cin]
      std::vector<half_t> wt((size_t)co * ci * kk);
      for (int o = 0; o < co; ++o)
        for (int c = 0; c < ci; ++c)
          for (int t = 0; t < kk; ++t) wt[((size_t)o * kk + t) * ci + c] = wh[((size_t)o * ci + c) * kk + t];
      return wt;
    };
    std::vector<half_t> wt = retile(reinterpret_cast<const half_t*>(w), Cout, Cin, ksize * ksize);
    std::vector<half_t> wt2 = retile(reinterpret_cast<const half_t*>(w2), N2, Cout, 9);
    ConvDesc d;
    d.x0 = sc.dev<half_t>(xt.size(), xt.data());
    d.C0 = Cin;
    d.w = sc.dev<half_t>(wt.size(), wt.data());
    d.bias = bias ? sc.dev<float>(Cout, bias) : nullptr;
    std::vector<half_t> rt, rt2;
    if (res) {
      rt = nchw_to_nhwc(reinterpret_cast<const half_t*>(res), B, Cout, H, W);
      d.res = sc.dev<half_t>(rt.size(), rt.data());
    }
    const size_t on = (size_t)B * H * W * Cout, pn = (size_t)B * H * W * N2;
    half_t* dconv = sc.dev<half_t>(on);
    half_t* dnorm = sc.dev<half_t>(on);
    half_t* dy = sc.dev<half_t>(pn);
    d.out = dconv;
    d.B = B; d.Hi = H; d.Wi = W; d.Ho = H; d.Wo = W;
    d.ksize = ksize; d.stride = 1; d.up = 1; d.N = Cout;
    d.tile = tile % 10;
    d.staging = tile / 10;
    d.splitk = 1;
    SD_REQUIRE(conv_fast_path_ok(d), kInvalidArgument, "conv2d_groupnorm_conv3x3: the producer must run on the MFMA path");
    const size_t pf = groupnorm_scratch_floats(B, H * W, groups);
    float* partial = sc.dev<float>(pf);
    {   // poison: the loader's fold must only use what the producer wrote
      std::vector<float> poison(pf, 1.0e30f);
      SD_HIP(hipMemcpy(partial, poison.data(), pf * sizeof(float), hipMemcpyHostToDevice));
    }
    d.gn_partial = partial;
    d.gn_groups = groups;
    float* dgw = sc.dev<float>(Cout, gn_weight);
    float* dgb = sc.dev<float>(Cout, gn_bias);
    ConvDesc cd;   // the 3x3 conv over the normalised tensor
    cd.C0 = Cout;
    cd.w = sc.dev<half_t>(wt2.size(), wt2.data());
    cd.bias = bias2 ? sc.dev<float>(N2, bias2) : nullptr;
    if (res2) {
      rt2 = nchw_to_nhwc(reinterpret_cast<const half_t*>(res2), B, N2, H, W);
      cd.res = sc.dev<half_t>(rt2.size(), rt2.data());
    }
    cd.out = dy;
    cd.B = B; cd.Hi = H; cd.Wi = W; cd.Ho = H; cd.Wo = W;
    cd.ksize = 3; cd.stride = 1; cd.up = 1; cd.N = N2;
    cd.staging = staging2;
    cd.gnf_groups = groups;
    SD_REQUIRE(conv_fast_path_ok(cd), kInvalidArgument, "conv2d_groupnorm_conv3x3: the second conv must run on the MFMA path");
    const bool can_fold = fold && silu && conv_gn_loader_ok(cd);   // (the loader always applies SiLU: every such GroupNorm of the graph has one)
    SD_REQUIRE(!fold || can_fold, kUnsupported, "conv2d_groupnorm_conv3x3: the halo loader cannot normalise C=%d groups=%d @%dx%d", Cout, groups, H, W);
    cd.gnf_groups = 0;
    ConvWorkspace ws;
    ws.partial_bytes = std::max(conv_workspace_bytes(d), conv_workspace_bytes(cd));
    if (ws.partial_bytes) ws.partial = reinterpret_cast<float*>(sc.dev<char>(ws.partial_bytes));
    int n_entries = 0;
    sc.timed(iters, ms, [&] {
      n_entries = launch_conv(d, ws, sc.stream);
      ConvDesc cc = cd;
      if (fold && n_entries >= 1 && n_entries <= 128) {
        cc.x0 = dconv;
        cc.gnf_partial = partial;
        cc.gnf_gamma = dgw;
        cc.gnf_beta = dgb;
        cc.gnf_eps = eps;
        cc.gnf_groups = groups;
        cc.gnf_entries = n_entries;
        cc.gnf_silu = silu ? 1 : 0;
      } else {
        launch_groupnorm(dconv, Cout, nullptr, 0, partial, dgw, dgb, dnorm, B, H * W, groups, eps, silu ? 1 : 0, sc.stream, n_entries);
        cc.x0 = dnorm;
      }
      launch_conv(cc, ws, sc.stream);
    });
    if (entries) *entries = (fold && n_entries >= 1 && n_entries <= 128) ? n_entries : 0;
    if (conv_out) {
      std::vector<half_t> ot(on);
      SD_HIP(hipMemcpy(ot.data(), dconv, on * 2, hipMemcpyDeviceToHost));
      nhwc_to_nchw(ot.data(), reinterpret_cast<half_t*>(conv_out), B, Cout, H, W);
    }
    std::vector<half_t> ot(pn);
    SD_HIP(hipMemcpy(ot.data(), dy, pn * 2, hipMemcpyDeviceToHost));
    nhwc_to_nchw(ot.data(), reinterpret_cast<half_t*>(out), B, N2, H, W);
  });
}

int sd_op_cross_attention_fused(const void* x, const float* ln_weight, const float* ln_bias, const void* wq, const void* k,
                                const void* v, void* out, int B, int heads, int Sq, int Sk, float eps, int nst, int iters,
                                float* ms) {
  return guarded([&] {
    SD_REQUIRE(x && ln_weight && ln_bias && wq && k && v && out, kInvalidArgument, "NULL argument");
    const int C = heads * 64;
    SD_REQUIRE(B > 0 && heads > 0 && xattn_fused_ok(C, heads, Sq, Sk), kUnsupported,
               "cross_attention_fused: heads %d x 64 channels, Sq %d, Sk %d (<= 96)", heads, Sq, Sk);
    Scratch sc;
    const int ldv = (Sk + 7) / 8 * 8;
    const half_t* xh = reinterpret_cast<const half_t*>(x);
    const half_t* kh = reinterpret_cast<const half_t*>(k);
    const half_t* vh = reinterpret_cast<const half_t*>(v);
    const half_t* wh = reinterpret_cast<const half_t*>(wq);
    std::vector<half_t> xt((size_t)B * Sq * C), kt((size_t)B * Sk * C), vt((size_t)B * C * ldv, (half_t)0);
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < C; ++c) {
        for (int s = 0; s < Sq; ++s) xt[((size_t)b * Sq + s) * C + c] = xh[((size_t)b * C + c) * Sq + s];
        for (int s = 0; s < Sk; ++s) {
          kt[((size_t)b * Sk + s) * C + c] = kh[((size_t)b * C + c) * Sk + s];
          vt[((size_t)b * C + c) * ldv + s] = vh[((size_t)b * C + c) * Sk + s];
        }
      }
    // the LayerNorm fold of UNet::fold_layernorm: w' = W * gamma (fp16), colsum of what the MFMA multiplies, bias' = W . beta
    std::vector<half_t> wf((size_t)C * C);
    std::vector<float> colsum(C), biasf(C);
    for (int o = 0; o < C; ++o) {
      double cs = 0.0, bb = 0.0;
      for (int c = 0; c < C; ++c) {
        const float wv = (float)wh[(size_t)o * C + c];
        const half_t hq = (half_t)(wv * ln_weight[c]);
        wf[(size_t)o * C + c] = hq;
        cs += (double)(float)hq;
        bb += (double)wv * (double)ln_bias[c];
      }
      colsum[o] = (float)cs;
      biasf[o] = (float)bb;
    }
    XAttnDesc d;
    d.x = sc.dev<half_t>(xt.size(), xt.data());
    d.wq = sc.dev<half_t>(wf.size(), wf.data());
    d.bias = sc.dev<float>(C, biasf.data());
    d.colsum = sc.dev<float>(C, colsum.data());
    d.k = sc.dev<half_t>(kt.size(), kt.data());
    d.vt = sc.dev<half_t>(vt.size(), vt.data());
    half_t* o = sc.dev<half_t>((size_t)B * Sq * C);
    d.out = o;
    d.M = B * Sq; d.C = C; d.S = Sq; d.L = Sk; d.ldv = ldv; d.heads = heads;
    d.ln_eps = eps;
    d.nst = nst;
    sc.timed(iters, ms, [&] { launch_xattn_fused(d, sc.stream); });
    std::vector<half_t> ot((size_t)B * Sq * C);
    SD_HIP(hipMemcpy(ot.data(), o, ot.size() * 2, hipMemcpyDeviceToHost));
    half_t* oh = reinterpret_cast<half_t*>(out);
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < C; ++c)
        for (int s = 0; s < Sq; ++s) oh[((size_t)b * C + c) * Sq + s] = ot[((size_t)b * Sq + s) * C + c];
  });
}

int sd_op_cross_attention_block(const void* x, const float* ln_weight, const float* ln_bias, const void* wq, const void* k, const void* v,
                                const void* wo, const float* bo, const void* a1, const void* wo1, const float* bo1, void* out, int B, int heads,
                                int Sq, int Sk, float eps, int fused, int iters, float* ms) {
  return guarded([&] {
    SD_REQUIRE(x && ln_weight && ln_bias && wq && k && v && wo && bo && out, kInvalidArgument, "NULL argument");
    const bool pre = a1 != nullptr;
    SD_REQUIRE(!pre || (wo1 && bo1), kInvalidArgument, "cross_attention_block: a1 needs wo1 and bo1");
    const int C = heads * 64;
    SD_REQUIRE(B > 0 && heads > 0 && xattn_fused_ok(C, heads, Sq, Sk), kUnsupported,
               "cross_attention_block: heads %d x 64 channels, Sq %d, Sk %d (<= 96)", heads, Sq, Sk);
    SD_REQUIRE(!fused || (xattn_out_ok(C, heads, Sq, Sk) && (!pre || heads == 5)), kUnsupported,
               "cross_attention_block: the one-launch form takes 5 or 10 heads (with a1: 5) and Sq %% 32 == 0 (heads %d, Sq %d)", heads, Sq);
    Scratch sc;
    const int ldv = (Sk + 7) / 8 * 8;
    const half_t* kh = reinterpret_cast<const half_t*>(k);
    const half_t* vh = reinterpret_cast<const half_t*>(v);
    const half_t* wh = reinterpret_cast<const half_t*>(wq);
    auto to_tokens = [&](const void* p) {   // (B, C, 1, Sq) -> [B * Sq][C]
      const half_t* h = reinterpret_cast<const half_t*>(p);
      std::vector<half_t> t((size_t)B * Sq * C);
      for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
          for (int s = 0; s < Sq; ++s) t[((size_t)b * Sq + s) * C + c] = h[((size_t)b * C + c) * Sq + s];
      return t;
    };
    std::vector<half_t> xt = to_tokens(x), a1t;
    if (pre) a1t = to_tokens(a1);
    std::vector<half_t> kt((size_t)B * Sk * C), vt((size_t)B * C * ldv, (half_t)0);
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < C; ++c)
        for (int s = 0; s < Sk; ++s) {
          kt[((size_t)b * Sk + s) * C + c] = kh[((size_t)b * C + c) * Sk + s];
          vt[((size_t)b * C + c) * ldv + s] = vh[((size_t)b * C + c) * Sk + s];
        }
    // the LayerNorm fold of UNet::fold_layernorm: w' = W * gamma (fp16), colsum of what the MFMA multiplies, bias' = W . beta
    std::vector<half_t> wf((size_t)C * C);
    std::vector<float> colsum(C), biasf(C);
    for (int o = 0; o < C; ++o) {
      double cs = 0.0, bb = 0.0;
      for (int c = 0; c < C; ++c) {
        const float wv = (float)wh[(size_t)o * C + c];
        const half_t hq = (half_t)(wv * ln_weight[c]);
        wf[(size_t)o * C + c] = hq;
        cs += (double)(float)hq;
        bb += (double)wv * (double)ln_bias[c];
      }
      colsum[o] = (float)cs;
      biasf[o] = (float)bb;
    }
    half_t* dx = sc.dev<half_t>(xt.size(), xt.data());
    half_t* da1 = pre ? sc.dev<half_t>(a1t.size(), a1t.data()) : nullptr;
    half_t* dwq = sc.dev<half_t>(wf.size(), wf.data());
    float* dqb = sc.dev<float>(C, biasf.data());
    float* dqc = sc.dev<float>(C, colsum.data());
    half_t* dk = sc.dev<half_t>(kt.size(), kt.data());
    half_t* dvt = sc.dev<half_t>(vt.size(), vt.data());
    half_t* dwo = sc.dev<half_t>((size_t)C * C, reinterpret_cast<const half_t*>(wo));
    float* dbo = sc.dev<float>(C, bo);
    half_t* dwo1 = pre ? sc.dev<half_t>((size_t)C * C, reinterpret_cast<const half_t*>(wo1)) : nullptr;
    float* dbo1 = pre ? sc.dev<float>(C, bo1) : nullptr;
    half_t* dh1 = sc.dev<half_t>((size_t)B * Sq * C);
    half_t* da2 = sc.dev<half_t>((size_t)B * Sq * C);
    half_t* o = sc.dev<half_t>((size_t)B * Sq * C);
    if (fused) {
      half_t* wq_t = sc.dev<half_t>((size_t)C * C);
      half_t* wo_t = sc.dev<half_t>((size_t)C * C);
      launch_xattn_out_retile(dwq, wq_t, C, sc.stream);
      launch_xattn_out_retile(dwo, wo_t, C, sc.stream);
      XAttnOutDesc d;
      d.x = pre ? da1 : dx; d.wq_t = wq_t; d.q_bias = dqb; d.q_colsum = dqc; d.k = dk; d.vt = dvt; d.wo_t = wo_t; d.o_bias = dbo; d.out = o;
      d.M = B * Sq; d.C = C; d.S = Sq; d.L = Sk; d.ldv = ldv; d.heads = heads; d.ln_eps = eps;
      if (pre) {
        half_t* wo1_t = sc.dev<half_t>((size_t)C * C);
        launch_xattn_out_retile(dwo1, wo1_t, C, sc.stream);
        d.h0 = dx; d.wo1_t = wo1_t; d.o1_bias = dbo1;
      }
      sc.timed(iters, ms, [&] { launch_xattn_out(d, sc.stream); });
    } else {
      auto gemm_res = [&](const half_t* in, const half_t* w, const float* bias, const half_t* res, half_t* dst) {   // 1x1 GEMM + residual
        ConvDesc cd;
        cd.x0 = in; cd.C0 = C; cd.w = w; cd.bias = bias; cd.res = res; cd.out = dst;
        cd.B = B; cd.Hi = 1; cd.Wi = Sq; cd.Ho = 1; cd.Wo = Sq; cd.N = C;
        SD_REQUIRE(conv_fast_path_ok(cd), kInvalidArgument, "cross_attention_block: to_out off the MFMA path");
        return cd;
      };
      const half_t* h1 = pre ? dh1 : dx;
      XAttnDesc d;
      d.x = h1; d.wq = dwq; d.bias = dqb; d.colsum = dqc; d.k = dk; d.vt = dvt; d.out = da2;
      d.M = B * Sq; d.C = C; d.S = Sq; d.L = Sk; d.ldv = ldv; d.heads = heads; d.ln_eps = eps;
      ConvDesc c1 = gemm_res(da1 ? da1 : dx, dwo1 ? dwo1 : dwo, dbo1 ? dbo1 : dbo, dx, dh1);   // (only launched with a1)
      ConvDesc c2 = gemm_res(da2, dwo, dbo, h1, o);
      ConvWorkspace ws;
      ws.partial_bytes = std::max(conv_workspace_bytes(c1), conv_workspace_bytes(c2));
      if (ws.partial_bytes) ws.partial = reinterpret_cast<float*>(sc.dev<char>(ws.partial_bytes));
      sc.timed(iters, ms, [&] {
        if (pre) launch_conv(c1, ws, sc.stream);
        launch_xattn_fused(d, sc.stream);
        launch_conv(c2, ws, sc.stream);
      });
    }
    std::vector<half_t> ot((size_t)B * Sq * C);
    SD_HIP(hipMemcpy(ot.data(), o, ot.size() * 2, hipMemcpyDeviceToHost));
    half_t* oh = reinterpret_cast<half_t*>(out);
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < C; ++c)
        for (int s = 0; s < Sq; ++s) oh[((size_t)b * C + c) * Sq + s] = ot[((size_t)b * Sq + s) * C + c];
  });
}

int sd_op_ffn_out_proj(const void* g, const void* w1, const float* b1, const void* res1, const void* w2, const float* b2, const void* res2,
                       void* out, float* gn_sums, int B, int C, int S, int groups, int fused, int iters, float* ms) {
  return guarded([&] {
    SD_REQUIRE(g && w1 && b1 && res1 && w2 && b2 && res2 && out, kInvalidArgument, "NULL argument");
    const int K1 = 4 * C, M = B * S;
    SD_REQUIRE(B > 0 && C % 64 == 0 && S > 0, kInvalidArgument, "ffn_out_proj: B=%d C=%d S=%d", B, C, S);
    SD_REQUIRE(!fused || ffn_proj_ok(C, K1, M, S), kUnsupported, "ffn_out_proj: the one-launch form takes C = 320 and S %% 32 == 0 (C=%d S=%d)", C, S);
    SD_REQUIRE(!gn_sums || (groups >= 1 && C % groups == 0), kInvalidArgument, "ffn_out_proj: groups %d", groups);
    Scratch sc;
    auto to_tokens = [&](const void* p, int ch) {   // (B, ch, 1, S) -> [B * S][ch]
      const half_t* h = reinterpret_cast<const half_t*>(p);
      std::vector<half_t> t((size_t)M * ch);
      for (int b = 0; b < B; ++b)
        for (int c = 0; c < ch; ++c)
          for (int s = 0; s < S; ++s) t[((size_t)b * S + s) * ch + c] = h[((size_t)b * ch + c) * S + s];
      return t;
    };
    std::vector<half_t> gt = to_tokens(g, K1), r1 = to_tokens(res1, C), r2 = to_tokens(res2, C);
    half_t* dg = sc.dev<half_t>(gt.size(), gt.data());
    half_t* dr1 = sc.dev<half_t>(r1.size(), r1.data());
    half_t* dr2 = sc.dev<half_t>(r2.size(), r2.data());
    half_t* dw1 = sc.dev<half_t>((size_t)C * K1, reinterpret_cast<const half_t*>(w1));
    half_t* dw2 = sc.dev<half_t>((size_t)C * C, reinterpret_cast<const half_t*>(w2));
    float* db1 = sc.dev<float>(C, b1);
    float* db2 = sc.dev<float>(C, b2);
    half_t* dh3 = sc.dev<half_t>((size_t)M * C);
    half_t* o = sc.dev<half_t>((size_t)M * C);
    const size_t pf = gn_sums ? groupnorm_scratch_floats(B, S, groups) : 0;
    float* partial = gn_sums ? sc.dev<float>(pf) : nullptr;
    if (partial) {   // poison: only what the producer wrote may be folded
      std::vector<float> poison(pf, 1.0e30f);
      SD_HIP(hipMemcpy(partial, poison.data(), pf * sizeof(float), hipMemcpyHostToDevice));
    }
    int n_entries = 0;
    if (fused) {
      half_t* w1_t = sc.dev<half_t>((size_t)C * K1);
      half_t* w2_t = sc.dev<half_t>((size_t)C * C);
      launch_xattn_out_retile_nk(dw1, w1_t, C, K1, sc.stream);
      launch_xattn_out_retile_nk(dw2, w2_t, C, C, sc.stream);
      FfnProjDesc d;
      d.g = dg; d.w1_t = w1_t; d.b1 = db1; d.res1 = dr1; d.w2_t = w2_t; d.b2 = db2; d.res2 = dr2; d.out = o;
      d.gn_partial = partial; d.gn_groups = groups; d.M = M; d.C = C; d.K1 = K1; d.S = S;
      sc.timed(iters, ms, [&] { n_entries = launch_ffn_proj(d, sc.stream); });
    } else {
      auto gemm_res = [&](const half_t* in, int cin, const half_t* w, const float* bias, const half_t* res, half_t* dst) {
        ConvDesc cd;
        cd.x0 = in; cd.C0 = cin; cd.w = w; cd.bias = bias; cd.res = res; cd.out = dst;
        cd.B = B; cd.Hi = 1; cd.Wi = S; cd.Ho = 1; cd.Wo = S; cd.N = C;
        SD_REQUIRE(conv_fast_path_ok(cd), kInvalidArgument, "ffn_out_proj: off the MFMA path");
        return cd;
      };
      ConvDesc c1 = gemm_res(dg, K1, dw1, db1, dr1, dh3);
      ConvDesc c2 = gemm_res(dh3, C, dw2, db2, dr2, o);
      c2.gn_partial = partial;
      c2.gn_groups = groups;
      ConvWorkspace ws;
      ws.partial_bytes = std::max(conv_workspace_bytes(c1), conv_workspace_bytes(c2));
      if (ws.partial_bytes) ws.partial = reinterpret_cast<float*>(sc.dev<char>(ws.partial_bytes));
      sc.timed(iters, ms, [&] {
        launch_conv(c1, ws, sc.stream);
        n_entries = launch_conv(c2, ws, sc.stream);
      });
    }
    std::vector<half_t> ot((size_t)M * C);
    SD_HIP(hipMemcpy(ot.data(), o, ot.size() * 2, hipMemcpyDeviceToHost));
    half_t* oh = reinterpret_cast<half_t*>(out);
    for (int b = 0; b < B; ++b)
      for (int c = 0; c < C; ++c)
        for (int s = 0; s < S; ++s) oh[((size_t)b * C + c) * S + s] = ot[((size_t)b * S + s) * C + c];
    if (gn_sums) {   // the producer's entries folded on the host: (sum, sumsq) per (sample, group); -1 entries: none written
      std::vector<float> hp(pf);
      SD_HIP(hipMemcpy(hp.data(), partial, pf * sizeof(float), hipMemcpyDeviceToHost));
      for (int b = 0; b < B; ++b)
        for (int gi = 0; gi < groups; ++gi) {
          double s1 = 0.0, s2 = 0.0;
          for (int e = 0; e < n_entries; ++e) {
            s1 += hp[(((size_t)b * groups + gi) * kGnMaxSlabs + e) * 2];
            s2 += hp[(((size_t)b * groups + gi) * kGnMaxSlabs + e) * 2 + 1];
          }
          gn_sums[((size_t)b * groups + gi) * 2] = n_entries ? (float)s1 : NAN;
          gn_sums[((size_t)b * groups + gi) * 2 + 1] = n_entries ? (float)s2 : NAN;
        }
    }
  });
}

int sd_op_geglu(const void* x, const void* w, const float* bias, void* out, int M, int C, int N2, int iters, float* ms) {
  return guarded([&] {
    SD_REQUIRE(x && w && out && N2 % 2 == 0, kInvalidArgument, "bad GEGLU arguments");
    Scratch sc;
    const half_t* wh = reinterpret_cast<const half_t*>(w);
    const int half_n = N2 / 2;
    SD_REQUIRE(half_n % 32 == 0, kUnsupported, "GEGLU needs (N/2) %% 32 == 0");
    std::vector<half_t> wt((size_t)N2 * C);
    std::vector<float> bt(N2, 0.f);
    for (int o = 0; o < N2; ++o) {
      const bool gate = o >= half_n;
      const int j = gate ? o - half_n : o;
      const int dst = (j / 32) * 64 + (gate ? 32 : 0) + (j % 32);
      std::memcpy(&wt[(size_t)dst * C], &wh[(size_t)o * C], (size_t)C * 2);
      if (bias) bt[dst] = bias[o];
    }
    ConvDesc d;
    d.x0 = sc.dev<half_t>((size_t)M * C, reinterpret_cast<const half_t*>(x));
    d.C0 = C;
    d.w = sc.dev<half_t>(wt.size(), wt.data());
    d.bias = bias ? sc.dev<float>(N2, bt.data()) : nullptr;
    half_t* dout = sc.dev<half_t>((size_t)M * half_n);
    d.out = dout;
    d.B = 1; d.Hi = 1; d.Wi = M; d.Ho = 1; d.Wo = M;
    d.N = N2;
    d.out_mode = kOutGeglu;
    const bool fast = conv_fast_path_ok(d);
    ConvWorkspace ws;
    sc.timed(iters, ms, [&] {
      if (fast)
        launch_conv(d, ws, sc.stream);
      else
        launch_conv_generic(d, 0, sc.stream);
    });
    SD_HIP(hipMemcpy(out, dout, (size_t)M * half_n * 2, hipMemcpyDeviceToHost));
  });
}

// GEGLU projection with the LayerNorm in front of it folded in (unet.py:583-591 norm3 -> :609-617 ff.net.0.proj), exactly as the
// UNet builder folds it (UNet::fold_layernorm): x (M, C) f16 un-normalised rows, ln_weight / ln_bias (C) f32 or both NULL (plain
// GEGLU), w (N2, C) f16 [values | gates], bias (N2) f32 or NULL -> out (M, N2 / 2) f16.  kernel: 0 = the plan the library picks,
// 1 = the tiled igemm / gemm_pipe kernels, 2 = the weight-stationary kernel of wsgemm.hip (plan tile 10; refused for other shapes).
int sd_op_geglu_ln(const void* x, const float* ln_weight, const float* ln_bias, const void* w, const float* bias, void* out, int M, int C,
                   int N2, float eps, int kernel, int iters, float* ms) {
  return guarded([&] {
    const int abl = kernel / 10;   // kernel = 2 + 10 * n: ablation build n of the weight-stationary kernel (measurement tools only)
    kernel %= 10;
    SD_REQUIRE(x && w && out && N2 % 64 == 0 && (ln_weight == nullptr) == (ln_bias == nullptr) && kernel >= 0 && kernel <= 9 &&
                   (abl == 0 || kernel == 2), kInvalidArgument, "bad GEGLU arguments");
    Scratch sc;
    const half_t* wh = reinterpret_cast<const half_t*>(w);
    const int half_n = N2 / 2;
    std::vector<half_t> wt((size_t)N2 * C);
    std::vector<float> bt(N2, 0.f), cst(N2, 0.f);
    for (int o = 0; o < N2; ++o) {
      const bool gate = o >= half_n;
      const int j = gate ? o - half_n : o;
      const int dst = (j / 32) * 64 + (gate ? 32 : 0) + (j % 32);
      double cs = 0.0, bb = bias ? (double)bias[o] : 0.0;
      for (int c = 0; c < C; ++c) {
        const float wv = (float)wh[(size_t)o * C + c];
        const half_t h = ln_weight ? (half_t)(wv * ln_weight[c]) : wh[(size_t)o * C + c];
        wt[(size_t)dst * C + c] = h;
        cs += (double)(float)h;
        if (ln_bias) bb += (double)wv * (double)ln_bias[c];
      }
      bt[dst] = (float)bb;
      cst[dst] = (float)cs;
    }
    ConvDesc d;
    d.x0 = sc.dev<half_t>((size_t)M * C, reinterpret_cast<const half_t*>(x));
    d.C0 = C;
    d.w = sc.dev<half_t>(wt.size(), wt.data());
    d.bias = sc.dev<float>(N2, bt.data());
    if (ln_weight) d.ln_colsum = sc.dev<float>(N2, cst.data());
    d.ln_eps = eps;
    half_t* dout = sc.dev<half_t>((size_t)M * half_n);
    d.out = dout;
    d.B = 1; d.Hi = 1; d.Wi = M; d.Ho = 1; d.Wo = M;
    d.N = N2;
    d.out_mode = kOutGeglu;
    SD_REQUIRE(conv_fast_path_ok(d), kUnsupported, "GEGLU shape off the MFMA path (C=%d N2=%d)", C, N2);
    if (kernel != 1 && kernel < 3 && wsgemm_shape_ok(d)) {
      half_t* wtd = sc.dev<half_t>(wsgemm_tiled_halves(N2));
      launch_wsgemm_retile(d.w, wtd, N2, true, sc.stream);
      d.w_ws = wtd;
    }
    if (kernel == 2) d.tile = 10;
    if (kernel >= 3) {   // 3 / 4 / 5 / 6: plan tile 11 (bvgemm.hip) by grid size / variants 1-3
      SD_REQUIRE(bvgemm_shape_ok(d), kInvalidArgument, "GEGLU shape not eligible for plan tile 11 (bvgemm.hip)");
      half_t* wtd = sc.dev<half_t>(bvgemm_tiled_halves(N2, C));
      launch_bvgemm_retile(d.w, wtd, N2, C, true, sc.stream);
      d.w_bv = wtd;
      d.w_ws = nullptr;
      d.tile = 11;
      d.staging = kernel - 3;
    }
    d.debug = abl;
    if (abl == 5) d.prof = sc.dev<long long>(64);
    ConvWorkspace ws;
    sc.timed(iters, ms, [&] { launch_conv(d, ws, sc.stream); });
    if (d.prof) {   // workgroup (0, 0), thread 0: shader-clock stamps of its first pipeline iterations
      long long t[64];
      SD_HIP(hipMemcpy(t, d.prof, sizeof(t), hipMemcpyDeviceToHost));
      for (int i = 0; i < 6; ++i)
        fprintf(stderr, "[sd prof] wsgemm iteration %d: barrier wait %lld, DMA issue + store %lld, statistics %lld, MFMA || epilogue %lld cycles\n",
                i + 1, t[i * 8 + 1] - t[i * 8], t[i * 8 + 2] - t[i * 8 + 1], t[i * 8 + 3] - t[i * 8 + 2], t[i * 8 + 4] - t[i * 8 + 3]);
    }
    SD_HIP(hipMemcpy(out, dout, (size_t)M * half_n * 2, hipMemcpyDeviceToHost));
  });
}

// Fused q|k|v projection of self-attention with norm1 folded in (unet.py:583-586 norm1 -> :74-84 to_q / to_k / to_v as ONE GEMM, as the
// UNet graph runs it): x (B * HW, C) f16 un-normalised tokens, ln_weight / ln_bias (C) f32, w (3C, C) f16 = [Wq | Wk | Wv] (no bias)
// -> out_qk (B * HW, 2C) f16 (the queries multiplied by q_scale on the fp32 accumulator), out_vt (B, C, HW) f16 = V^T, with
// vt_perm in attention8's key order (AttnDesc::vt_perm).  kernel: 0 = the library's plan, 1 = the tiled kernels, 3 = bvgemm.hip
// (its own variant choice), 4 + v = bvgemm variant v + 1.
int sd_op_qkv_ln(const void* x, const float* ln_weight, const float* ln_bias, const void* w, void* out_qk, void* out_vt, int B, int HW, int C,
                 float eps, float q_scale, int vt_perm, int kernel, int iters, float* ms) {
  return guarded([&] {
    SD_REQUIRE(x && ln_weight && ln_bias && w && out_qk && out_vt && B >= 1 && HW >= 1 && C % 64 == 0 && kernel >= 0 && kernel <= 9,
               kInvalidArgument, "bad q|k|v arguments");
    Scratch sc;
    const half_t* wh = reinterpret_cast<const half_t*>(w);
    const int N = 3 * C, M = B * HW;
    std::vector<half_t> wt((size_t)N * C);
    std::vector<float> bt(N, 0.f), cst(N, 0.f);
    for (int o = 0; o < N; ++o) {   // UNet::fold_layernorm
      double cs = 0.0, bb = 0.0;
      for (int c = 0; c < C; ++c) {
        const float wv = (float)wh[(size_t)o * C + c];
        const half_t h = (half_t)(wv * ln_weight[c]);
        wt[(size_t)o * C + c] = h;
        cs += (double)(float)h;
        bb += (double)wv * (double)ln_bias[c];
      }
      bt[o] = (float)bb;
      cst[o] = (float)cs;
    }
    ConvDesc d;
    d.x0 = sc.dev<half_t>((size_t)M * C, reinterpret_cast<const half_t*>(x));
    d.C0 = C;
    d.w = sc.dev<half_t>(wt.size(), wt.data());
    d.bias = sc.dev<float>(N, bt.data());
    d.ln_colsum = sc.dev<float>(N, cst.data());
    d.ln_eps = eps;
    half_t* dqk = sc.dev<half_t>((size_t)M * 2 * C);
    half_t* dvt = sc.dev<half_t>((size_t)B * C * HW);
    d.out = dqk;
    d.out_t = dvt;
    d.n_trans = 2 * C;
    d.ldT = HW;
    d.vt_perm = vt_perm ? 1 : 0;
    d.q_scale = q_scale;
    d.q_cols = C;
    d.B = B; d.Hi = 1; d.Wi = HW; d.Ho = 1; d.Wo = HW;
    d.N = N;
    SD_REQUIRE(conv_fast_path_ok(d), kUnsupported, "q|k|v shape off the MFMA path (C=%d)", C);
    if (kernel == 2 || (kernel == 0 && wsgemm_wanted(d))) {   // the weight-stationary kernel (wsgemm.hip, plan tile 10)
      SD_REQUIRE(wsgemm_shape_ok(d), kInvalidArgument, "q|k|v shape not eligible for plan tile 10 (wsgemm.hip)");
      half_t* wtd = sc.dev<half_t>(wsgemm_tiled_halves(N));
      launch_wsgemm_retile(d.w, wtd, N, false, sc.stream);
      d.w_ws = wtd;
      if (kernel == 2) d.tile = 10;
    } else if (kernel >= 3 || (kernel == 0 && bvgemm_wanted(d))) {
      SD_REQUIRE(bvgemm_shape_ok(d), kInvalidArgument, "q|k|v shape not eligible for plan tile 11 (bvgemm.hip)");
      half_t* wtd = sc.dev<half_t>(bvgemm_tiled_halves(N, C));
      launch_bvgemm_retile(d.w, wtd, N, C, false, sc.stream);
      d.w_bv = wtd;
      if (kernel >= 3) {
        d.tile = 11;
        d.staging = kernel - 3;
      }
    }
    ConvWorkspace ws;
    sc.timed(iters, ms, [&] { launch_conv(d, ws, sc.stream); });
    SD_HIP(hipMemcpy(out_qk, dqk, (size_t)M * 2 * C * 2, hipMemcpyDeviceToHost));
    SD_HIP(hipMemcpy(out_vt, dvt, (size_t)B * C * HW * 2, hipMemcpyDeviceToHost));
  });
}

// The head of a SpatialTransformer (unet.py:553-556 norm -> proj_in, :583-586 norm1 -> :74-84 fused to_q | to_k | to_v) behind a 1x1 conv
// that produces its input x = conv(x_in) and - like the resnet conv in front of it in the UNet - leaves the GroupNorm statistics of x in
// its epilogue.  fused = 1: ONE launch (xattn_out.hip gn_proj_qkv_kernel; 2 / 3: its 64- / 32-token form); 0: GroupNorm launch, proj_in GEMM,
// LayerNorm-folded q|k|v GEMM.
// x_in (B, C, H, W) f16 NCHW; conv_w (C, C); gn_* (C) f32; proj_w (C, C), proj_bias (C); ln_* (C); wqkv (3C, C) -> out_h (B * HW, C),
// out_qk (B * HW, 2C), out_vt (B, C, HW), all f16.  *entries = the producer's partial entries per (sample, group) the fused launch folded.
int sd_op_gn_proj_qkv(const void* x_in, const void* conv_w, const float* gn_weight, const float* gn_bias, const void* proj_w,
                      const float* proj_bias, const float* ln_weight, const float* ln_bias, const void* wqkv, void* out_h, void* out_qk,
                      void* out_vt, int B, int H, int W, int C, int groups, float gn_eps, float ln_eps, float q_scale, int vt_perm, int fused,
                      int* entries, int iters, float* ms) {
  return guarded([&] {
    SD_REQUIRE(x_in && conv_w && gn_weight && gn_bias && proj_w && proj_bias && ln_weight && ln_bias && wqkv && out_h && out_qk && out_vt,
               kInvalidArgument, "NULL argument");
    const int HW = H * W, M = B * HW, N = 3 * C;
    SD_REQUIRE(gn_proj_qkv_ok(C, C / 64, HW, M, HW, groups), kInvalidArgument, "gn_proj_qkv: C=%d HW=%d groups=%d", C, HW, groups);
    Scratch sc;
    std::vector<half_t> xt = nchw_to_nhwc(reinterpret_cast<const half_t*>(x_in), B, C, H, W);
    ConvDesc d;   // the producer
    d.x0 = sc.dev<half_t>(xt.size(), xt.data());
    d.C0 = C;
    d.w = sc.dev<half_t>((size_t)C * C, reinterpret_cast<const half_t*>(conv_w));
    half_t* dx = sc.dev<half_t>((size_t)M * C);
    d.out = dx;
    d.B = B; d.Hi = H; d.Wi = W; d.Ho = H; d.Wo = W;
    d.N = C;
    d.splitk = 1;
    const size_t pf = groupnorm_scratch_floats(B, HW, groups);
    float* partial = sc.dev<float>(pf);
    {   // poison: the fold must only read what the producer wrote
      std::vector<float> poison(pf, 1.0e30f);
      SD_HIP(hipMemcpy(partial, poison.data(), pf * sizeof(float), hipMemcpyHostToDevice));
    }
    d.gn_partial = partial;
    d.gn_groups = groups;
    float* dgw = sc.dev<float>(C, gn_weight);
    float* dgb = sc.dev<float>(C, gn_bias);
    // LayerNorm fold of the fused q|k|v (UNet::fold_layernorm)
    const half_t* wh = reinterpret_cast<const half_t*>(wqkv);
    std::vector<half_t> wf((size_t)N * C);
    std::vector<float> bt(N, 0.f), cst(N, 0.f);
    for (int o = 0; o < N; ++o) {
      double cs = 0.0, bb = 0.0;
      for (int c = 0; c < C; ++c) {
        const float wv = (float)wh[(size_t)o * C + c];
        const half_t h = (half_t)(wv * ln_weight[c]);
        wf[(size_t)o * C + c] = h;
        cs += (double)(float)h;
        bb += (double)wv * (double)ln_bias[c];
      }
      bt[o] = (float)bb;
      cst[o] = (float)cs;
    }
    half_t* dwp = sc.dev<half_t>((size_t)C * C, reinterpret_cast<const half_t*>(proj_w));
    float* dpb = sc.dev<float>(C, proj_bias);
    half_t* dwq = sc.dev<half_t>(wf.size(), wf.data());
    float* dqb = sc.dev<float>(N, bt.data());
    float* dqc = sc.dev<float>(N, cst.data());
    half_t* dnorm = sc.dev<half_t>((size_t)M * C);
    half_t* dh = sc.dev<half_t>((size_t)M * C);
    half_t* dqk = sc.dev<half_t>((size_t)M * 2 * C);
    half_t* dvt = sc.dev<half_t>((size_t)B * C * HW);
    half_t* dwp_t = sc.dev<half_t>((size_t)C * C);
    half_t* dwq_t = sc.dev<half_t>((size_t)N * C);
    launch_xattn_out_retile_nk(dwp, dwp_t, C, C, sc.stream);
    launch_xattn_out_retile_nk(dwq, dwq_t, N, C, sc.stream);
    ConvDesc pd;   // proj_in of the three-launch path
    pd.x0 = dnorm; pd.C0 = C; pd.w = dwp; pd.bias = dpb; pd.out = dh;
    pd.B = B; pd.Hi = H; pd.Wi = W; pd.Ho = H; pd.Wo = W; pd.N = C;
    ConvDesc qd;   // fused q|k|v of the three-launch path
    qd.x0 = dh; qd.C0 = C; qd.w = dwq; qd.bias = dqb; qd.ln_colsum = dqc; qd.ln_eps = ln_eps; qd.out = dqk; qd.out_t = dvt;
    qd.n_trans = 2 * C; qd.ldT = HW; qd.vt_perm = vt_perm ? 1 : 0; qd.q_scale = q_scale; qd.q_cols = C;
    qd.B = B; qd.Hi = H; qd.Wi = W; qd.Ho = H; qd.Wo = W; qd.N = N;
    SD_REQUIRE(conv_fast_path_ok(d) && conv_fast_path_ok(pd) && conv_fast_path_ok(qd), kInvalidArgument, "gn_proj_qkv: off the MFMA path");
    ConvWorkspace ws;
    ws.partial_bytes = std::max(conv_workspace_bytes(d), std::max(conv_workspace_bytes(pd), conv_workspace_bytes(qd)));
    if (ws.partial_bytes) ws.partial = reinterpret_cast<float*>(sc.dev<char>(ws.partial_bytes));
    int n_entries = 0;
    static const bool want_clk = tune_env_set("SD_GQ_CLOCK");   // phase clock of the one-launch kernel, printed to stderr
    const size_t n_clk = (size_t)(M / 32) * 5 * 16;
    long long* dclk = want_clk && fused ? sc.dev<long long>(n_clk) : nullptr;
    sc.timed(iters, ms, [&] {
      n_entries = launch_conv(d, ws, sc.stream);
      const bool have = n_entries >= 1 && n_entries <= 128;
      if (fused) {
        GnProjQkvDesc g;
        g.clk = dclk;
        g.tok = fused == 2 ? 64 : (fused == 3 ? 32 : 0);   // operator tests: the 64- / 32-token form whatever the launch's rule says
        g.x = dx;
        if (have) {
          g.gn_partial = partial; g.gn_gamma = dgw; g.gn_beta = dgb; g.gn_entries = n_entries;
        } else {   // no producer statistics: the GroupNorm launch, then the fused launch on the normalised tensor
          launch_groupnorm(dx, C, nullptr, 0, partial, dgw, dgb, dnorm, B, HW, groups, gn_eps, 0, sc.stream, n_entries);
          g.x = dnorm;
        }
        g.gn_groups = groups; g.gn_eps = gn_eps;
        g.wp_t = dwp_t; g.p_bias = dpb; g.h = dh;
        g.wqkv_t = dwq_t; g.qkv_bias = dqb; g.qkv_colsum = dqc; g.ln_eps = ln_eps;
        g.qk = dqk; g.vt = dvt; g.M = M; g.C = C; g.S = HW; g.ldT = HW; g.vt_perm = vt_perm != 0; g.q_scale = q_scale;
        launch_gn_proj_qkv(g, sc.stream);
      } else {
        launch_groupnorm(dx, C, nullptr, 0, partial, dgw, dgb, dnorm, B, HW, groups, gn_eps, 0, sc.stream, n_entries);
        launch_conv(pd, ws, sc.stream);
        launch_conv(qd, ws, sc.stream);
      }
    });
    if (entries) *entries = (fused && n_entries >= 1 && n_entries <= 128) ? n_entries : 0;
    if (dclk) {
      std::vector<long long> hc(n_clk);
      SD_HIP(hipMemcpy(hc.data(), dclk, n_clk * sizeof(long long), hipMemcpyDeviceToHost));
      static const char* names[13] = {"start -> group statistics folded", "-> constants in LDS", "-> normalised tile in LDS", "-> proj_in MFMAs issued",
                                      "-> h tile in LDS", "-> q MFMAs", "-> q stored", "-> k MFMAs", "-> k stored", "-> v MFMAs", "-> V^T stored", "", ""};
      const size_t nw = n_clk / 16;
      double total = 0;
      for (int ph = 1; ph <= 11; ++ph) {
        double sum = 0, mx = 0;
        for (size_t w = 0; w < nw; ++w) {
          const double dlt = (double)(hc[w * 16 + ph] - hc[w * 16 + ph - 1]);
          sum += dlt;
          mx = std::max(mx, dlt);
        }
        total += sum / nw;
        fprintf(stderr, "gn_proj_qkv phase %2d  mean %8.0f  max %8.0f cycles   %s\n", ph, sum / nw, mx, names[ph - 1]);
      }
      fprintf(stderr, "gn_proj_qkv mean wave %8.0f cycles (%zu waves; the XCDs' counters are not synchronised: no launch-wide span)\n", total, nw);
    }
    SD_HIP(hipMemcpy(out_h, dh, (size_t)M * C * 2, hipMemcpyDeviceToHost));
    SD_HIP(hipMemcpy(out_qk, dqk, (size_t)M * 2 * C * 2, hipMemcpyDeviceToHost));
    SD_HIP(hipMemcpy(out_vt, dvt, (size_t)B * C * HW * 2, hipMemcpyDeviceToHost));
  });
}

int sd_op_timestep_embedding(const float* t, float* out, int n, int dim, int flip_sin_to_cos, float freq_shift) {
  return guarded([&] {
    SD_REQUIRE(t && out && n > 0 && dim > 0 && dim % 2 == 0, kInvalidArgument, "bad arguments");
    SD_REQUIRE(flip_sin_to_cos == 1, kUnsupported, "flip_sin_to_cos=False is not on the path");
    Scratch sc;
    float* dt = sc.dev<float>(n, t);
    float* dout = sc.dev<float>((size_t)n * dim);
    std::vector<float> f = timestep_freq_table(dim, freq_shift);
    float* df = sc.dev<float>(f.size(), f.data());
    launch_timestep_embedding(dt, df, dout, n, dim, sc.stream);
    SD_HIP(hipStreamSynchronize(sc.stream));
    SD_HIP(hipMemcpy(out, dout, (size_t)n * dim * sizeof(float), hipMemcpyDeviceToHost));
  });
}

namespace {
// MT19937 as numpy's legacy RandomState and torch's CPU generator both use it (init_genrand + genrand_int32)
struct Mt19937 {
  uint32_t key[624];
  int pos = 624;
  explicit Mt19937(uint32_t seed) {
    uint32_t s = seed;
    for (uint32_t i = 0; i < 624; ++i) {
      key[i] = s;
      s = 1812433253u * (s ^ (s >> 30)) + i + 1;
    }
  }
  uint32_t next_u32() {
    if (pos == 624) {
      for (int i = 0; i < 624; ++i) {
        const uint32_t y = (key[i] & 0x80000000u) | (key[(i + 1) % 624] & 0x7fffffffu);
        key[i] = key[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
      }
      pos = 0;
    }
    uint32_t y = key[pos++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
  }
};
}  // namespace

// numpy legacy RandomState: MT19937 + 53-bit doubles + Marsaglia polar method
// (NumPyRandomSource.swift:28-102; golden: StableDiffusionTests.swift:52-62)
int sd_numpy_randn(uint32_t seed, double* out, size_t n) {
  return guarded([&] {
    SD_REQUIRE(out || n == 0, kInvalidArgument, "NULL output");
    Mt19937 mt(seed);
    auto next_double = [&]() -> double {
      const uint32_t a = mt.next_u32() >> 5, b = mt.next_u32() >> 6;
      return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
    };
    bool has_cached = false;
    double cached = 0.0;
    for (size_t i = 0; i < n; ++i) {
      if (has_cached) {
        out[i] = cached;
        has_cached = false;
        continue;
      }
      double x1, x2, r2;
      do {
        x1 = 2.0 * next_double() - 1.0;
        x2 = 2.0 * next_double() - 1.0;
        r2 = x1 * x1 + x2 * x2;
      } while (r2 >= 1.0 || r2 == 0.0);
      const double f = std::sqrt(-2.0 * std::log(r2) / r2);
      cached = f * x1;
      has_cached = true;
      out[i] = f * x2;
    }
  });
}

// torch.manual_seed(seed); torch.randn(n) on the CPU (TorchRandomSource.swift:116-150): 24-bit uniforms, Box-Muller
// over blocks of 16, ragged tail from 53-bit doubles over the last 16; n < 16 scalar Box-Muller with the sine cached
int sd_torch_randn(uint32_t seed, double* out, size_t n) {
  return guarded([&] {
    SD_REQUIRE(out || n == 0, kInvalidArgument, "NULL output");
    Mt19937 mt(seed);
    const double two_pi = 6.283185307179586476925286766559;
    auto next_double53 = [&]() -> double {
      const uint64_t hi = mt.next_u32(), lo = mt.next_u32();
      return (double)(((hi << 32) | lo) & 9007199254740991ull) * (1.0 / 9007199254740992.0);
    };
    if (n < 16) {
      bool has_cached = false;
      double cached = 0.0;
      for (size_t i = 0; i < n; ++i) {
        if (has_cached) {
          out[i] = cached;
          has_cached = false;
          continue;
        }
        const double u1 = next_double53();
        const double u2 = 1.0 - next_double53();
        const double radius = std::sqrt(-2.0 * std::log(u2));
        cached = radius * std::sin(two_pi * u1);
        has_cached = true;
        out[i] = radius * std::cos(two_pi * u1);
      }
      return;
    }
    for (size_t i = 0; i < n; ++i) out[i] = (double)(mt.next_u32() & 16777215u) * (1.0 / 16777216.0);
    auto fill16 = [&](size_t i) {
      for (size_t j = 0; j < 8; ++j) {
        const double u1 = 1.0 - out[i + j], u2 = out[i + j + 8];
        const double radius = std::sqrt(-2.0 * std::log(u1)), theta = two_pi * u2;
        out[i + j] = radius * std::cos(theta);
        out[i + j + 8] = radius * std::sin(theta);
      }
    };
    for (size_t i = 0; i + 15 < n; i += 16) fill16(i);
    if (n % 16) {
      // ragged tail: the last 16 values are redrawn.  torch draws them like the rest (24-bit floats); the Swift
      // restatement draws 53-bit doubles here (TorchRandomSource.swift:135-137).  Latent counts on the path are
      // multiples of 16, where the two agree; torch itself is followed for the tail (tests pin it against torch).
      for (size_t i = n - 16; i < n; ++i) out[i] = (double)(mt.next_u32() & 16777215u) * (1.0 / 16777216.0);
      fill16(n - 16);
    }
  });
}

// torch.randn on a CUDA device (NvRandomSource.swift:25-80): Philox4x32-10, counter (offset, 0, i, 0), key = seed,
// Box-Muller on the first two output words of element i
int sd_philox_randn(uint64_t seed, uint32_t offset, double* out, size_t n) {
  return guarded([&] {
    SD_REQUIRE(out || n == 0, kInvalidArgument, "NULL output");
    const double pi = 3.14159265358979323846;
    for (size_t i = 0; i < n; ++i) {
      uint32_t c0 = offset, c1 = 0, c2 = (uint32_t)i, c3 = 0;
      uint32_t k0 = (uint32_t)(seed & 0xffffffffu), k1 = (uint32_t)(seed >> 32);
      for (int r = 0; r < 10; ++r) {
        const uint64_t v1 = (uint64_t)c0 * 0xD2511F53u, v2 = (uint64_t)c2 * 0xCD9E8D57u;
        const uint32_t n0 = (uint32_t)(v2 >> 32) ^ c1 ^ k0, n1 = (uint32_t)v2;
        const uint32_t n2 = (uint32_t)(v1 >> 32) ^ c3 ^ k1, n3 = (uint32_t)v1;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        if (r < 9) {
          k0 += 0x9E3779B9u;
          k1 += 0xBB67AE85u;
        }
      }
      const double u = (double)c0 / 4294967296.0 + (1.0 / 8589934592.0);
      const double v = (double)c1 * (pi / 2147483648.0) + (pi / 4294967296.0);
      out[i] = std::sqrt(-2.0 * std::log(u)) * std::sin(v);
    }
  });
}

int sd_calibrate(int device, float* out9) {
  return guarded([&] {
    SD_REQUIRE(out9, kInvalidArgument, "NULL argument");
    require_device();
    run_calibration(device, out9);
  });
}

int sd_selftest_mfma(void) {
  int rc = 0;
  int st = guarded([&] {
    require_device();
    rc = selftest_mfma();
  });
  return st != kOk ? st : rc;
}

}  // extern "C"
