"""ctypes binding of ``libsdmi355.so`` (include/sd_mi355x.h).

The library is the product; this module only marshals numpy arrays across the C ABI and maps
status codes onto the exception classes the reference raises at the same seam
(python_coreml_stable_diffusion/coreml_model.py:97-116, :176-178).  There is no CPU fallback:
a missing library is an ImportError-class failure with build instructions, and a missing GPU
surfaces as RuntimeError from the first call that needs one.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SD_MI355X_LIB", os.path.join(os.path.dirname(_HERE), "lib", "libsdmi355.so"))

SD_MAX_LEVELS = 6
SD_FLAG_DEVICE_PTRS = 1
ATTENTION_IMPLEMENTATIONS = {"ORIGINAL": 0, "SPLIT_EINSUM": 1, "SPLIT_EINSUM_V2": 2}


class LibraryNotBuilt(ImportError):
    pass


class UNetConfig(C.Structure):
    _fields_ = [
        ("batch", C.c_int32), ("in_channels", C.c_int32), ("out_channels", C.c_int32),
        ("height", C.c_int32), ("width", C.c_int32), ("n_levels", C.c_int32),
        ("block_out_channels", C.c_int32 * SD_MAX_LEVELS),
        ("down_cross_attn", C.c_int32 * SD_MAX_LEVELS),
        ("up_cross_attn", C.c_int32 * SD_MAX_LEVELS),
        ("layers_per_block", C.c_int32),
        ("attention_head_dim", C.c_int32 * SD_MAX_LEVELS),
        ("transformer_layers_per_block", C.c_int32 * SD_MAX_LEVELS),
        ("cross_attention_dim", C.c_int32), ("context_len", C.c_int32),
        ("norm_num_groups", C.c_int32), ("norm_eps", C.c_float),
        ("flip_sin_to_cos", C.c_int32), ("freq_shift", C.c_float),
        ("addition_time_embed_dim", C.c_int32), ("projection_class_embeddings_input_dim", C.c_int32),
        ("num_time_ids", C.c_int32), ("support_controlnet", C.c_int32), ("is_controlnet", C.c_int32),
        ("is_vae_decoder", C.c_int32),
        ("attention_impl", C.c_int32), ("use_graph", C.c_int32), ("compute_fp32", C.c_int32),
    ]


class UNetIO(C.Structure):
    _fields_ = [
        ("sample", C.c_void_p), ("timestep", C.c_void_p), ("encoder_hidden_states", C.c_void_p),
        ("time_ids", C.c_void_p), ("text_embeds", C.c_void_p), ("controlnet_cond", C.c_void_p),
        ("additional_residuals", C.POINTER(C.c_void_p)), ("num_additional_residuals", C.c_int32),
        ("noise_pred", C.c_void_p), ("residual_outputs", C.POINTER(C.c_void_p)), ("flags", C.c_int32),
        ("step_noise", C.c_void_p),
    ]


_lib = None

# every symbol include/sd_mi355x.h declares: (name, restype, argtypes)
_P, _I, _F = C.c_void_p, C.c_int, C.c_float
_FP = C.POINTER(C.c_float)
SYMBOLS = [
    ("sd_last_error", C.c_char_p, []),
    ("sd_version", C.c_char_p, []),
    ("sd_device_count", _I, []),
    ("sd_weights_create", _I, [C.POINTER(_P)]),
    ("sd_weights_add", _I, [_P, C.c_char_p, _P, _I, C.POINTER(C.c_int64), _I]),
    ("sd_weights_load_safetensors", _I, [_P, C.c_char_p, C.c_char_p]),
    ("sd_weights_count", _I, [_P]),
    ("sd_weights_destroy", None, [_P]),
    ("sd_unet_create", _I, [C.POINTER(UNetConfig), _P, _I, C.POINTER(_P)]),
    ("sd_unet_destroy", None, [_P]),
    ("sd_unet_set_attention", _I, [_P, _I]),
    ("sd_unet_num_residuals", _I, [_P]),
    ("sd_unet_device_bytes", C.c_size_t, [_P]),
    ("sd_unet_forward", _I, [_P, C.POINTER(UNetIO)]),
    ("sd_unet_time_forward", _I, [_P, _I, _I, _FP]),
    ("sd_unet_denoise_loop", _I, [_P, C.POINTER(UNetIO), _FP, _I, _I, _FP, _FP, _FP, _I, _F, _FP, _FP]),
    ("sd_tune_set_candidate", _I, [_I, _I, _I]),
    ("sd_tune_set_plan_table", _I, [C.c_char_p, C.c_void_p, C.POINTER(C.c_int)]),
    ("sd_unet_profile", _I, [_P, _I, _I, _FP, C.POINTER(C.c_double), C.c_char_p, _I, C.POINTER(_I)]),
    ("sd_unet_attach_controlnets", _I, [_P, C.POINTER(_P), _I]),
    ("sd_controlnet_set_cond", _I, [_P, _P, _I]),
    ("sd_vae_decoder_create", _I, [C.POINTER(UNetConfig), _P, _I, C.POINTER(_P)]),
    ("sd_vae_decode", _I, [_P, _P, _I, _FP, _I]),
    ("sd_vae_encoder_create", _I, [C.POINTER(UNetConfig), _P, _I, C.POINTER(_P)]),
    ("sd_vae_encode", _I, [_P, _P, _I, _FP, _I]),
    ("sd_text_encoder_create", _I, [_P, _P, _I, C.POINTER(_P)]),
    ("sd_text_encoder_destroy", None, [_P]),
    ("sd_text_encoder_device_bytes", C.c_size_t, [_P]),
    ("sd_text_encoder_encode", _I, [_P, C.POINTER(C.c_int32), _I, _FP, _FP, _FP]),
    ("sd_op_attention", _I, [_I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _FP]),
    ("sd_op_layernorm", _I, [_P, _FP, _FP, _P, _I, _I, _I, _F, _I, _FP]),
    ("sd_op_groupnorm", _I, [_P, _FP, _FP, _P, _I, _I, _I, _I, _I, _F, _I, _I, _FP]),
    ("sd_op_groupnorm_shortcut", _I, [_P, _P, _FP, _FP, _P, _FP, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _I, _FP]),
    ("sd_op_conv2d", _I, [_P, _P, _FP, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _FP]),
    ("sd_op_conv2d_groupnorm", _I, [_P, _P, _FP, _P, _FP, _FP, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _I, C.POINTER(_I), _I, _FP]),
    ("sd_op_conv2d_groupnorm_proj", _I, [_P, _P, _FP, _P, _FP, _FP, _P, _FP, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, C.POINTER(_I), _I, _FP]),
    ("sd_op_conv2d_groupnorm_conv3x3", _I, [_P, _P, _FP, _P, _FP, _FP, _P, _FP, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _F, _I, _I, _I, _I,
                                            C.POINTER(_I), _I, _FP]),
    ("sd_op_cross_attention_fused", _I, [_P, _FP, _FP, _P, _P, _P, _P, _I, _I, _I, _I, _F, _I, _I, _FP]),
    ("sd_op_ffn_out_proj", _I, [_P, _P, _FP, _P, _P, _FP, _P, _P, _FP, _I, _I, _I, _I, _I, _I, _FP]),
    ("sd_op_cross_attention_block", _I, [_P, _FP, _FP, _P, _P, _P, _P, _FP, _P, _P, _FP, _P, _I, _I, _I, _I, _F, _I, _I, _FP]),
    ("sd_op_geglu", _I, [_P, _P, _FP, _P, _I, _I, _I, _I, _FP]),
    ("sd_op_geglu_ln", _I, [_P, _FP, _FP, _P, _FP, _P, _I, _I, _I, C.c_float, _I, _I, _FP]),
    ("sd_op_qkv_ln", _I, [_P, _FP, _FP, _P, _P, _P, _I, _I, _I, C.c_float, C.c_float, _I, _I, _I, _FP]),
    ("sd_op_gn_proj_qkv", _I, [_P, _P, _FP, _FP, _P, _FP, _FP, _FP, _P, _P, _P, _P, _I, _I, _I, _I, _I, C.c_float, C.c_float, C.c_float, _I, _I,
                               C.POINTER(C.c_int), _I, _FP]),
    ("sd_op_timestep_embedding", _I, [_FP, _FP, _I, _I, _I, _F]),
    ("sd_numpy_randn", _I, [C.c_uint32, C.POINTER(C.c_double), C.c_size_t]),
    ("sd_torch_randn", _I, [C.c_uint32, C.POINTER(C.c_double), C.c_size_t]),
    ("sd_philox_randn", _I, [C.c_uint64, C.c_uint32, C.POINTER(C.c_double), C.c_size_t]),
    ("sd_calibrate", _I, [_I, _FP]),
    ("sd_selftest_mfma", _I, []),
]


def lib():
    """Load the shared library (once).  torch is imported first when available so that both bind
    the same HIP runtime (torch ships its own libamdhip64 with the same soname)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryNotBuilt(
            f"{LIB_PATH} is missing - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C ml-stable-diffusion_amd/csrc`. There is no CPU fallback.")
    if not os.environ.get("SD_MI355X_NO_TORCH"):
        try:
            import torch  # noqa: F401  (plumbing only: shares libamdhip64 / RCCL with torch.distributed)
        except Exception:  # pragma: no cover - torch is optional for the library itself
            pass
    handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(handle, name)   # AttributeError here == header/library mismatch
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = handle
    return handle


_EXC = {-1: ValueError, -2: FileNotFoundError, -3: RuntimeError, -4: NotImplementedError, -5: RuntimeError}


def check(status):
    if status != 0:
        msg = lib().sd_last_error().decode("utf-8", "replace")
        raise _EXC.get(status, RuntimeError)(msg)


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def fptr(a):
    return None if a is None else a.ctypes.data_as(_FP)


def f16(a):
    return np.ascontiguousarray(a, dtype=np.float16)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ------------------------------------------------------------------------------------------------
# operator-level wrappers (tests / micro-benchmarks); all take & return the reference layouts
# ------------------------------------------------------------------------------------------------
def attention(impl, q, k, v, heads, dim_head, variant=0, iters=1):
    """attention.py:24-168.  q (B,h*d,1,Sq), k/v (B,h*d,1,Sk) -> (B,h*d,1,Sq) fp16, ms.  variant: sd_mi355x.h (0 default dispatch, 1 never the
    d = 64 pipelined kernel, 2 pre-scaled q, 100 + u that kernel's balanced grid with u units per workgroup)."""
    if impl not in ATTENTION_IMPLEMENTATIONS:
        raise ValueError(f"unknown attention implementation {impl!r}")
    q, k, v = f16(q), f16(k), f16(v)
    B, Cq, _, Sq = q.shape
    Sk = k.shape[3]
    if Cq != heads * dim_head or k.shape[1] != Cq or v.shape != k.shape:
        raise ValueError("attention: inconsistent q/k/v shapes")
    out = np.empty_like(q)
    ms = C.c_float(0)
    check(lib().sd_op_attention(ATTENTION_IMPLEMENTATIONS[impl], ptr(q), ptr(k), ptr(v), ptr(out), B, heads,
                                dim_head, Sq, Sk, variant, iters, C.byref(ms)))
    return out, ms.value


def layernorm(x, weight, bias, eps=1e-5, iters=1):
    x, weight, bias = f16(x), f32(weight), f32(bias)
    B, Cn, _, S = x.shape
    out = np.empty_like(x)
    ms = C.c_float(0)
    check(lib().sd_op_layernorm(ptr(x), fptr(weight), fptr(bias), ptr(out), B, Cn, S, eps, iters, C.byref(ms)))
    return out, ms.value


def groupnorm(x, weight, bias, groups=32, eps=1e-5, silu=False, iters=1):
    x, weight, bias = f16(x), f32(weight), f32(bias)
    B, Cn, H, W = x.shape
    out = np.empty_like(x)
    ms = C.c_float(0)
    check(lib().sd_op_groupnorm(ptr(x), fptr(weight), fptr(bias), ptr(out), B, Cn, H, W, groups, eps, int(silu),
                                iters, C.byref(ms)))
    return out, ms.value


def conv2d(x, w, bias=None, res=None, stride=1, upsample=False, tile=0, splitk=0, force_generic=False, iters=1):
    x, w = f16(x), f16(w)
    B, Cin, H, W = x.shape
    Cout, Cin2, k, k2 = w.shape
    if Cin2 != Cin or k != k2:
        raise ValueError("conv2d: weight shape does not match input")
    up = 2 if upsample else 1
    pad = k // 2
    Ho = (H * up + 2 * pad - k) // stride + 1
    Wo = (W * up + 2 * pad - k) // stride + 1
    bias = None if bias is None else f32(bias)
    res = None if res is None else f16(res)
    out = np.empty((B, Cout, Ho, Wo), np.float16)
    ms = C.c_float(0)
    check(lib().sd_op_conv2d(ptr(x), ptr(w), fptr(bias), ptr(res), ptr(out), B, Cin, H, W, Cout, k, stride,
                             int(upsample), tile, splitk, int(force_generic), iters, C.byref(ms)))
    return out, ms.value


def groupnorm_shortcut(x0, x1, gn_weight, gn_bias, w, bias=None, groups=32, eps=1e-5, silu=True, side=True, iters=1):
    """norm1 (+SiLU) over the concat (x0 | x1) and conv_shortcut (1x1) over the same concat; side=True: one launch.
    Returns (normalised tensor, shortcut output, ms)."""
    x0, w = f16(x0), f16(w)
    x1 = None if x1 is None else f16(x1)
    B, C0, H, W = x0.shape
    C1 = 0 if x1 is None else x1.shape[1]
    N = w.shape[0]
    w = np.ascontiguousarray(w.reshape(N, -1))
    if w.shape[1] != C0 + C1 or (x1 is not None and x1.shape != (B, C1, H, W)):
        raise ValueError("groupnorm_shortcut: inconsistent shapes")
    gn_weight, gn_bias = f32(gn_weight), f32(gn_bias)
    bias = None if bias is None else f32(bias)
    out_gn = np.empty((B, C0 + C1, H, W), np.float16)
    out_sc = np.empty((B, N, H, W), np.float16)
    ms = C.c_float(0)
    check(lib().sd_op_groupnorm_shortcut(ptr(x0), ptr(x1), fptr(gn_weight), fptr(gn_bias), ptr(w), fptr(bias), ptr(out_gn), ptr(out_sc),
                                         B, C0, C1, H, W, N, groups, eps, int(silu), int(side), iters, C.byref(ms)))
    return out_gn, out_sc, ms.value


def conv2d_groupnorm(x, w, gn_weight, gn_bias, bias=None, res=None, groups=32, eps=1e-5, silu=False, tile=0,
                     producer_stats=True, iters=1):
    """conv (stride 1) -> GroupNorm (+ SiLU); producer_stats: GroupNorm statistics from the conv kernel's epilogue.
    Returns (conv output, normalised output, entries the conv wrote per (sample, group), ms)."""
    x, w = f16(x), f16(w)
    B, Cin, H, W = x.shape
    Cout, Cin2, k, k2 = w.shape
    if Cin2 != Cin or k != k2:
        raise ValueError("conv2d_groupnorm: weight shape does not match input")
    bias = None if bias is None else f32(bias)
    res = None if res is None else f16(res)
    gn_weight, gn_bias = f32(gn_weight), f32(gn_bias)
    conv_out = np.empty((B, Cout, H, W), np.float16)
    out = np.empty((B, Cout, H, W), np.float16)
    ms, entries = C.c_float(0), C.c_int(0)
    check(lib().sd_op_conv2d_groupnorm(ptr(x), ptr(w), fptr(bias), ptr(res), fptr(gn_weight), fptr(gn_bias), ptr(conv_out),
                                       ptr(out), B, Cin, H, W, Cout, k, groups, eps, int(silu), tile, int(producer_stats),
                                       C.byref(entries), iters, C.byref(ms)))
    return conv_out, out, entries.value, ms.value


def conv2d_groupnorm_proj(x, w, gn_weight, gn_bias, proj_w, proj_bias=None, bias=None, res=None, groups=32, eps=1e-6, fold=True,
                          tile=0, iters=1):
    """proj(GroupNorm(conv(x))) - a resnet's last conv followed by SpatialTransformer.norm + proj_in; fold=True applies the
    GroupNorm inside the projection GEMM.  Returns (conv_out, out, entries consumed by the fold, ms)."""
    x, w, proj_w = f16(x), f16(w), f16(proj_w)
    B, Cin, H, W = x.shape
    Cout, k = w.shape[0], w.shape[2]
    Np = proj_w.shape[0]
    if w.shape[1] != Cin or proj_w.reshape(Np, -1).shape[1] != Cout:
        raise ValueError("conv2d_groupnorm_proj: inconsistent shapes")
    proj_w = np.ascontiguousarray(proj_w.reshape(Np, Cout))
    gn_weight, gn_bias = f32(gn_weight), f32(gn_bias)
    bias = None if bias is None else f32(bias)
    proj_bias = None if proj_bias is None else f32(proj_bias)
    res = None if res is None else f16(res)
    conv_out = np.empty((B, Cout, H, W), np.float16)
    out = np.empty((B, Np, H, W), np.float16)
    ms, entries = C.c_float(0), C.c_int(0)
    check(lib().sd_op_conv2d_groupnorm_proj(ptr(x), ptr(w), fptr(bias), ptr(res), fptr(gn_weight), fptr(gn_bias), ptr(proj_w),
                                            fptr(proj_bias), ptr(conv_out), ptr(out), B, Cin, H, W, Cout, k, Np, groups, eps,
                                            int(fold), tile, C.byref(entries), iters, C.byref(ms)))
    return conv_out, out, entries.value, ms.value


def conv2d_groupnorm_conv3x3(x, w, gn_weight, gn_bias, w2, bias2=None, res2=None, bias=None, res=None, groups=32, eps=1e-5, silu=True,
                             fold=True, tile=0, staging2=0, iters=1):
    """conv3x3(silu(GroupNorm(conv(x)))) - a resnet's norm -> SiLU -> conv behind its producer; fold=True applies the GroupNorm
    (+ SiLU) in the halo loader of the second conv.  Returns (conv_out, out, entries consumed by the loader, ms)."""
    x, w, w2 = f16(x), f16(w), f16(w2)
    B, Cin, H, W = x.shape
    Cout, k = w.shape[0], w.shape[2]
    N2 = w2.shape[0]
    if w.shape[1] != Cin or w2.shape[1:] != (Cout, 3, 3):
        raise ValueError("conv2d_groupnorm_conv3x3: inconsistent shapes")
    gn_weight, gn_bias = f32(gn_weight), f32(gn_bias)
    bias = None if bias is None else f32(bias)
    bias2 = None if bias2 is None else f32(bias2)
    res = None if res is None else f16(res)
    res2 = None if res2 is None else f16(res2)
    conv_out = np.empty((B, Cout, H, W), np.float16)
    out = np.empty((B, N2, H, W), np.float16)
    ms, entries = C.c_float(0), C.c_int(0)
    check(lib().sd_op_conv2d_groupnorm_conv3x3(ptr(x), ptr(w), fptr(bias), ptr(res), fptr(gn_weight), fptr(gn_bias), ptr(w2), fptr(bias2),
                                               ptr(res2), ptr(conv_out), ptr(out), B, Cin, H, W, Cout, k, N2, groups, eps, int(silu),
                                               int(fold), tile, staging2, C.byref(entries), iters, C.byref(ms)))
    return conv_out, out, entries.value, ms.value


def cross_attention_block(x, ln_weight, ln_bias, wq, k, v, wo, bo, heads, eps=1e-5, fused=True, iters=1, a1=None, wo1=None, bo1=None):
    """x + to_out(softmax(to_q(LayerNormANE(x)) k^T / 8) v) + bo - the cross-attention branch of a BasicTransformerBlock; fused=True
    runs it as ONE launch (5 or 10 heads of 64, Sq % 32 == 0), else as the q-projection + attention launch and the to_out GEMM.
    With a1 / wo1 / bo1 the self-attention's output projection comes first: the branch runs on h1 = x + to_out1(a1) + bo1 (one
    launch: 5 heads only).  x, a1 (B,C,1,Sq), k / v (B,C,1,Sk) f16, wq / wo / wo1 (C,C) f16, bo / bo1 (C) f32.  Returns (out, ms)."""
    x, k, v, wq, wo = f16(x), f16(k), f16(v), f16(wq), f16(wo)
    B, Cn, _, Sq = x.shape
    Sk = k.shape[3]
    if Cn != heads * 64 or k.shape[1] != Cn or v.shape != k.shape or wq.shape != (Cn, Cn) or wo.shape != (Cn, Cn):
        raise ValueError("cross_attention_block: inconsistent shapes")
    ln_weight, ln_bias, bo = f32(ln_weight), f32(ln_bias), f32(bo)
    if a1 is not None:
        a1, wo1, bo1 = f16(a1), f16(wo1), f32(bo1)
        if a1.shape != x.shape or wo1.shape != (Cn, Cn):
            raise ValueError("cross_attention_block: inconsistent shapes of a1 / wo1")
    out = np.empty_like(x)
    ms = C.c_float(0)
    check(lib().sd_op_cross_attention_block(ptr(x), fptr(ln_weight), fptr(ln_bias), ptr(wq), ptr(k), ptr(v), ptr(wo), fptr(bo), ptr(a1), ptr(wo1),
                                            fptr(bo1), ptr(out), B, heads, Sq, Sk, eps, int(fused), iters, C.byref(ms)))
    return out, ms.value


def ffn_out_proj(g, w1, b1, res1, w2, b2, res2, groups=0, fused=True, iters=1):
    """res2 + proj_out(res1 + ff.net.2(g) + b1) + b2 - the tail of a SpatialTransformer; fused=True runs it as ONE launch (C = 320,
    S % 32 == 0).  g (B,4C,1,S), res1 / res2 (B,C,1,S) f16, w1 (C,4C), w2 (C,C) f16, b1 / b2 (C) f32.  groups > 0: also returns the
    GroupNorm statistics the launch left for its consumer, folded: (B, groups, 2) = (sum, sum of squares).  Returns (out, gn_sums or None, ms)."""
    g, res1, res2, w1, w2 = f16(g), f16(res1), f16(res2), f16(w1), f16(w2)
    B, Cn, _, S = res1.shape
    if g.shape != (B, 4 * Cn, 1, S) or res2.shape != res1.shape or w1.shape != (Cn, 4 * Cn) or w2.shape != (Cn, Cn):
        raise ValueError("ffn_out_proj: inconsistent shapes")
    b1, b2 = f32(b1), f32(b2)
    out = np.empty_like(res1)
    sums = np.empty((B, groups, 2), np.float32) if groups > 0 else None
    ms = C.c_float(0)
    check(lib().sd_op_ffn_out_proj(ptr(g), ptr(w1), fptr(b1), ptr(res1), ptr(w2), fptr(b2), ptr(res2), ptr(out), fptr(sums), B, Cn, S, groups,
                                   int(fused), iters, C.byref(ms)))
    return out, sums, ms.value


def cross_attention_fused(x, ln_weight, ln_bias, wq, k, v, heads, eps=1e-5, nst=0, iters=1):
    """softmax(to_q(LayerNormANE(x)) k^T / 8) v per head (head dim 64) as one launch.  x (B,C,1,Sq), k/v (B,C,1,Sk)."""
    x, k, v, wq = f16(x), f16(k), f16(v), f16(wq)
    B, Cn, _, Sq = x.shape
    Sk = k.shape[3]
    if Cn != heads * 64 or k.shape[1] != Cn or v.shape != k.shape or wq.shape != (Cn, Cn):
        raise ValueError("cross_attention_fused: inconsistent shapes")
    ln_weight, ln_bias = f32(ln_weight), f32(ln_bias)
    out = np.empty_like(x)
    ms = C.c_float(0)
    check(lib().sd_op_cross_attention_fused(ptr(x), fptr(ln_weight), fptr(ln_bias), ptr(wq), ptr(k), ptr(v), ptr(out), B, heads,
                                            Sq, Sk, eps, nst, iters, C.byref(ms)))
    return out, ms.value


def geglu(x, w, bias=None, iters=1):
    x, w = f16(x), f16(w)
    M, Cn = x.shape
    N2 = w.shape[0]
    bias = None if bias is None else f32(bias)
    out = np.empty((M, N2 // 2), np.float16)
    ms = C.c_float(0)
    check(lib().sd_op_geglu(ptr(x), ptr(w), fptr(bias), ptr(out), M, Cn, N2, iters, C.byref(ms)))
    return out, ms.value


def geglu_ln(x, w, bias=None, ln_weight=None, ln_bias=None, eps=1e-5, kernel=0, iters=1):
    """GEGLU projection with the LayerNorm in front of it folded in (unet.py:583-591 -> :609-617).  kernel: 0 the library's plan,
    1 the tiled GEMM kernels, 2 the weight-stationary kernel (wsgemm.hip)."""
    x, w = f16(x), f16(w)
    M, Cn = x.shape
    N2 = w.shape[0]
    bias = None if bias is None else f32(bias)
    ln_weight = None if ln_weight is None else f32(ln_weight)
    ln_bias = None if ln_bias is None else f32(ln_bias)
    out = np.empty((M, N2 // 2), np.float16)
    ms = C.c_float(0)
    check(lib().sd_op_geglu_ln(ptr(x), fptr(ln_weight), fptr(ln_bias), ptr(w), fptr(bias), ptr(out), M, Cn, N2, eps, kernel, iters,
                               C.byref(ms)))
    return out, ms.value


def qkv_ln(x, ln_weight, ln_bias, w, batch, q_scale=1.0, vt_perm=True, eps=1e-5, kernel=0, iters=1):
    """Fused q|k|v projection with the LayerNorm in front of it folded in (unet.py:583-586 -> :74-84).  x (batch * HW, C), w (3C, C).
    Returns (out_qk (batch * HW, 2C), out_vt (batch, C, HW), ms)."""
    x, w = f16(x), f16(w)
    M, Cn = x.shape
    HW = M // batch
    out_qk = np.empty((M, 2 * Cn), np.float16)
    out_vt = np.empty((batch, Cn, HW), np.float16)
    ms = C.c_float(0)
    check(lib().sd_op_qkv_ln(ptr(x), fptr(f32(ln_weight)), fptr(f32(ln_bias)), ptr(w), ptr(out_qk), ptr(out_vt), batch, HW, Cn, eps, q_scale,
                             int(vt_perm), kernel, iters, C.byref(ms)))
    return out_qk, out_vt, ms.value


def gn_proj_qkv(x_in, conv_w, gn_weight, gn_bias, proj_w, proj_bias, ln_weight, ln_bias, wqkv, groups=32, gn_eps=1e-6, ln_eps=1e-5,
                q_scale=1.0, vt_perm=True, fused=True, iters=1):
    """conv1x1 (producer, leaves GroupNorm statistics) -> GroupNorm -> proj_in -> LayerNorm -> fused q|k|v; fused: the last four in ONE
    launch (2 / 3: its 64- / 32-token form).  Returns (h (B * HW, C), qk (B * HW, 2C), vt (B, C, HW), entries, ms)."""
    x_in, conv_w, proj_w, wqkv = f16(x_in), f16(conv_w), f16(proj_w), f16(wqkv)
    B, Cn, H, W = x_in.shape
    M = B * H * W
    h = np.empty((M, Cn), np.float16)
    qk = np.empty((M, 2 * Cn), np.float16)
    vt = np.empty((B, Cn, H * W), np.float16)
    ms, entries = C.c_float(0), C.c_int(0)
    check(lib().sd_op_gn_proj_qkv(ptr(x_in), ptr(conv_w), fptr(f32(gn_weight)), fptr(f32(gn_bias)), ptr(proj_w), fptr(f32(proj_bias)),
                                  fptr(f32(ln_weight)), fptr(f32(ln_bias)), ptr(wqkv), ptr(h), ptr(qk), ptr(vt), B, H, W, Cn, groups, gn_eps,
                                  ln_eps, q_scale, int(vt_perm), int(fused), C.byref(entries), iters, C.byref(ms)))
    return h, qk, vt, entries.value, ms.value


def timestep_embedding(t, dim, flip_sin_to_cos=True, freq_shift=0.0):
    t = f32(t)
    out = np.empty((t.shape[0], dim), np.float32)
    check(lib().sd_op_timestep_embedding(fptr(t), fptr(out), t.shape[0], dim, int(flip_sin_to_cos), freq_shift))
    return out


def numpy_randn(seed, n):
    out = np.empty(n, np.float64)
    check(lib().sd_numpy_randn(seed, out.ctypes.data_as(C.POINTER(C.c_double)), n))
    return out


def torch_randn(seed, n):
    """torch.manual_seed(seed); torch.randn(n) on the CPU (TorchRandomSource.swift)."""
    out = np.empty(n, np.float64)
    check(lib().sd_torch_randn(seed, out.ctypes.data_as(C.POINTER(C.c_double)), n))
    return out


def philox_randn(seed, n, offset=0):
    """torch.randn on a CUDA device (NvRandomSource.swift); offset = arrays drawn before this one."""
    out = np.empty(n, np.float64)
    check(lib().sd_philox_randn(seed, offset, out.ctypes.data_as(C.POINTER(C.c_double)), n))
    return out


def calibrate(device=0):
    """Nine fixed micro-measurements of the box (calib.hip): what bench.py prints as `calibration` and normalises by."""
    out = (C.c_float * 9)()
    check(lib().sd_calibrate(device, out))
    return {"copy_gbs": round(out[0], 1), "mfma_tflops": round(out[1], 1), "empty_launch_us": round(out[2], 3),
            "chain_us": round(out[3], 3), "handover_us": round(out[4], 3), "latency_hbm_ns": round(out[5], 1),
            "latency_cache_ns": round(out[6], 1), "small_grid_us": round(out[7], 3), "cold_code_us": round(out[8], 3)}
