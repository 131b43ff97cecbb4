"""Oracle (test infrastructure): UNet2DConditionModel(+XL) and ControlNet as a functional
torch-CPU fp32 graph over a flat ``{diffusers key: tensor}`` state dict.

Restates (never imports) the reference:
  layer_norm   python_coreml_stable_diffusion/layer_norm.py:51-80 (+ bias hook unet.py:132-146)
  resnet       unet.py:406-489         up/down-sample unet.py:492-510
  transformer  unet.py:513-617         time embed     unet.py:630-728
  blocks       unet.py:151-403,731-795 UNet forward   unet.py:975-1048, XL :1051-1152
  controlnet   controlnet.py:15-47, 199-250
Layouts are the reference's: NCHW images, BC1S sequences.  The state dict holds the ORIGINAL
checkpoint tensors (LayerNorm as x_hat*w+b, which is what the reference computes after its
load hook rewrites b'=b/w and applies (x_hat+b')*w; Linear weights may be 2-D or 4-D).

Why torch and not numpy/C: the path is fp32/fp16 floating point and torch-CPU is the substrate
the reference itself uses for its parity checks (torch2coreml.py:970-975).  Pinned against the
real reference modules by ``oracle/pin_against_reference.py`` (max |diff| <= 2e-5 fp32) and by
``tests/golden/unet_*.npz``.
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

from . import attention_ref

# --------------------------------------------------------------------------------------
# Configs (public HF config.json values; SURVEY.md Appendix B, validated by parameter count)
# --------------------------------------------------------------------------------------
CA, DN, UP, CAUP = "CrossAttnDownBlock2D", "DownBlock2D", "UpBlock2D", "CrossAttnUpBlock2D"


def make_config(**kw):
    cfg = dict(
        in_channels=4, out_channels=4, sample_size=64,
        block_out_channels=(320, 640, 1280, 1280),
        down_block_types=(CA, CA, CA, DN), up_block_types=(UP, CAUP, CAUP, CAUP),
        layers_per_block=2, attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024,
        transformer_layers_per_block=1, norm_num_groups=32, norm_eps=1e-5,
        flip_sin_to_cos=True, freq_shift=0,
        addition_embed_type=None, addition_time_embed_dim=None,
        projection_class_embeddings_input_dim=None, support_controlnet=False,
    )
    cfg.update(kw)
    n = len(cfg["block_out_channels"])
    for key in ("attention_head_dim", "transformer_layers_per_block"):
        if isinstance(cfg[key], int):
            cfg[key] = (cfg[key],) * n
        cfg[key] = tuple(cfg[key])
    cfg["block_out_channels"] = tuple(cfg["block_out_channels"])
    return cfg


CONFIGS = {
    "sd21-base": make_config(),
    "sd15": make_config(attention_head_dim=8, cross_attention_dim=768),
    "sd15-control": make_config(attention_head_dim=8, cross_attention_dim=768, support_controlnet=True),
    "sdxl-base": make_config(
        sample_size=128, block_out_channels=(320, 640, 1280), down_block_types=(DN, CA, CA),
        up_block_types=(CAUP, CAUP, UP), attention_head_dim=(5, 10, 20), cross_attention_dim=2048,
        transformer_layers_per_block=(1, 2, 10), addition_embed_type="text_time",
        addition_time_embed_dim=256, projection_class_embeddings_input_dim=2816),
    "sdxl-refiner": make_config(
        sample_size=128, block_out_channels=(384, 768, 1536, 1536), down_block_types=(DN, CA, CA, DN),
        up_block_types=(UP, CAUP, CAUP, UP), attention_head_dim=(6, 12, 24, 24), cross_attention_dim=1280,
        transformer_layers_per_block=4, addition_embed_type="text_time",
        addition_time_embed_dim=256, projection_class_embeddings_input_dim=2560),
    # kernel-realistic miniature (d_head 64, channels % 64 == 0) for fast parity runs
    "mini": make_config(sample_size=16, block_out_channels=(64, 128, 256, 256),
                        attention_head_dim=(1, 2, 4, 4), cross_attention_dim=128),
    "mini-xl": make_config(
        sample_size=16, block_out_channels=(64, 128, 256), down_block_types=(DN, CA, CA),
        up_block_types=(CAUP, CAUP, UP), attention_head_dim=(1, 2, 4), cross_attention_dim=128,
        transformer_layers_per_block=(1, 2, 2), addition_embed_type="text_time",
        addition_time_embed_dim=32, projection_class_embeddings_input_dim=64 + 6 * 32),
    "mini-control": make_config(sample_size=16, block_out_channels=(64, 128, 256, 256),
                                attention_head_dim=(1, 2, 4, 4), cross_attention_dim=128,
                                support_controlnet=True),
    # SDXL-refiner topology in miniature (4 levels DN,CA,CA,DN, depth 2, 5 time ids -> 32*5 + 64)
    "mini-refiner": make_config(
        sample_size=16, block_out_channels=(64, 128, 256, 256), down_block_types=(DN, CA, CA, DN),
        up_block_types=(UP, CAUP, CAUP, UP), attention_head_dim=(1, 2, 4, 4), cross_attention_dim=128,
        transformer_layers_per_block=2, addition_embed_type="text_time",
        addition_time_embed_dim=32, projection_class_embeddings_input_dim=64 + 5 * 32),
    # odd little config: exercises the generic (non-MFMA) fallbacks and d_head != 64
    "tiny": make_config(sample_size=8, block_out_channels=(32, 64), down_block_types=(CA, DN),
                        up_block_types=(UP, CAUP), layers_per_block=1, attention_head_dim=(2, 4),
                        cross_attention_dim=48),
}


# --------------------------------------------------------------------------------------
# Parameter inventory (diffusers key names, unet.py / SURVEY.md Appendix A.7)
# --------------------------------------------------------------------------------------
def _conv(sh, name, cin, cout, k=1, bias=True):
    sh[name + ".weight"] = (cout, cin, k, k)
    if bias:
        sh[name + ".bias"] = (cout,)


def _norm(sh, name, c):
    sh[name + ".weight"] = (c,)
    sh[name + ".bias"] = (c,)


def _resnet(sh, p, cin, cout, temb):
    _norm(sh, p + ".norm1", cin)
    _conv(sh, p + ".conv1", cin, cout, 3)
    _conv(sh, p + ".time_emb_proj", temb, cout, 1)
    _norm(sh, p + ".norm2", cout)
    _conv(sh, p + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(sh, p + ".conv_shortcut", cin, cout, 1)


def _transformer(sh, p, c, ctx, depth):
    _norm(sh, p + ".norm", c)
    _conv(sh, p + ".proj_in", c, c, 1)
    for d in range(depth):
        b = f"{p}.transformer_blocks.{d}"
        for i, kv in ((1, c), (2, ctx)):
            _conv(sh, f"{b}.attn{i}.to_q", c, c, 1, bias=False)
            _conv(sh, f"{b}.attn{i}.to_k", kv, c, 1, bias=False)
            _conv(sh, f"{b}.attn{i}.to_v", kv, c, 1, bias=False)
            _conv(sh, f"{b}.attn{i}.to_out.0", c, c, 1)
            _norm(sh, f"{b}.norm{i}", c)
        _norm(sh, f"{b}.norm3", c)
        _conv(sh, f"{b}.ff.net.0.proj", c, 8 * c, 1)
        _conv(sh, f"{b}.ff.net.2", 4 * c, c, 1)
    _conv(sh, p + ".proj_out", c, c, 1)


def _down_and_mid_shapes(sh, cfg):
    boc = cfg["block_out_channels"]
    temb = boc[0] * 4
    ctx = cfg["cross_attention_dim"]
    _conv(sh, "conv_in", cfg["in_channels"], boc[0], 3)
    _conv(sh, "time_embedding.linear_1", boc[0], temb)
    _conv(sh, "time_embedding.linear_2", temb, temb)
    if cfg["addition_embed_type"] == "text_time":
        _conv(sh, "add_embedding.linear_1", cfg["projection_class_embeddings_input_dim"], temb)
        _conv(sh, "add_embedding.linear_2", temb, temb)
    out = boc[0]
    for i, t in enumerate(cfg["down_block_types"]):
        cin, out = out, boc[i]
        for j in range(cfg["layers_per_block"]):
            _resnet(sh, f"down_blocks.{i}.resnets.{j}", cin if j == 0 else out, out, temb)
            if t == CA:
                _transformer(sh, f"down_blocks.{i}.attentions.{j}", out, ctx,
                             cfg["transformer_layers_per_block"][i])
        if i != len(boc) - 1:
            _conv(sh, f"down_blocks.{i}.downsamplers.0.conv", out, out, 3)
    c = boc[-1]
    _resnet(sh, "mid_block.resnets.0", c, c, temb)
    _transformer(sh, "mid_block.attentions.0", c, ctx, cfg["transformer_layers_per_block"][-1])
    _resnet(sh, "mid_block.resnets.1", c, c, temb)


def unet_param_shapes(cfg):
    """Ordered {key: shape} of a UNet checkpoint for ``cfg`` (conv-shaped 4-D weights)."""
    sh = OrderedDict()
    _down_and_mid_shapes(sh, cfg)
    boc = cfg["block_out_channels"]
    temb = boc[0] * 4
    ctx = cfg["cross_attention_dim"]
    rev = list(reversed(boc))
    rev_depth = list(reversed(cfg["transformer_layers_per_block"]))
    out = rev[0]
    n = len(boc)
    for i, t in enumerate(cfg["up_block_types"]):
        prev, out = out, rev[i]
        cin = rev[min(i + 1, n - 1)]
        nl = cfg["layers_per_block"] + 1
        for j in range(nl):
            skip = cin if j == nl - 1 else out
            rin = prev if j == 0 else out
            _resnet(sh, f"up_blocks.{i}.resnets.{j}", rin + skip, out, temb)
            if t == CAUP:
                _transformer(sh, f"up_blocks.{i}.attentions.{j}", out, ctx, rev_depth[i])
        if i != n - 1:
            _conv(sh, f"up_blocks.{i}.upsamplers.0.conv", out, out, 3)
    _norm(sh, "conv_norm_out", boc[0])
    _conv(sh, "conv_out", boc[0], cfg["out_channels"], 3)
    return sh


def controlnet_param_shapes(cfg, cond_channels=(16, 32, 96, 256)):
    """controlnet.py:49-189: UNet down+mid + conditioning embedding + zero-conv taps."""
    sh = OrderedDict()
    _down_and_mid_shapes(sh, cfg)
    boc = cfg["block_out_channels"]
    p = "controlnet_cond_embedding"
    _conv(sh, p + ".conv_in", 3, cond_channels[0], 3)
    for i in range(len(cond_channels) - 1):
        _conv(sh, f"{p}.blocks.{2 * i}", cond_channels[i], cond_channels[i], 3)
        _conv(sh, f"{p}.blocks.{2 * i + 1}", cond_channels[i], cond_channels[i + 1], 3)
    _conv(sh, p + ".conv_out", cond_channels[-1], boc[0], 3)
    taps = [boc[0]]
    for i in range(len(boc)):
        taps += [boc[i]] * cfg["layers_per_block"]
        if i != len(boc) - 1:
            taps.append(boc[i])
    for i, c in enumerate(taps):
        _conv(sh, f"controlnet_down_blocks.{i}", c, c, 1)
    _conv(sh, "controlnet_mid_block", boc[-1], boc[-1], 1)
    return sh


def residual_shapes(cfg, batch, h=None, w=None):
    """Shapes of the 12+1 ControlNet residuals == UNet skip tensors (unet.py:1009-1022)."""
    h = h or cfg["sample_size"]
    w = w or cfg["sample_size"]
    boc = cfg["block_out_channels"]
    shapes = [(batch, boc[0], h, w)]
    for i in range(len(boc)):
        shapes += [(batch, boc[i], h, w)] * cfg["layers_per_block"]
        if i != len(boc) - 1:
            h, w = (h + 1) // 2, (w + 1) // 2
            shapes.append((batch, boc[i], h, w))
    shapes.append((batch, boc[-1], h, w))
    return shapes


# --------------------------------------------------------------------------------------
# Functional graph
# --------------------------------------------------------------------------------------
def _w4(w):
    return w if w.dim() == 4 else w[:, :, None, None]      # unet.py:121-127


def conv(sd, name, x, stride=1, padding=0):
    return F.conv2d(x, _w4(sd[name + ".weight"]), sd.get(name + ".bias"), stride=stride, padding=padding)


def layer_norm_ane(x, weight, bias, eps=1e-5):
    """layer_norm.py:51-80 over the channel dim of BC1S; with the unet.py:132-138 hook this is
    algebraically x_hat*w + b on the original checkpoint tensors."""
    mu = x.mean(dim=1, keepdim=True)
    zm = x - mu
    denom = (zm * zm).mean(dim=1, keepdim=True).add(eps).rsqrt()
    return zm * denom * weight.view(1, -1, 1, 1) + bias.view(1, -1, 1, 1)


def group_norm(sd, name, x, eps, groups=32):
    return F.group_norm(x, groups, sd[name + ".weight"], sd[name + ".bias"], eps)


def timestep_embedding(t, dim, flip_sin_to_cos=True, freq_shift=0):
    """unet.py:703-728: fp32 sinusoid, [sin|cos] then flipped to [cos|sin]."""
    half = dim // 2
    exponent = -math.log(10000) * torch.arange(half, dtype=torch.float32) / (half - freq_shift)
    arg = t[:, None].float() * torch.exp(exponent)[None, :]
    emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    return emb


def time_mlp(sd, name, x):
    """unet.py:665-682 TimestepEmbedding: 1x1 conv -> SiLU -> 1x1 conv on (B,C,1,1)."""
    x = x[:, :, None, None]
    return conv(sd, name + ".linear_2", F.silu(conv(sd, name + ".linear_1", x)))


def resnet(sd, p, x, temb, eps):
    """unet.py:470-489."""
    h = conv(sd, p + ".conv1", F.silu(group_norm(sd, p + ".norm1", x, eps)), padding=1)
    h = h + conv(sd, p + ".time_emb_proj", F.silu(temb))
    h = conv(sd, p + ".conv2", F.silu(group_norm(sd, p + ".norm2", h, eps)), padding=1)
    if p + ".conv_shortcut.weight" in sd:
        x = conv(sd, p + ".conv_shortcut", x)
    return x + h


def cross_attention(sd, p, x, context, heads, impl):
    """unet.py:87-118; Einsum dispatch unet.py:51-59."""
    ctx = x if context is None else context
    q = conv(sd, p + ".to_q", x)
    k = conv(sd, p + ".to_k", ctx)
    v = conv(sd, p + ".to_v", ctx)
    d = q.shape[1] // heads
    o = attention_ref.IMPLS[impl](q, k, v, heads, d)
    return conv(sd, p + ".to_out.0", o)


def transformer_block(sd, b, x, context, heads, impl):
    """unet.py:586-591 + GEGLU unet.py:609-617 (exact-erf GELU, value = first half)."""
    x = cross_attention(sd, b + ".attn1", layer_norm_ane(x, sd[b + ".norm1.weight"], sd[b + ".norm1.bias"]),
                        None, heads, impl) + x
    x = cross_attention(sd, b + ".attn2", layer_norm_ane(x, sd[b + ".norm2.weight"], sd[b + ".norm2.bias"]),
                        context, heads, impl) + x
    h = conv(sd, b + ".ff.net.0.proj", layer_norm_ane(x, sd[b + ".norm3.weight"], sd[b + ".norm3.bias"]))
    val, gate = h.chunk(2, dim=1)
    return conv(sd, b + ".ff.net.2", val * F.gelu(gate)) + x


def spatial_transformer(sd, p, x, context, heads, depth, impl):
    """unet.py:553-563; GroupNorm eps hard-coded 1e-6 (unet.py:528-531)."""
    b, c, hh, ww = x.shape
    h = conv(sd, p + ".proj_in", group_norm(sd, p + ".norm", x, 1e-6))
    h = h.reshape(b, c, 1, hh * ww)
    for d in range(depth):
        h = transformer_block(sd, f"{p}.transformer_blocks.{d}", h, context, heads, impl)
    h = conv(sd, p + ".proj_out", h.reshape(b, c, hh, ww))
    return h + x


def _down_and_mid(sd, cfg, sample, emb, ehs, impl):
    eps = cfg["norm_eps"]
    skips = [sample]
    for i, t in enumerate(cfg["down_block_types"]):
        for j in range(cfg["layers_per_block"]):
            sample = resnet(sd, f"down_blocks.{i}.resnets.{j}", sample, emb, eps)
            if t == CA:
                sample = spatial_transformer(sd, f"down_blocks.{i}.attentions.{j}", sample, ehs,
                                             cfg["attention_head_dim"][i],
                                             cfg["transformer_layers_per_block"][i], impl)
            skips.append(sample)
        if i != len(cfg["block_out_channels"]) - 1:
            sample = conv(sd, f"down_blocks.{i}.downsamplers.0.conv", sample, stride=2, padding=1)
            skips.append(sample)
    # mid (unet.py:789-795); the reference passes attention_head_dim[i] with the leaked loop
    # variable i == last block (unet.py:929)
    sample = resnet(sd, "mid_block.resnets.0", sample, emb, eps)
    sample = spatial_transformer(sd, "mid_block.attentions.0", sample, ehs, cfg["attention_head_dim"][-1],
                                 cfg["transformer_layers_per_block"][-1], impl)
    sample = resnet(sd, "mid_block.resnets.1", sample, emb, eps)
    return sample, skips


def _time_embedding(sd, cfg, timestep, time_ids=None, text_embeds=None):
    boc = cfg["block_out_channels"]
    t_emb = timestep_embedding(timestep, boc[0], cfg["flip_sin_to_cos"], cfg["freq_shift"])
    emb = time_mlp(sd, "time_embedding", t_emb)
    if cfg["addition_embed_type"] == "text_time":                       # unet.py:1076-1088
        te = timestep_embedding(time_ids.flatten(), cfg["addition_time_embed_dim"],
                                cfg["flip_sin_to_cos"], cfg["freq_shift"])
        te = te.reshape(text_embeds.shape[0], -1)
        emb = emb + time_mlp(sd, "add_embedding", torch.cat([text_embeds.float(), te], dim=-1))
    return emb


@torch.no_grad()
def unet_forward(sd, cfg, sample, timestep, encoder_hidden_states, time_ids=None, text_embeds=None,
                 additional_residuals=None, impl="ORIGINAL"):
    """unet.py:975-1048 (XL: 1055-1152).  sample (B,4,H,W); timestep (B,);
    encoder_hidden_states (B,Cctx,1,77) BC1S.  Returns noise_pred (B,4,H,W) fp32."""
    sample = sample.float()
    ehs = encoder_hidden_states.float()
    emb = _time_embedding(sd, cfg, timestep.float(), time_ids, text_embeds)
    sample = conv(sd, "conv_in", sample, padding=1)
    sample, skips = _down_and_mid(sd, cfg, sample, emb, ehs, impl)
    if cfg["support_controlnet"]:                                       # unet.py:1009-1022
        skips = [s + r.float() for s, r in zip(skips, additional_residuals[:-1])]
        sample = sample + additional_residuals[-1].float()
    eps = cfg["norm_eps"]
    rev_heads = list(reversed(cfg["attention_head_dim"]))
    rev_depth = list(reversed(cfg["transformer_layers_per_block"]))
    n = len(cfg["block_out_channels"])
    for i, t in enumerate(cfg["up_block_types"]):
        for j in range(cfg["layers_per_block"] + 1):
            sample = torch.cat([sample, skips.pop()], dim=1)            # unet.py:213-216
            sample = resnet(sd, f"up_blocks.{i}.resnets.{j}", sample, emb, eps)
            if t == CAUP:
                sample = spatial_transformer(sd, f"up_blocks.{i}.attentions.{j}", sample, ehs,
                                             rev_heads[i], rev_depth[i], impl)
        if i != n - 1:                                                   # unet.py:498-500
            sample = F.interpolate(sample, scale_factor=2.0, mode="nearest")
            sample = conv(sd, f"up_blocks.{i}.upsamplers.0.conv", sample, padding=1)
    sample = F.silu(group_norm(sd, "conv_norm_out", sample, eps))
    return conv(sd, "conv_out", sample, padding=1)


@torch.no_grad()
def controlnet_cond_embedding(sd, cond):
    """controlnet.py:35-47."""
    p = "controlnet_cond_embedding"
    e = F.silu(conv(sd, p + ".conv_in", cond.float(), padding=1))
    i = 0
    while f"{p}.blocks.{i}.weight" in sd:
        e = F.silu(conv(sd, f"{p}.blocks.{i}", e, stride=1 + (i % 2), padding=1))
        i += 1
    return conv(sd, p + ".conv_out", e, padding=1)


@torch.no_grad()
def controlnet_forward(sd, cfg, sample, timestep, encoder_hidden_states, controlnet_cond, impl="ORIGINAL"):
    """controlnet.py:199-250 -> list of 12 down residuals + [mid residual]."""
    emb = _time_embedding(sd, cfg, timestep.float())
    sample = conv(sd, "conv_in", sample.float(), padding=1) + controlnet_cond_embedding(sd, controlnet_cond)
    sample, skips = _down_and_mid(sd, cfg, sample, emb, encoder_hidden_states.float(), impl)
    res = [conv(sd, f"controlnet_down_blocks.{i}", s) for i, s in enumerate(skips)]
    res.append(conv(sd, "controlnet_mid_block", sample))
    return res
