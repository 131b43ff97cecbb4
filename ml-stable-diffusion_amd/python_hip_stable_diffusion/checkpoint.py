"""Checkpoint inventory of the models on the path (diffusers key names, SURVEY.md Appendix A.7;
module names mirror python_coreml_stable_diffusion/unet.py and controlnet.py so that a diffusers
``unet/diffusion_pytorch_model.safetensors`` loads as-is, like the reference's strict
``load_state_dict`` torch2coreml.py:917-918) and a fast random-init generator for benchmarks
(no real weights exist offline)."""
from collections import OrderedDict

import numpy as np

from .hip_model import CROSS_ATTN_DOWN, CROSS_ATTN_UP, normalize_unet_config


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
            if t == CROSS_ATTN_DOWN:
                _transformer(sh, f"down_blocks.{i}.attentions.{j}", out, ctx,
                             cfg["transformer_layers_per_block"][i])
        if i != len(boc) - 1:
            _conv(sh, f"down_blocks.{i}.downsamplers.0.conv", out, out, 3)
    c = boc[-1]
    _resnet(sh, "mid_block.resnets.0", c, c, temb)
    _transformer(sh, "mid_block.attentions.0", c, ctx, cfg["transformer_layers_per_block"][-1])
    _resnet(sh, "mid_block.resnets.1", c, c, temb)


def unet_param_shapes(cfg):
    cfg = normalize_unet_config(cfg)
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
            if t == CROSS_ATTN_UP:
                _transformer(sh, f"up_blocks.{i}.attentions.{j}", out, ctx, rev_depth[i])
        if i != n - 1:
            _conv(sh, f"up_blocks.{i}.upsamplers.0.conv", out, out, 3)
    _norm(sh, "conv_norm_out", boc[0])
    _conv(sh, "conv_out", boc[0], cfg["out_channels"], 3)
    return sh


def controlnet_param_shapes(cfg, cond_channels=(16, 32, 96, 256)):
    cfg = normalize_unet_config(cfg)
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


def vae_decoder_param_shapes(cfg):
    """AutoencoderKL decoder + post_quant_conv (diffusers >= 0.15 key names)."""
    sh = OrderedDict()
    boc = tuple(cfg["block_out_channels"])
    cz, top = cfg["latent_channels"], boc[-1]

    def resnet(p, cin, cout):
        _norm(sh, p + ".norm1", cin)
        _conv(sh, p + ".conv1", cin, cout, 3)
        _norm(sh, p + ".norm2", cout)
        _conv(sh, p + ".conv2", cout, cout, 3)
        if cin != cout:
            _conv(sh, p + ".conv_shortcut", cin, cout, 1)

    _conv(sh, "post_quant_conv", cz, cz, 1)
    _conv(sh, "decoder.conv_in", cz, top, 3)
    resnet("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    _norm(sh, a + ".group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        sh[f"{a}.{n}.weight"] = (top, top)
        sh[f"{a}.{n}.bias"] = (top,)
    resnet("decoder.mid_block.resnets.1", top, top)
    cin = top
    for i, cout in enumerate(reversed(boc)):
        for j in range(cfg["layers_per_block"] + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i != len(boc) - 1:
            _conv(sh, f"decoder.up_blocks.{i}.upsamplers.0.conv", cout, cout, 3)
        cin = cout
    _norm(sh, "decoder.conv_norm_out", boc[0])
    _conv(sh, "decoder.conv_out", boc[0], cfg["out_channels"], 3)
    return sh


# text towers of the BASELINE models (transformers CLIPTextModel config keys): SD2.1's OpenCLIP ViT-H (23 of 24 layers kept),
# SD1.5's CLIP ViT-L
TEXT_ENCODER_CONFIGS = {
    "stabilityai/stable-diffusion-2-1-base": dict(vocab_size=49408, hidden_size=1024, intermediate_size=4096, num_hidden_layers=23,
                                                  num_attention_heads=16, max_position_embeddings=77, hidden_act="gelu",
                                                  layer_norm_eps=1e-5, eos_token_id=2, architectures=["CLIPTextModel"]),
    "runwayml/stable-diffusion-v1-5": dict(vocab_size=49408, hidden_size=768, intermediate_size=3072, num_hidden_layers=12,
                                           num_attention_heads=12, max_position_embeddings=77, hidden_act="quick_gelu",
                                           layer_norm_eps=1e-5, eos_token_id=2, architectures=["CLIPTextModel"]),
}


def text_encoder_param_shapes(cfg):
    """transformers' CLIPTextModel[WithProjection] state-dict inventory (what HipTextEncoder loads)."""
    sh = OrderedDict()
    d, i = cfg["hidden_size"], cfg["intermediate_size"]
    sh["text_model.embeddings.token_embedding.weight"] = (cfg["vocab_size"], d)
    sh["text_model.embeddings.position_embedding.weight"] = (cfg.get("max_position_embeddings", 77), d)
    for layer in range(cfg["num_hidden_layers"]):
        p = f"text_model.encoder.layers.{layer}"
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            sh[f"{p}.self_attn.{n}.weight"] = (d, d)
            sh[f"{p}.self_attn.{n}.bias"] = (d,)
        for n in ("layer_norm1", "layer_norm2"):
            sh[f"{p}.{n}.weight"] = (d,)
            sh[f"{p}.{n}.bias"] = (d,)
        sh[f"{p}.mlp.fc1.weight"], sh[f"{p}.mlp.fc1.bias"] = (i, d), (i,)
        sh[f"{p}.mlp.fc2.weight"], sh[f"{p}.mlp.fc2.bias"] = (d, i), (d,)
    sh["text_model.final_layer_norm.weight"] = (d,)
    sh["text_model.final_layer_norm.bias"] = (d,)
    if cfg.get("projection_dim"):
        sh["text_projection.weight"] = (cfg["projection_dim"], d)
    return sh


def validate_checkpoint(tensors, shapes):
    """Strict key/shape check (Linear weights may be stored 2-D, unet.py:121-127)."""
    missing = [k for k in shapes if k not in tensors]
    if missing:
        raise KeyError(f"checkpoint is missing {len(missing)} tensors, e.g. {missing[:3]}")
    for k, s in shapes.items():
        got = tuple(tensors[k].shape)
        if got != tuple(s) and not (len(got) == 2 and tuple(s) == got + (1, 1)):
            raise ValueError(f"{k}: shape {got}, expected {tuple(s)}")


def random_checkpoint(shapes, seed=0, dtype=np.float16):
    """Random-init weights of the architecture: N(0, 1/fan_in) matrices, near-identity norms."""
    rng = np.random.default_rng(seed)
    out = {}
    for key, shape in shapes.items():
        is_norm = ".norm" in key or key.startswith("conv_norm_out")
        if len(shape) == 1:
            if is_norm and key.endswith(".weight"):
                w = 1.0 + 0.1 * rng.standard_normal(shape, dtype=np.float32)
            else:
                w = 0.05 * rng.standard_normal(shape, dtype=np.float32)
        else:
            fan_in = int(np.prod(shape[1:]))
            w = rng.standard_normal(shape, dtype=np.float32)
            w *= np.float32(1.0 / np.sqrt(fan_in))
        out[key] = w.astype(dtype)
    return out
