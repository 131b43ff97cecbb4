"""Oracle (test infrastructure): AutoencoderKL decoder / encoder (SD 1.x / 2.x VAE), torch-CPU fp32.

Third-party arithmetic: the reference wraps diffusers' ``AutoencoderKL`` as ``decoder(post_quant_conv(z))``
(python_coreml_stable_diffusion/torch2coreml.py:584-594) and calls it from pipeline.py:313-320; diffusers is not available
offline and the reference itself checks the decoder only by a live PSNR >= 35 dB against diffusers (torch2coreml.py:631-639).
Status (round 4):
  * DECODER - **arithmetic pinned, topology restated**.  Its blocks ARE the reference's: ``unet.ResnetBlock2D(temb_channels=
    None, eps=1e-6)`` (unet.py:406-489), ``unet.Upsample2D`` (unet.py:492-500) and single-head ``attention.original``
    (attention.py:147-168).  ``oracle/pin_round4.py --vae`` wires those modules in the topology below and ``vae_decode`` agrees
    with them to <= 1.6e-7 (tests/golden/vae_decoder_*_golden.npz, tests/test_oracle.py); the topology itself is restated from
    the public architecture:
      post_quant_conv 1x1 -> conv_in 3x3 -> mid [ResNet, 1-head self-attention (group_norm 32, eps 1e-6, Linear q/k/v/out with
      bias, residual), ResNet] -> up blocks (layers_per_block+1 ResNets, nearest-x2 + conv3x3 after all but the last; ResNet
      eps 1e-6, no time embedding, conv_shortcut 1x1 when channels change) -> GroupNorm(32, 1e-6) -> SiLU -> conv_out 3x3.
  * ENCODER - the same blocks, but its asymmetric-pad stride-2 downsample (F.pad (0, 1, 0, 1), diffusers Downsample2D with
    padding=0) has no counterpart in the reference (unet.Downsample2D pads symmetrically, unet.py:503-510): **PARITY UNPINNED**.
Key names follow diffusers >= 0.15 (``decoder.mid_block.attentions.0.to_q`` ...).
"""
from collections import OrderedDict

import torch
import torch.nn.functional as F

VAE_CONFIGS = {
    "sd": dict(latent_channels=4, out_channels=3, block_out_channels=(128, 256, 512, 512), layers_per_block=2),
    "mini": dict(latent_channels=4, out_channels=3, block_out_channels=(64, 64, 128, 128), layers_per_block=1),
}


def vae_decoder_param_shapes(cfg):
    sh = OrderedDict()

    def conv(name, cin, cout, k):
        sh[name + ".weight"] = (cout, cin, k, k)
        sh[name + ".bias"] = (cout,)

    def norm(name, c):
        sh[name + ".weight"] = (c,)
        sh[name + ".bias"] = (c,)

    def resnet(p, cin, cout):
        norm(p + ".norm1", cin)
        conv(p + ".conv1", cin, cout, 3)
        norm(p + ".norm2", cout)
        conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".conv_shortcut", cin, cout, 1)

    boc = cfg["block_out_channels"]
    cz, top = cfg["latent_channels"], boc[-1]
    conv("post_quant_conv", cz, cz, 1)
    conv("decoder.conv_in", cz, top, 3)
    resnet("decoder.mid_block.resnets.0", top, top)
    a = "decoder.mid_block.attentions.0"
    norm(a + ".group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        sh[f"{a}.{n}.weight"] = (top, top)
        sh[f"{a}.{n}.bias"] = (top,)
    resnet("decoder.mid_block.resnets.1", top, top)
    cin = top
    for i, cout in enumerate(reversed(boc)):
        for j in range(cfg["layers_per_block"] + 1):
            resnet(f"decoder.up_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i != len(boc) - 1:
            conv(f"decoder.up_blocks.{i}.upsamplers.0.conv", cout, cout, 3)
        cin = cout
    norm("decoder.conv_norm_out", boc[0])
    conv("decoder.conv_out", boc[0], cfg["out_channels"], 3)
    return sh


def _conv(sd, name, x, padding=0):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], padding=padding)


def _gn(sd, name, x):
    return F.group_norm(x, 32, sd[name + ".weight"], sd[name + ".bias"], 1e-6)


def _resnet(sd, p, x):
    h = _conv(sd, p + ".conv1", F.silu(_gn(sd, p + ".norm1", x)), 1)
    h = _conv(sd, p + ".conv2", F.silu(_gn(sd, p + ".norm2", h)), 1)
    if p + ".conv_shortcut.weight" in sd:
        x = _conv(sd, p + ".conv_shortcut", x)
    return x + h


@torch.no_grad()
def vae_decode(sd, cfg, z):
    """z (B, 4, h, w) = latents / scaling_factor -> image (B, 3, 8h, 8w) in [-1, 1]."""
    x = _conv(sd, "post_quant_conv", z.float())
    x = _conv(sd, "decoder.conv_in", x, 1)
    x = _resnet(sd, "decoder.mid_block.resnets.0", x)
    a = "decoder.mid_block.attentions.0"
    b, c, h, w = x.shape
    t = _gn(sd, a + ".group_norm", x).reshape(b, c, h * w).transpose(1, 2)          # (B, S, C)
    q = F.linear(t, sd[a + ".to_q.weight"], sd[a + ".to_q.bias"])
    k = F.linear(t, sd[a + ".to_k.weight"], sd[a + ".to_k.bias"])
    v = F.linear(t, sd[a + ".to_v.weight"], sd[a + ".to_v.bias"])
    p = torch.softmax(q @ k.transpose(1, 2) * c ** -0.5, dim=-1)
    o = F.linear(p @ v, sd[a + ".to_out.0.weight"], sd[a + ".to_out.0.bias"])
    x = x + o.transpose(1, 2).reshape(b, c, h, w)
    x = _resnet(sd, "decoder.mid_block.resnets.1", x)
    n = len(cfg["block_out_channels"])
    for i in range(n):
        for j in range(cfg["layers_per_block"] + 1):
            x = _resnet(sd, f"decoder.up_blocks.{i}.resnets.{j}", x)
        if i != n - 1:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = _conv(sd, f"decoder.up_blocks.{i}.upsamplers.0.conv", x, 1)
    x = F.silu(_gn(sd, "decoder.conv_norm_out", x))
    return _conv(sd, "decoder.conv_out", x, 1)


def vae_encoder_param_shapes(cfg):
    """AutoencoderKL encoder + quant_conv (diffusers key names): conv_in, down_blocks.{i}.resnets.{j},
    downsamplers.0.conv, mid_block, conv_norm_out, conv_out (-> 2 * latent channels), quant_conv."""
    sh = OrderedDict()

    def conv(name, cin, cout, k):
        sh[name + ".weight"] = (cout, cin, k, k)
        sh[name + ".bias"] = (cout,)

    def norm(name, c):
        sh[name + ".weight"] = (c,)
        sh[name + ".bias"] = (c,)

    def resnet(p, cin, cout):
        norm(p + ".norm1", cin)
        conv(p + ".conv1", cin, cout, 3)
        norm(p + ".norm2", cout)
        conv(p + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(p + ".conv_shortcut", cin, cout, 1)

    boc = cfg["block_out_channels"]
    cz = cfg["latent_channels"]
    conv("encoder.conv_in", 3, boc[0], 3)
    cin = boc[0]
    for i, cout in enumerate(boc):
        for j in range(cfg["layers_per_block"]):
            resnet(f"encoder.down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout)
        if i != len(boc) - 1:
            conv(f"encoder.down_blocks.{i}.downsamplers.0.conv", cout, cout, 3)
        cin = cout
    top = boc[-1]
    resnet("encoder.mid_block.resnets.0", top, top)
    a = "encoder.mid_block.attentions.0"
    norm(a + ".group_norm", top)
    for n in ("to_q", "to_k", "to_v", "to_out.0"):
        sh[f"{a}.{n}.weight"] = (top, top)
        sh[f"{a}.{n}.bias"] = (top,)
    resnet("encoder.mid_block.resnets.1", top, top)
    norm("encoder.conv_norm_out", top)
    conv("encoder.conv_out", top, 2 * cz, 3)
    conv("quant_conv", 2 * cz, 2 * cz, 1)
    return sh


def _attention(sd, a, x):
    b, c, h, w = x.shape
    t = _gn(sd, a + ".group_norm", x).reshape(b, c, h * w).transpose(1, 2)
    q = F.linear(t, sd[a + ".to_q.weight"], sd[a + ".to_q.bias"])
    k = F.linear(t, sd[a + ".to_k.weight"], sd[a + ".to_k.bias"])
    v = F.linear(t, sd[a + ".to_v.weight"], sd[a + ".to_v.bias"])
    p = torch.softmax(q @ k.transpose(1, 2) * c ** -0.5, dim=-1)
    o = F.linear(p @ v, sd[a + ".to_out.0.weight"], sd[a + ".to_out.0.bias"])
    return x + o.transpose(1, 2).reshape(b, c, h, w)


@torch.no_grad()
def vae_encode(sd, cfg, x):
    """x (B, 3, H, W) in [-1, 1] -> moments quant_conv(encoder(x)) (B, 2*Cz, H/8, W/8); the wrapper of
    python_coreml_stable_diffusion/torch2coreml.py:739-749.  Downsample = F.pad(x, (0, 1, 0, 1)) + stride-2 conv
    (diffusers Downsample2D with padding=0).  PARITY UNPINNED (module docstring)."""
    x = _conv(sd, "encoder.conv_in", x.float(), 1)
    n = len(cfg["block_out_channels"])
    for i in range(n):
        for j in range(cfg["layers_per_block"]):
            x = _resnet(sd, f"encoder.down_blocks.{i}.resnets.{j}", x)
        if i != n - 1:
            p = f"encoder.down_blocks.{i}.downsamplers.0.conv"
            x = F.conv2d(F.pad(x, (0, 1, 0, 1)), sd[p + ".weight"], sd[p + ".bias"], stride=2)
    x = _resnet(sd, "encoder.mid_block.resnets.0", x)
    x = _attention(sd, "encoder.mid_block.attentions.0", x)
    x = _resnet(sd, "encoder.mid_block.resnets.1", x)
    x = F.silu(_gn(sd, "encoder.conv_norm_out", x))
    return _conv(sd, "quant_conv", _conv(sd, "encoder.conv_out", x, 1))
