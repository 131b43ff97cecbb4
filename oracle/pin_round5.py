"""Round-5 pins against the REAL reference (build container only; needs ``/root/reference``).  TEST INFRASTRUCTURE.

  python -m oracle.pin_round5 --vae-encoder   # VAE encoder assembled from the reference's OWN blocks vs oracle/vae_ref.vae_encode
  python -m oracle.pin_round5 --vae64         # VAE decoder (reference blocks) at the BENCHMARKED 64x64 latents
  python -m oracle.pin_round5 --batch16-v2    # the reference's SPLIT_EINSUM_V2 on the batch-16 inputs vs the stored ORIGINAL golden

Same protocol as ``pin_round4.py``: the reference's modules are imported from where they lie, loaded with the deterministic
synthetic checkpoint of ``oracle/weights.py`` (fp16-representable values), evaluated on torch-CPU fp32; only seeds and output
tensors go to ``tests/golden/``.

VAE encoder (wrapper at torch2coreml.py:739-749: ``quant_conv(encoder(x))``).  diffusers' AutoencoderKL is
absent offline, but every block of its encoder except the down-samplers is a module the reference itself defines and uses -
``unet.ResnetBlock2D(temb_channels=None, eps=1e-6)`` (unet.py:406-489) and single-head ``attention.original``
(attention.py:147-168); the down-sampler is two plain torch calls, ``F.pad(x, (0, 1, 0, 1))`` + ``F.conv2d(stride=2)``
(diffusers ``Downsample2D(padding=0)``; the reference's own Downsample2D pads symmetrically, unet.py:503-510, so it is NOT
used here).  ``pin_vae_encoder`` wires them in the encoder's public topology and requires ``vae_ref.vae_encode`` to agree to
1e-5: arithmetic pinned by the reference's blocks and by torch, topology restated.
"""
import argparse
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from oracle import vae_ref, weights
from oracle.pin_against_reference import GOLDEN, _maxdiff, load_reference
from oracle.pin_round2 import _build_reference
from oracle.pin_round4 import _RefVaeDecoder, batch_inputs


class _RefVaeEncoder(torch.nn.Module):
    """AutoencoderKL encoder + quant_conv from the reference's blocks (module names = diffusers key names: the synthetic
    checkpoint of vae_ref.vae_encoder_param_shapes loads with strict=True)."""

    def __init__(self, unet, att, cfg):
        super().__init__()
        self.att = att
        boc = cfg["block_out_channels"]
        cz, top = cfg["latent_channels"], boc[-1]
        nn = torch.nn

        def res(cin, cout):
            return unet.ResnetBlock2D(in_channels=cin, out_channels=cout, temb_channels=None, groups=32, eps=1e-6)

        enc = nn.Module()
        enc.conv_in = nn.Conv2d(3, boc[0], 3, padding=1)
        downs = []
        cin = boc[0]
        for i, cout in enumerate(boc):
            d = nn.Module()
            d.resnets = nn.ModuleList([res(cin if j == 0 else cout, cout) for j in range(cfg["layers_per_block"])])
            if i != len(boc) - 1:
                ds = nn.Module()
                ds.conv = nn.Conv2d(cout, cout, 3, stride=2, padding=0)   # called through F.conv2d behind the asymmetric pad
                d.downsamplers = nn.ModuleList([ds])
            downs.append(d)
            cin = cout
        enc.down_blocks = nn.ModuleList(downs)
        mid = nn.Module()
        mid.resnets = nn.ModuleList([res(top, top), res(top, top)])
        a = nn.Module()
        a.group_norm = nn.GroupNorm(32, top, eps=1e-6)
        a.to_q, a.to_k, a.to_v = nn.Linear(top, top), nn.Linear(top, top), nn.Linear(top, top)
        a.to_out = nn.ModuleList([nn.Linear(top, top)])
        mid.attentions = nn.ModuleList([a])
        enc.mid_block = mid
        enc.conv_norm_out = nn.GroupNorm(32, top, eps=1e-6)
        enc.conv_out = nn.Conv2d(top, 2 * cz, 3, padding=1)
        self.encoder = enc
        self.quant_conv = nn.Conv2d(2 * cz, 2 * cz, 1)

    def forward(self, x):
        e = self.encoder
        x = e.conv_in(x)
        for d in e.down_blocks:
            for r in d.resnets:
                x = r(x, None)
            if hasattr(d, "downsamplers"):
                c = d.downsamplers[0].conv
                x = F.conv2d(F.pad(x, (0, 1, 0, 1)), c.weight, c.bias, stride=2)
        x = e.mid_block.resnets[0](x, None)
        a = e.mid_block.attentions[0]
        b, c, h, w = x.shape
        t = a.group_norm(x).reshape(b, c, h * w).transpose(1, 2)
        to_bc1s = lambda y: y.transpose(1, 2).reshape(b, c, 1, h * w)
        o = self.att.original(to_bc1s(a.to_q(t)), to_bc1s(a.to_k(t)), to_bc1s(a.to_v(t)), None, 1, c)
        o = a.to_out[0](o.reshape(b, c, h * w).transpose(1, 2))
        x = x + o.transpose(1, 2).reshape(b, c, h, w)
        x = e.mid_block.resnets[1](x, None)
        return self.quant_conv(e.conv_out(F.silu(e.conv_norm_out(x))))


ENC_CASES = (("mini", 71, 64), ("sd", 71, 128))   # (config, checkpoint seed, image size): the shapes tests/test_round2_gpu.py uses


def encoder_image(hw, seed=72):
    """an "image" in [-1, 1], fp16-representable (the GPU tests regenerate it from the seed)"""
    return np.tanh(weights.seeded_normal((1, 3, hw, hw), seed)).astype(np.float16)


def pin_vae_encoder(att, unet, report):
    for name, seed, hw in ENC_CASES:
        cfg = vae_ref.VAE_CONFIGS[name]
        shapes = vae_ref.vae_encoder_param_shapes(cfg)
        sd16 = weights.make_state_dict(shapes, seed=seed, dtype=np.float16, gain=1.4)
        sd = weights.to_torch({k: v.astype(np.float32) for k, v in sd16.items()})
        model = _RefVaeEncoder(unet, att, cfg).eval()
        assert set(model.state_dict().keys()) == set(shapes.keys()), sorted(set(model.state_dict()) ^ set(shapes))[:8]
        model.load_state_dict({k: v.clone() for k, v in sd.items()})
        x = encoder_image(hw).astype(np.float32)
        t0 = time.time()
        ref = model(torch.from_numpy(x)).numpy()
        mine = vae_ref.vae_encode(sd, cfg, torch.from_numpy(x)).numpy()
        d = _maxdiff(mine, ref)
        assert d <= 1e-5 * max(1.0, np.abs(ref).max()), (name, d)
        report.append(f"vae encoder {name} @{hw}x{hw} image: reference blocks (ResnetBlock2D temb=None eps=1e-6, attention.original "
                      f"heads=1) + F.pad((0,1,0,1)) / F.conv2d(stride=2) vs oracle/vae_ref.vae_encode: max|diff| {d:.2e} "
                      f"(max|y| {np.abs(ref).max():.3f}), {time.time() - t0:.0f} s")
        np.savez_compressed(os.path.join(GOLDEN, f"vae_encoder_{name}_golden.npz"), seed=np.array(seed), hw=np.array(hw),
                            x_seed=np.array(72), gain=np.array(1.4), moments=ref.astype(np.float32))


def pin_vae64(att, unet, report):
    """The decode bench.py times: SD-sized decoder at 64x64 latents (512x512 image).  Stored as fp16 (2.4 MB): the values are
    images in about [-0.2, 0.2] with random weights, fp16 keeps 11 bits of them - 30 dB below the 60-dB gate of the test."""
    name, seed, hw = "sd", 72, 64
    cfg = vae_ref.VAE_CONFIGS[name]
    shapes = vae_ref.vae_decoder_param_shapes(cfg)
    sd = weights.to_torch(weights.round_to_fp16(weights.make_state_dict(shapes, seed=seed)))
    model = _RefVaeDecoder(unet, att, cfg).eval()
    model.load_state_dict({k: v.clone() for k, v in sd.items()})
    z = weights.seeded_normal((1, cfg["latent_channels"], hw, hw), seed + 3).astype(np.float16).astype(np.float32)
    t0 = time.time()
    ref = model(torch.from_numpy(z)).numpy()
    mine = vae_ref.vae_decode(sd, cfg, torch.from_numpy(z)).numpy()
    d = _maxdiff(mine, ref)
    assert d <= 1e-5 * max(1.0, np.abs(ref).max()), d
    report.append(f"vae decoder sd @64x64 latents (the benchmarked decode): reference blocks vs oracle/vae_ref.vae_decode: max|diff| "
                  f"{d:.2e} (max|y| {np.abs(ref).max():.3f}), {time.time() - t0:.0f} s")
    np.savez_compressed(os.path.join(GOLDEN, "vae_decoder_sd64_golden.npz"), seed=np.array(seed), hw=np.array(hw),
                        z_seed=np.array(seed + 3), image=ref.astype(np.float16))


def pin_batch16_v2(unet, report):
    """config 3 names SPLIT_EINSUM_V2: the reference's V2 path on the batch-16 inputs against the stored (ORIGINAL) golden."""
    g = np.load(os.path.join(GOLDEN, "unet_sd21-base_b16_golden.npz"))
    cfg, sd, model = _build_reference(unet, "sd21-base", 0)
    sample, ts, ehs = batch_inputs(16)
    unet.ATTENTION_IMPLEMENTATION_IN_EFFECT = unet.AttentionImplementations.SPLIT_EINSUM_V2
    t0 = time.time()
    rows = [model(torch.from_numpy(sample[i:i + 2].astype(np.float32)), torch.from_numpy(ts[i:i + 2]),
                  torch.from_numpy(ehs[i:i + 2].astype(np.float32)))[0].numpy() for i in range(0, 16, 2)]
    unet.ATTENTION_IMPLEMENTATION_IN_EFFECT = unet.AttentionImplementations.ORIGINAL
    v2 = np.concatenate(rows)
    d = _maxdiff(v2, g["noise_pred"])
    assert d < 1e-4, d
    report.append(f"unet sd21-base @64x64 batch 16, SPLIT_EINSUM_V2 (eight batch-2 reference calls, {time.time() - t0:.0f} s): "
                  f"max|V2 - stored ORIGINAL golden| {d:.2e} -> the batch-16 golden serves both modes")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vae-encoder", action="store_true")
    ap.add_argument("--vae64", action="store_true")
    ap.add_argument("--batch16-v2", action="store_true")
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    att, _, unet, _ = load_reference()
    report = []
    if args.vae_encoder:
        pin_vae_encoder(att, unet, report)
    if args.vae64:
        pin_vae64(att, unet, report)
    if args.batch16_v2:
        pin_batch16_v2(unet, report)
    with open(os.path.join(GOLDEN, "PIN_REPORT_r05.txt"), "a") as f:
        f.write("\n".join(report) + "\n")
    print("\n".join(report))


if __name__ == "__main__":
    main()
