"""Round-4 pins against the REAL reference (build container only; needs ``/root/reference``).

  python -m oracle.pin_round4 --batch4     # full SD2.1-base forward at UNet batch 4 (BASELINE config 3 per GPU: two prompts)
  python -m oracle.pin_round4 --batch16    # ... at UNet batch 16 (eight prompts), evaluated as eight batch-2 reference calls
  python -m oracle.pin_round4 --vae        # VAE decoder assembled from the reference's OWN blocks vs oracle/vae_ref.py

Same protocol as ``pin_against_reference.py`` / ``pin_round2.py``: the reference's modules are imported from where they
lie, loaded with the deterministic synthetic checkpoint of ``oracle/weights.py`` and evaluated on torch-CPU fp32; only
seeds and output tensors are stored under ``tests/golden/`` (inputs are regenerated from the seeds on both machines).

VAE: diffusers' AutoencoderKL is absent offline, but its decoder is made of blocks the reference itself
defines and uses: ``unet.ResnetBlock2D(temb_channels=None, eps=1e-6)`` (unet.py:406-489), ``unet.Upsample2D``
(unet.py:492-500) and single-head ``attention.original`` (attention.py:147-168).  ``pin_vae`` wires those modules in the
decoder's TOPOLOGY (restated from the public architecture) and requires ``vae_ref.vae_decode`` to agree
to 1e-5 - arithmetic pinned by the reference's blocks, topology restated.  The q/k/v/out projections with bias and the
GroupNorm in front of the attention are torch.nn.functional calls in both.  The ENCODER's asymmetric-pad stride-2 conv has
no counterpart in the reference (its Downsample2D pads symmetrically, unet.py:503-510) and stays unpinned.
"""
import argparse
import os
import time

import numpy as np
import torch
import torch.nn.functional as F

from oracle import unet_ref, vae_ref, weights
from oracle.pin_against_reference import GOLDEN, _maxdiff, load_reference
from oracle.pin_round2 import _build_reference

# input seeds of the batch goldens (tests regenerate the same tensors): sample, encoder_hidden_states
BATCH_SEEDS = {4: (401, 402), 16: (1601, 1602)}


def batch_inputs(batch, hw=64, ctx=1024):
    """fp16-representable inputs of the batch goldens; timesteps differ per CFG pair like two prompts mid-schedule would not
    (one schedule), so all rows share the timestep - but every sample / prompt row is distinct."""
    s_seed, e_seed = BATCH_SEEDS[batch]
    sample = weights.seeded_normal((batch, 4, hw, hw), s_seed).astype(np.float16)
    ehs = weights.seeded_normal((batch, ctx, 1, 77), e_seed).astype(np.float16)
    ts = np.full((batch,), 601.0, np.float32)
    return sample, ts, ehs


def pin_batch(unet, report, batch):
    cfg, sd, model = _build_reference(unet, "sd21-base", 0)
    sample, ts, ehs = batch_inputs(batch)
    outs = {}
    t0 = time.time()
    impls = ("ORIGINAL", "SPLIT_EINSUM_V2") if batch == 4 else ("ORIGINAL",)
    for impl in impls:
        unet.ATTENTION_IMPLEMENTATION_IN_EFFECT = getattr(unet.AttentionImplementations, impl)
        rows = []
        for i in range(0, batch, 2 if batch == 16 else batch):      # batch 16: eight independent batch-2 reference calls
            n = 2 if batch == 16 else batch
            rows.append(model(torch.from_numpy(sample[i:i + n].astype(np.float32)), torch.from_numpy(ts[i:i + n]),
                              torch.from_numpy(ehs[i:i + n].astype(np.float32)))[0].numpy())
        outs[impl] = np.concatenate(rows)
    dt = time.time() - t0
    ref = outs["ORIGINAL"]
    line = f"unet sd21-base @64x64 batch {batch}: {dt:.0f} s on {torch.get_num_threads()} threads, max|y| {np.abs(ref).max():.3f}"
    if "SPLIT_EINSUM_V2" in outs:
        d2 = _maxdiff(ref, outs["SPLIT_EINSUM_V2"])
        assert d2 < 1e-4, d2
        line += f", ORIGINAL vs SPLIT_EINSUM_V2 {d2:.2e}"
    if batch == 4:   # the oracle restatement on the same inputs
        mine = unet_ref.unet_forward(sd, cfg, torch.from_numpy(sample.astype(np.float32)), torch.from_numpy(ts),
                                     torch.from_numpy(ehs.astype(np.float32))).numpy()
        d = _maxdiff(mine, ref)
        assert d < 2e-5 * max(1.0, np.abs(ref).max()), d
        line += f", max|oracle-ref| {d:.2e}"
    report.append(line)
    s_seed, e_seed = BATCH_SEEDS[batch]
    np.savez_compressed(os.path.join(GOLDEN, f"unet_sd21-base_b{batch}_golden.npz"), seed=np.array(0), batch=np.array(batch),
                        sample_seed=np.array(s_seed), ehs_seed=np.array(e_seed), timestep=ts,
                        noise_pred=ref.astype(np.float32))


class _RefVaeDecoder(torch.nn.Module):
    """AutoencoderKL decoder + post_quant_conv wired from the reference's own blocks (module names = diffusers key names,
    so the synthetic checkpoint of vae_ref.vae_decoder_param_shapes loads with strict=True)."""

    def __init__(self, unet, att, cfg):
        super().__init__()
        self.att = att
        boc = cfg["block_out_channels"]
        cz, top = cfg["latent_channels"], boc[-1]
        nn = torch.nn

        def res(cin, cout):
            return unet.ResnetBlock2D(in_channels=cin, out_channels=cout, temb_channels=None, groups=32, eps=1e-6)

        self.post_quant_conv = nn.Conv2d(cz, cz, 1)
        dec = nn.Module()
        dec.conv_in = nn.Conv2d(cz, top, 3, padding=1)
        mid = nn.Module()
        mid.resnets = nn.ModuleList([res(top, top), res(top, top)])
        a = nn.Module()
        a.group_norm = nn.GroupNorm(32, top, eps=1e-6)
        a.to_q, a.to_k, a.to_v = nn.Linear(top, top), nn.Linear(top, top), nn.Linear(top, top)
        a.to_out = nn.ModuleList([nn.Linear(top, top)])
        mid.attentions = nn.ModuleList([a])
        dec.mid_block = mid
        ups = []
        cin = top
        for i, cout in enumerate(reversed(boc)):
            u = nn.Module()
            u.resnets = nn.ModuleList([res(cin if j == 0 else cout, cout) for j in range(cfg["layers_per_block"] + 1)])
            if i != len(boc) - 1:
                u.upsamplers = nn.ModuleList([unet.Upsample2D(cout)])
            ups.append(u)
            cin = cout
        dec.up_blocks = nn.ModuleList(ups)
        dec.conv_norm_out = nn.GroupNorm(32, boc[0], eps=1e-6)
        dec.conv_out = nn.Conv2d(boc[0], cfg["out_channels"], 3, padding=1)
        self.decoder = dec

    def forward(self, z):
        d = self.decoder
        x = d.conv_in(self.post_quant_conv(z))
        x = d.mid_block.resnets[0](x, None)
        a = d.mid_block.attentions[0]
        b, c, h, w = x.shape
        t = a.group_norm(x).reshape(b, c, h * w).transpose(1, 2)              # (B, S, C)
        # the reference's attention takes (B, C, 1, S) tensors (attention.py:147-168); one head of dim C
        to_bc1s = lambda y: y.transpose(1, 2).reshape(b, c, 1, h * w)
        o = self.att.original(to_bc1s(a.to_q(t)), to_bc1s(a.to_k(t)), to_bc1s(a.to_v(t)), None, 1, c)   # (B, C, 1, S)
        o = a.to_out[0](o.reshape(b, c, h * w).transpose(1, 2))
        x = x + o.transpose(1, 2).reshape(b, c, h, w)
        x = d.mid_block.resnets[1](x, None)
        for u in d.up_blocks:
            for r in u.resnets:
                x = r(x, None)
            if hasattr(u, "upsamplers"):
                x = u.upsamplers[0](x)
        return d.conv_out(F.silu(d.conv_norm_out(x)))


def pin_vae(att, unet, report):
    for name, seed, hw in (("mini", 71, 16), ("sd", 72, 32)):
        cfg = vae_ref.VAE_CONFIGS[name]
        shapes = vae_ref.vae_decoder_param_shapes(cfg)
        sd = weights.to_torch(weights.round_to_fp16(weights.make_state_dict(shapes, seed=seed)))
        model = _RefVaeDecoder(unet, att, cfg).eval()
        assert set(model.state_dict().keys()) == set(shapes.keys()), sorted(set(model.state_dict()) ^ set(shapes))[:8]
        model.load_state_dict({k: v.clone() for k, v in sd.items()})
        z = weights.seeded_normal((1, cfg["latent_channels"], hw, hw), seed + 1).astype(np.float16).astype(np.float32)
        t0 = time.time()
        ref = model(torch.from_numpy(z)).numpy()
        mine = vae_ref.vae_decode(sd, cfg, torch.from_numpy(z)).numpy()
        d = _maxdiff(mine, ref)
        assert d <= 1e-5 * max(1.0, np.abs(ref).max()), (name, d)
        report.append(f"vae decoder {name} @{hw}x{hw} latents: reference blocks (ResnetBlock2D temb=None eps=1e-6, Upsample2D, "
                      f"attention.original heads=1) vs oracle/vae_ref.vae_decode: max|diff| {d:.2e} (max|y| {np.abs(ref).max():.3f}), "
                      f"{time.time() - t0:.0f} s")
        np.savez_compressed(os.path.join(GOLDEN, f"vae_decoder_{name}_golden.npz"), seed=np.array(seed), hw=np.array(hw),
                            z_seed=np.array(seed + 1), image=ref.astype(np.float32))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch4", action="store_true")
    ap.add_argument("--batch16", action="store_true")
    ap.add_argument("--vae", action="store_true")
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    att, _, unet, _ = load_reference()
    report = []
    if args.vae:
        pin_vae(att, unet, report)
    if args.batch4:
        pin_batch(unet, report, 4)
    if args.batch16:
        pin_batch(unet, report, 16)
    with open(os.path.join(GOLDEN, "PIN_REPORT_r04.txt"), "a") as f:
        f.write("\n".join(report) + "\n")
    print("\n".join(report))


if __name__ == "__main__":
    main()
