"""fp32 compute path of the VAE graphs: the reference converts an SDXL checkpoint's own VAE with
float32 inputs and FLOAT32 compute precision because its activations leave the fp16 range (torch2coreml.py:570-578 decoder,
:726-733 encoder), and the pipeline reads the dtype off the model (pipeline.py:315).  HipVaeDecoder / HipVaeEncoder(dtype=
np.float32) run the same graph on fp32 kernels.  Oracle: oracle/vae_ref.py (torch fp32; PARITY UNPINNED - diffusers absent).
Tolerance: fp32 vs fp32, max |err| <= 2e-4 * max|ref| (summation order only)."""
import numpy as np
import pytest
import torch

from oracle import psnr, vae_ref, weights
from python_hip_stable_diffusion import HipVaeDecoder, HipVaeEncoder

pytestmark = pytest.mark.gpu


def _close32(got, ref, what, rel=2e-4):
    assert got.shape == ref.shape and np.isfinite(got).all(), what
    err = float(np.abs(got.astype(np.float64) - ref).max())
    bound = rel * float(np.abs(ref).max()) + 1e-6
    assert err <= bound, f"{what}: max|err| {err:.3e} > {bound:.3e} (PSNR {psnr.compute_psnr(got, ref):.1f} dB)"


@pytest.mark.parametrize("name,hw", [("mini", 8), ("mini", 12), ("sd", 32)])
def test_vae_decoder_fp32_matches_oracle(name, hw):
    cfg = vae_ref.VAE_CONFIGS[name]
    sd16 = weights.make_state_dict(vae_ref.vae_decoder_param_shapes(cfg), seed=61, dtype=np.float16, gain=1.6)
    sd = weights.to_torch({k: v.astype(np.float32) for k, v in sd16.items()})
    vae = HipVaeDecoder(cfg, sd16, batch=1, latent_height=hw, latent_width=hw, dtype=np.float32)
    assert vae.expected_inputs["z"]["dtype"] == np.float32                       # what pipeline.py:315 reads
    z = (weights.seeded_normal((1, 4, hw, hw), 63) / 0.18215).astype(np.float32)
    out = vae(z=z)["image"]
    ref = vae_ref.vae_decode(sd, cfg, torch.from_numpy(z)).numpy()
    _close32(out, ref, f"fp32 VAE decoder {name} @{hw}")
    assert np.array_equal(out, vae(z=z)["image"])                                # graph replay is deterministic
    with pytest.raises(TypeError):
        vae(z=z.astype(np.float16))                                              # the declared dtype is enforced (coreml_model.py:104-108)
    vae.close()


def test_vae_decoder_fp32_survives_activations_beyond_the_fp16_range():
    """The reason the path exists: scale conv_in so that the trunk carries values ~1e6 (every GroupNorm renormalises them, so
    the image is ordinary).  The fp16-storage handle overflows to inf / NaN, the fp32 handle matches the oracle."""
    cfg = vae_ref.VAE_CONFIGS["mini"]
    sd16 = weights.make_state_dict(vae_ref.vae_decoder_param_shapes(cfg), seed=61, dtype=np.float16, gain=1.6)
    sd16 = dict(sd16)
    sd16["decoder.conv_in.weight"] = (sd16["decoder.conv_in.weight"].astype(np.float32) * 2.0e3).astype(np.float16)
    sd = weights.to_torch({k: v.astype(np.float32) for k, v in sd16.items()})
    hw = 8
    z = (weights.seeded_normal((1, 4, hw, hw), 64) * 200.0).astype(np.float32)
    ref = vae_ref.vae_decode(sd, cfg, torch.from_numpy(z)).numpy()
    assert np.isfinite(ref).all() and np.abs(ref).max() < 1e3
    f32 = HipVaeDecoder(cfg, sd16, batch=1, latent_height=hw, latent_width=hw, dtype=np.float32)
    out = f32(z=z)["image"]
    _close32(out, ref, "fp32 VAE decoder with a 1e6-scale trunk", rel=1e-3)
    f16 = HipVaeDecoder(cfg, sd16, batch=1, latent_height=hw, latent_width=hw)
    bad = f16(z=z.astype(np.float16))["image"]
    assert (not np.isfinite(bad).all()) or psnr.compute_psnr(bad, ref) < 20.0   # the fp16-storage path cannot represent the trunk
    f32.close(), f16.close()


@pytest.mark.parametrize("name,hw", [("mini", 64), ("mini", 40)])
def test_vae_encoder_fp32_matches_oracle(name, hw):
    cfg = vae_ref.VAE_CONFIGS[name]
    sd16 = weights.make_state_dict(vae_ref.vae_encoder_param_shapes(cfg), seed=71, dtype=np.float16, gain=1.4)
    sd = weights.to_torch({k: v.astype(np.float32) for k, v in sd16.items()})
    enc = HipVaeEncoder(cfg, sd16, batch=1, height=hw, width=hw, dtype=np.float32)
    x = np.tanh(weights.seeded_normal((1, 3, hw, hw), 72)).astype(np.float32)
    out = enc(x=x)["latent"]
    ref = vae_ref.vae_encode(sd, cfg, torch.from_numpy(x)).numpy()
    _close32(out, ref, f"fp32 VAE encoder {name} @{hw}")
    enc.close()


def test_vae_fp32_handle_keeps_fp32_weights():
    """An fp32 checkpoint (an SDXL VAE is shipped in fp32) must not be rounded to fp16 on upload: weights that are
    NOT fp16-representable, against the oracle on the same fp32 weights - within the fp32 tolerance, which fp16-rounded weights
    (2^-11 relative per weight) miss by an order of magnitude."""
    cfg = vae_ref.VAE_CONFIGS["mini"]
    sd32 = weights.make_state_dict(vae_ref.vae_decoder_param_shapes(cfg), seed=67, dtype=np.float32, gain=1.6)
    assert any(not np.array_equal(v, v.astype(np.float16).astype(np.float32)) for v in sd32.values())
    sd = weights.to_torch(sd32)
    hw = 8
    z = (weights.seeded_normal((1, 4, hw, hw), 68) / 0.18215).astype(np.float32)
    ref = vae_ref.vae_decode(sd, cfg, torch.from_numpy(z)).numpy()
    vae = HipVaeDecoder(cfg, sd32, batch=1, latent_height=hw, latent_width=hw, dtype=np.float32)
    _close32(vae(z=z)["image"], ref, "fp32 VAE decoder with fp32 (not fp16-representable) weights")
    vae.close()
    # the same checkpoint rounded to fp16 first is measurably further away: the test would notice a rounding upload
    sd16 = weights.to_torch({k: v.astype(np.float16).astype(np.float32) for k, v in sd32.items()})
    rounded = vae_ref.vae_decode(sd16, cfg, torch.from_numpy(z)).numpy()
    assert np.abs(rounded - ref).max() > 2e-4 * np.abs(ref).max()
