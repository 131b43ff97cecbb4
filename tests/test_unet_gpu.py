"""GPU parity tests of the whole UNet / ControlNet graph through the CoreMLModel-shaped boundary
(HipModel -> C ABI -> HIP graph), against
  (a) the golden outputs of the REAL reference modules (tests/golden/unet_*_golden.npz, written
      by oracle/pin_against_reference.py from /root/reference), and
  (b) the oracle run live on the same seeded inputs.
Tolerance: fp16 HIP vs fp32 reference, PSNR (torch2coreml.py:59-74) >= 60 dB on a single forward
(the reference's own floor is 35 dB, torch2coreml.py:77); 20-step final latents >= 35 dB.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import psnr, scheduler_ref, unet_ref, weights
from python_hip_stable_diffusion import HipModel, schedulers

pytestmark = pytest.mark.gpu
IMPLS = ["ORIGINAL", "SPLIT_EINSUM", "SPLIT_EINSUM_V2"]
# SD1.5 ControlNet residuals at full size: gate = measured - 6 dB (round 4, gpurun_out/psnr_r4e.tsv: the 13 residuals measured
# 81.7 / 76.3 / 74.3 / 74.4 / 74.1 / 71.9 / 71.3 / 70.7 / 70.7 / 69.7 / 72.5 / 68.6 / 68.6 dB; the minimum sets the gate; was 50)
CONTROLNET_GATE_DB = 62.5


def synthetic_checkpoint(shapes, seed):
    return weights.make_state_dict(shapes, seed=seed, dtype=np.float16)


def golden_inputs(g, model):
    kw = dict(sample=g["sample"].astype(np.float16), timestep=g["timestep"].astype(np.float16),
              encoder_hidden_states=g["encoder_hidden_states"].astype(np.float16))
    for k in ("time_ids", "text_embeds"):
        if k in g:
            kw[k] = g[k].astype(np.float16)
    for k in model.expected_inputs:
        if k.startswith("additional_residual_"):
            kw[k] = g[k].astype(np.float16)
    return kw


@pytest.mark.parametrize("name", ["tiny", "mini", "mini-xl", "mini-control"])
def test_unet_matches_reference_golden_all_attention_modes(name):
    g = load_golden(f"unet_{name}_golden.npz")
    cfg = unet_ref.CONFIGS[name]
    sd = synthetic_checkpoint(unet_ref.unet_param_shapes(cfg), int(g["seed"]))
    model = HipModel(cfg, sd, batch=2, attention_implementation="ORIGINAL")
    kw = golden_inputs(g, model)
    outs = {}
    for impl in IMPLS:
        model.set_attention_implementation(impl)
        y = model(**kw)["noise_pred"]
        assert y.dtype == np.float32 and y.shape == g["noise_pred"].shape
        p = psnr.compute_psnr(y, g["noise_pred"])
        assert p >= 60.0, f"{name} {impl}: PSNR {p:.1f} dB vs reference golden"
        outs[impl] = y
    for impl in IMPLS[1:]:
        assert psnr.compute_psnr(outs[impl], outs["ORIGINAL"]) >= 60.0
    # replaying the captured HIP graph is deterministic
    model.set_attention_implementation("ORIGINAL")
    assert np.array_equal(model(**kw)["noise_pred"], outs["ORIGINAL"])
    model.close()


def test_graph_replay_equals_eager_launches():
    g = load_golden("unet_mini_golden.npz")
    cfg = unet_ref.CONFIGS["mini"]
    sd = synthetic_checkpoint(unet_ref.unet_param_shapes(cfg), int(g["seed"]))
    a = HipModel(cfg, sd, batch=2, use_graph=True)
    b = HipModel(cfg, sd, batch=2, use_graph=False)
    kw = golden_inputs(g, a)
    ya, yb = a(**kw)["noise_pred"], b(**kw)["noise_pred"]
    assert np.array_equal(ya, yb)
    # changing the prompt embedding re-projects the hoisted cross-attention K/V
    kw2 = dict(kw, encoder_hidden_states=(kw["encoder_hidden_states"] * np.float16(0.5)))
    y2 = a(**kw2)["noise_pred"]
    assert not np.array_equal(y2, ya)
    assert np.array_equal(a(**kw)["noise_pred"], ya)
    a.close(), b.close()


def test_boundary_validation_matches_coreml_model():
    """coreml_model.py:97-116: TypeError for wrong ndarray/dtype/shape, ValueError for unknown kwarg."""
    cfg = unet_ref.CONFIGS["tiny"]
    model = HipModel(cfg, synthetic_checkpoint(unet_ref.unet_param_shapes(cfg), 11), batch=2)
    g = load_golden("unet_tiny_golden.npz")
    kw = golden_inputs(g, model)
    assert set(model.expected_inputs) == {"sample", "timestep", "encoder_hidden_states"}
    assert model.expected_inputs["sample"]["shape"] == (2, 4, 8, 8)
    with pytest.raises(TypeError):
        model(**dict(kw, sample=kw["sample"].astype(np.float32)))
    with pytest.raises(TypeError):
        model(**dict(kw, sample=kw["sample"][:1]))
    with pytest.raises(TypeError):
        model(**dict(kw, timestep=[981, 981]))
    with pytest.raises(ValueError):
        model(**dict(kw, bogus=kw["sample"]))
    model.close()
    with pytest.raises(KeyError if False else FileNotFoundError):     # missing checkpoint tensor
        HipModel(cfg, {"conv_in.weight": np.zeros((32, 4, 3, 3), np.float16)}, batch=2)


def test_controlnet_matches_reference_golden():
    g = load_golden("controlnet_mini_golden.npz")
    cfg = unet_ref.CONFIGS["mini-control"]
    sd = synthetic_checkpoint(unet_ref.controlnet_param_shapes(cfg), int(g["seed"]))
    model = HipModel(cfg, sd, kind="controlnet", batch=2, attention_implementation="SPLIT_EINSUM")
    assert model.num_residuals == 13
    out = model(sample=g["sample"], timestep=g["timestep"].astype(np.float16),
                encoder_hidden_states=g["encoder_hidden_states"], controlnet_cond=g["controlnet_cond"])
    for i in range(13):
        ref = g[f"additional_residual_{i}"]
        p = psnr.compute_psnr(out[f"additional_residual_{i}"], ref)
        assert p >= 55.0, f"residual {i}: PSNR {p:.1f} dB"
    model.close()


def test_unet_batch_four_matches_two_batch_two_calls():
    """Config 3 shape: two prompts per GPU -> UNet batch 4; samples are independent."""
    g = load_golden("unet_mini_golden.npz")
    cfg = unet_ref.CONFIGS["mini"]
    sd = synthetic_checkpoint(unet_ref.unet_param_shapes(cfg), int(g["seed"]))
    m2 = HipModel(cfg, sd, batch=2, attention_implementation="SPLIT_EINSUM_V2")
    m4 = HipModel(cfg, sd, batch=4, attention_implementation="SPLIT_EINSUM_V2")
    kw = golden_inputs(g, m2)
    y2 = m2(**kw)["noise_pred"]
    kw4 = {k: np.concatenate([v, v[::-1]]) for k, v in kw.items()}
    y4 = m4(**kw4)["noise_pred"]
    assert psnr.compute_psnr(y4[:2], y2) >= 70 and psnr.compute_psnr(y4[2:], y2[::-1]) >= 70
    m2.close(), m4.close()


def test_device_resident_denoise_loop_matches_oracle_loop():
    """pipeline.py:500-573: 6 DDIM steps, CFG 7.5, latents from the numpy legacy stream; the HIP
    loop keeps latents in HBM, the oracle loop is the host restatement around the oracle UNet."""
    name = "mini"
    cfg = unet_ref.CONFIGS[name]
    shapes = unet_ref.unet_param_shapes(cfg)
    sd16 = synthetic_checkpoint(shapes, 21)
    sd = weights.to_torch({k: v.astype(np.float32) for k, v in sd16.items()})
    model = HipModel(cfg, sd16, batch=2, attention_implementation="SPLIT_EINSUM")
    hw = cfg["sample_size"]
    np.random.seed(93)                                                    # pipeline.py:726, :800
    latents0 = np.random.randn(1, 4, hw, hw).astype(np.float16)           # pipeline.py:331
    ehs = weights.seeded_normal((2, cfg["cross_attention_dim"], 1, 77), 94).astype(np.float16)

    def oracle_unet(x, ts, e):
        return unet_ref.unet_forward(sd, cfg, torch.from_numpy(x.astype(np.float32)),
                                     torch.from_numpy(ts.astype(np.float32)),
                                     torch.from_numpy(e.astype(np.float32))).numpy()

    n_steps, g = 6, 7.5
    ref = scheduler_ref.denoise_loop(oracle_unet, scheduler_ref.DDIM(), latents0.astype(np.float32), ehs, n_steps, g)
    sch = schedulers.DDIMScheduler()
    sch.set_timesteps(n_steps)
    ts, coef, hist = sch.device_tables()
    lat, ms = model.denoise_loop(latents0.astype(np.float32) * sch.init_noise_sigma, ts, coef, g, history=hist,
                                 encoder_hidden_states=ehs)
    assert lat.shape == ref.shape and len(ms) == n_steps and (ms > 0).all()
    p = psnr.compute_psnr(lat, ref)
    assert p >= 51.0, f"final latents PSNR {p:.1f} dB"   # measured 57.7 (r3): gate = measured - 6
    # and the host-stepped path through the boundary gives the same trajectory
    host = scheduler_ref.denoise_loop(lambda x, t, e: model(sample=x, timestep=t, encoder_hidden_states=e)["noise_pred"],
                                      scheduler_ref.DDIM(), latents0.astype(np.float32), ehs, n_steps, g)
    assert psnr.compute_psnr(lat, host) >= 60.0
    model.close()


def test_full_sd21_base_matches_reference_golden():
    """BASELINE config 2 at full size: SD2.1-base, 64x64 latents, CFG batch 2 (865.9 M params)."""
    g = load_golden("unet_sd21-base_golden.npz")
    cfg = unet_ref.CONFIGS["sd21-base"]
    shapes = unet_ref.unet_param_shapes(cfg)
    sd = synthetic_checkpoint(shapes, int(g["seed"]))
    model = HipModel("stabilityai/stable-diffusion-2-1-base", sd, batch=2, attention_implementation="ORIGINAL")
    del sd
    kw = golden_inputs(g, model)
    for impl in IMPLS:
        model.set_attention_implementation(impl)
        y = model(**kw)["noise_pred"]
        p = psnr.compute_psnr(y, g["noise_pred"])
        assert p >= 60.0, f"sd21-base {impl}: PSNR {p:.1f} dB vs reference golden"
    model.close()


@pytest.mark.parametrize("name,hw", [("mini", 8), ("sd", 16)])
def test_vae_decoder_matches_oracle(name, hw):
    """decoder(post_quant_conv(z)) (torch2coreml.py:584-594).  The oracle restates diffusers'
    AutoencoderKL from the public architecture (PARITY UNPINNED: no golden exists offline)."""
    from oracle import vae_ref
    from python_hip_stable_diffusion import HipVaeDecoder
    cfg = vae_ref.VAE_CONFIGS[name]
    shapes = vae_ref.vae_decoder_param_shapes(cfg)
    sd16 = weights.make_state_dict(shapes, seed=61, dtype=np.float16, gain=1.6)
    sd = weights.to_torch({k: v.astype(np.float32) for k, v in sd16.items()})
    vae = HipVaeDecoder(cfg, sd16, batch=1, latent_height=hw, latent_width=hw)
    z = (weights.seeded_normal((1, 4, hw, hw), 62) / 0.18215).astype(np.float16)     # pipeline.py:314
    out = vae(z=z)["image"]
    ref = vae_ref.vae_decode(sd, cfg, torch.from_numpy(z.astype(np.float32))).numpy()
    assert out.shape == ref.shape == (1, 3, hw * 8, hw * 8)
    p = psnr.compute_psnr(out, ref)
    assert p >= 68.0, f"VAE decoder {name}: PSNR {p:.1f} dB"   # measured 74.4 / 79.6 (r3)
    with pytest.raises(TypeError):
        vae(z=z.astype(np.float32))
    vae.close()


def test_full_sdxl_base_768_matches_reference_golden():
    """BASELINE config 4 (UNet part): SDXL-base, 96x96 latents (768x768), text_time add-embedding,
    transformer depth (1,2,10), 2.57 B parameters.  S_q = 9216 / 2304 are not multiples of 512,
    so SPLIT_EINSUM_V2 is rejected loudly instead of dropping the tail like attention.py:86."""
    g = load_golden("unet_sdxl-base_golden.npz")
    cfg = unet_ref.CONFIGS["sdxl-base"]
    sd = synthetic_checkpoint(unet_ref.unet_param_shapes(cfg), int(g["seed"]))
    model = HipModel("stabilityai/stable-diffusion-xl-base-1.0", sd, batch=2, latent_height=96, latent_width=96,
                     attention_implementation="ORIGINAL")
    del sd
    assert model.expected_inputs["time_ids"]["shape"] == (2, 6) and model.expected_inputs["text_embeds"]["shape"] == (2, 1280)
    kw = golden_inputs(g, model)
    for impl in ("ORIGINAL", "SPLIT_EINSUM"):
        model.set_attention_implementation(impl)
        y = model(**kw)["noise_pred"]
        p = psnr.compute_psnr(y, g["noise_pred"])
        assert p >= 60.0, f"sdxl-base {impl}: PSNR {p:.1f} dB vs reference golden"
    model.set_attention_implementation("SPLIT_EINSUM_V2")
    with pytest.raises(ValueError, match="512"):
        model(**kw)
    model.close()


def test_full_sd15_control_unet_and_controlnet_match_reference_golden():
    """BASELINE config 5: SD1.5 (8 heads -> head dims 40/80/160) control-UNet consuming 13 residuals,
    and the SD1.5 ControlNet producing them (controlnet.py:199-250)."""
    g = load_golden("unet_sd15-control_golden.npz")
    cfg = unet_ref.CONFIGS["sd15-control"]
    seed = int(g["seed"])
    sd = synthetic_checkpoint(unet_ref.unet_param_shapes(cfg), seed)
    model = HipModel(cfg, sd, batch=2, attention_implementation="SPLIT_EINSUM")
    del sd
    kw = dict(sample=g["sample"].astype(np.float16), timestep=g["timestep"].astype(np.float16),
              encoder_hidden_states=g["encoder_hidden_states"].astype(np.float16))
    for i, s_ in enumerate(unet_ref.residual_shapes(cfg, 2)):
        kw[f"additional_residual_{i}"] = (0.1 * weights.seeded_normal(s_, seed + 10 + i)).astype(np.float16)
    assert set(kw) == set(model.expected_inputs)
    y = model(**kw)["noise_pred"]
    p = psnr.compute_psnr(y, g["noise_pred"])
    assert p >= 60.0, f"sd15 control-UNet: PSNR {p:.1f} dB vs reference golden"
    model.close()

    gc = load_golden("controlnet_sd15_golden.npz")
    cn = HipModel(cfg, synthetic_checkpoint(unet_ref.controlnet_param_shapes(cfg), int(gc["seed"])), kind="controlnet",
                  batch=2, attention_implementation="ORIGINAL")
    out = cn(sample=weights.seeded_normal((2, 4, 64, 64), 72).astype(np.float16), timestep=np.array([981, 981], np.float16),
             encoder_hidden_states=weights.seeded_normal((2, 768, 1, 77), 73).astype(np.float16),
             controlnet_cond=np.random.RandomState(74).rand(2, 3, 512, 512).astype(np.float16))
    stride = int(gc["stride"])
    for i in range(13):   # the reference module's own outputs (a strided channel subset: the fixture stays small)
        ref = gc[f"additional_residual_{i}"].astype(np.float32)
        p = psnr.compute_psnr(out[f"additional_residual_{i}"][:, ::stride], ref)
        assert p >= CONTROLNET_GATE_DB, f"sd15 ControlNet residual {i}: PSNR {p:.1f} dB vs the reference golden"
    cn.close()
    # ... and the FULL tensors against the oracle run live on the same inputs (the oracle is pinned to the reference on this
    # very model by oracle/pin_against_reference.py --control, max |oracle - reference| <= 4e-6)
    sdc = weights.to_torch({k: v.astype(np.float32) for k, v in
                            synthetic_checkpoint(unet_ref.controlnet_param_shapes(cfg), int(gc["seed"])).items()})
    full = unet_ref.controlnet_forward(
        sdc, cfg, torch.from_numpy(weights.seeded_normal((2, 4, 64, 64), 72).astype(np.float16).astype(np.float32)),
        torch.tensor([981.0, 981.0]), torch.from_numpy(weights.seeded_normal((2, 768, 1, 77), 73).astype(np.float16).astype(np.float32)),
        torch.from_numpy(np.random.RandomState(74).rand(2, 3, 512, 512).astype(np.float16).astype(np.float32)))
    for i in range(13):
        ref = full[i].numpy()
        # (the golden stores the reference module's output rounded to fp16: half an fp16 ulp of the largest value)
        assert np.abs(ref[:, ::stride] - gc[f"additional_residual_{i}"]).max() <= 1e-3 * max(1.0, np.abs(ref).max())
        p = psnr.compute_psnr(out[f"additional_residual_{i}"], ref)
        assert p >= CONTROLNET_GATE_DB, f"sd15 ControlNet residual {i} (full tensor): PSNR {p:.1f} dB vs the oracle"

