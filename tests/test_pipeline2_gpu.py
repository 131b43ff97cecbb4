"""Round-2 end-to-end pipeline tests on the GPU: batched prompts / images, --unet-batch-one, ControlNet
residuals handed over on the device inside the loop, the SDXL base -> refiner swap, PNDM as the default
scheduler - each against the reference loop restated in oracle/ around the oracle networks."""
import numpy as np
import pytest
import torch

from oracle import psnr, scheduler_ref, unet_ref, weights
from python_hip_stable_diffusion import HipModel, schedulers
from python_hip_stable_diffusion.pipeline import HipStableDiffusionPipeline
from test_pipeline_gpu import StubTextEncoder, StubTokenizer

pytestmark = pytest.mark.gpu


def _ckpt(shapes, seed):
    sd16 = weights.make_state_dict(shapes, seed=seed, dtype=np.float16)
    return sd16, weights.to_torch({k: v.astype(np.float32) for k, v in sd16.items()})


def _oracle_unet(sd, cfg, **kw):
    def fn(x, ts, e):
        return unet_ref.unet_forward(sd, cfg, torch.from_numpy(x.astype(np.float32)), torch.from_numpy(ts.astype(np.float32)),
                                     torch.from_numpy(e.astype(np.float32)), **kw).numpy()
    return fn


def test_two_prompts_in_one_batched_call_match_two_single_calls():
    cfg = unet_ref.CONFIGS["mini"]
    sd16, _ = _ckpt(unet_ref.unet_param_shapes(cfg), 21)
    enc, tok = StubTextEncoder(cfg["cross_attention_dim"]), StubTokenizer4()
    one = HipStableDiffusionPipeline(enc, HipModel(cfg, sd16, batch=2), None, schedulers.PNDMScheduler(), tok,
                                     force_zeros_for_empty_prompt=False)
    two = HipStableDiffusionPipeline(enc, HipModel(cfg, sd16, batch=4), None, schedulers.PNDMScheduler(), tok,
                                     force_zeros_for_empty_prompt=False)
    prompts = ["a red cube", "a blue sphere on a table"]
    hw = cfg["sample_size"]
    lat = np.random.RandomState(5).randn(2, 4, hw, hw).astype(np.float16)
    both = two(prompts, num_inference_steps=4, latents=lat, output_type="latent", negative_prompt="blurry")
    assert both.images.shape == (2, 4, hw, hw) and len(both.step_ms) == 5           # PNDM: N + 1 evaluations, fused
    for i, p in enumerate(prompts):
        single = one(p, num_inference_steps=4, latents=lat[i:i + 1], output_type="latent", negative_prompt="blurry")
        assert psnr.compute_psnr(both.images[i:i + 1], single.images) >= 60.0
    # two images of one prompt (Swift imageCount): same embeddings, different latents
    imgs = two("a red cube", num_images_per_prompt=2, num_inference_steps=4, latents=lat, output_type="latent",
               negative_prompt="blurry")
    s0 = one("a red cube", num_inference_steps=4, latents=lat[:1], output_type="latent", negative_prompt="blurry")
    assert psnr.compute_psnr(imgs.images[:1], s0.images) >= 60.0
    assert not np.array_equal(imgs.images[0], imgs.images[1])
    one.unet.close(), two.unet.close()


class StubTokenizer4(StubTokenizer):
    """list-of-prompts aware variant of the stub tokenizer"""

    def __call__(self, text, **kw):
        if isinstance(text, list):
            ids = np.concatenate([StubTokenizer.__call__(self, t).input_ids for t in text])
            return type("Enc", (), {"input_ids": ids})()
        return StubTokenizer.__call__(self, text)


def test_unet_batch_one_evaluates_the_two_guidance_halves_sequentially():
    """pipeline.py:537-556 with a batch-1 handle; same result as the batched evaluation."""
    cfg = unet_ref.CONFIGS["mini"]
    sd16, _ = _ckpt(unet_ref.unet_param_shapes(cfg), 21)
    enc, tok = StubTextEncoder(cfg["cross_attention_dim"]), StubTokenizer()
    b2 = HipStableDiffusionPipeline(enc, HipModel(cfg, sd16, batch=2), None, schedulers.DDIMScheduler(), tok,
                                    force_zeros_for_empty_prompt=False)
    b1 = HipStableDiffusionPipeline(enc, HipModel(cfg, sd16, batch=1), None, schedulers.DDIMScheduler(), tok,
                                    force_zeros_for_empty_prompt=False)
    a = b2("a prompt", num_inference_steps=3, seed=7, output_type="latent")
    b = b1("a prompt", num_inference_steps=3, seed=7, output_type="latent", unet_batch_one=True)
    assert b.step_ms is None and a.step_ms is not None
    assert psnr.compute_psnr(b.images, a.images) >= 60.0
    with pytest.raises(ValueError, match="static batch"):
        b2("a prompt", num_inference_steps=3, unet_batch_one=True)
    b1.unet.close(), b2.unet.close()


def test_controlnet_residuals_stay_on_the_device_inside_the_loop():
    """BASELINE config 5's structure in miniature: two ControlNets (pipeline.py:269-282 sum) feeding the
    control-UNet (unet.py:1009-1022) every step of sd_unet_denoise_loop, against the reference loop restated
    around the oracle ControlNet + control-UNet, and against the host-stepped boundary path."""
    cfg = unet_ref.CONFIGS["mini-control"]
    usd16, usd = _ckpt(unet_ref.unet_param_shapes(cfg), 41)
    cn16, cnsd = zip(*[_ckpt(unet_ref.controlnet_param_shapes(cfg), s) for s in (51, 52)])
    hw = cfg["sample_size"]
    unet = HipModel(cfg, usd16, batch=2, attention_implementation="SPLIT_EINSUM")
    cns = [HipModel(cfg, c, kind="controlnet", batch=2, attention_implementation="SPLIT_EINSUM") for c in cn16]
    assert "additional_residual_0" in unet.expected_inputs
    pipe = HipStableDiffusionPipeline(StubTextEncoder(cfg["cross_attention_dim"]), unet, None, schedulers.DDIMScheduler(),
                                      StubTokenizer(), controlnet=cns, force_zeros_for_empty_prompt=False)
    conds = [np.random.RandomState(60 + i).rand(3, hw * 8, hw * 8) for i in range(2)]      # pipeline.py:717-721 range
    steps, gs, seed = 4, 7.5, 11
    fused = pipe("a prompt", num_inference_steps=steps, guidance_scale=gs, seed=seed, output_type="latent",
                 controlnet_cond=conds)
    assert fused.step_ms is not None and len(fused.step_ms) == steps                      # ran inside the device loop
    assert "additional_residual_0" not in unet.expected_inputs                            # residuals no longer cross the host
    host = pipe("a prompt", num_inference_steps=steps, guidance_scale=gs, seed=seed, output_type="latent",
                controlnet_cond=conds, device_loop=False)
    assert host.step_ms is None and "additional_residual_0" in unet.expected_inputs
    assert psnr.compute_psnr(fused.images, host.images) >= 50.0

    emb, _ = pipe._encode_prompt("a prompt", None, True, None, None)
    cond16 = [np.concatenate([np.stack([c])] * 2).astype(np.float16).astype(np.float32) for c in conds]

    def oracle_step(x, ts, e):
        xt, tt, et = (torch.from_numpy(v.astype(np.float32)) for v in (x, ts, e))
        total = None
        for sd_c, c in zip(cnsd, cond16):
            res = unet_ref.controlnet_forward(sd_c, cfg, xt, tt, et, torch.from_numpy(c))
            total = res if total is None else [a + b for a, b in zip(total, res)]
        total = [r.half().float() for r in total]                                          # fp16 hand-off (pipeline.py:284)
        return unet_ref.unet_forward(usd, cfg, xt, tt, et, additional_residuals=total).numpy()

    np.random.seed(seed)
    lat0 = np.random.randn(1, 4, hw, hw).astype(np.float16)
    want = scheduler_ref.denoise_loop(oracle_step, scheduler_ref.DDIM(), lat0.astype(np.float32), emb, steps, gs)
    p = psnr.compute_psnr(fused.images, want)
    assert p >= 52.0, f"device-resident ControlNet loop: PSNR {p:.1f} dB vs the oracle loop"   # measured 58.2 (r3)
    # one forward through the boundary with the ControlNets attached: no residual inputs needed
    unet.attach_controlnets(cns)
    x = np.concatenate([lat0, lat0]).astype(np.float16)
    t = np.array([981, 981], np.float16)
    for cn, c in zip(cns, cond16):
        cn.set_controlnet_cond(c.astype(np.float16))
    y = unet(sample=x, timestep=t, encoder_hidden_states=emb.astype(np.float16))["noise_pred"]
    assert psnr.compute_psnr(y, oracle_step(x, t, emb)) >= 55.0
    unet.attach_controlnets([])
    with pytest.raises(ValueError):            # detached again: the residual inputs are required
        unet(sample=x, timestep=t, encoder_hidden_states=emb.astype(np.float16))
    unet.close()
    for cn in cns:
        cn.close()


class StubXLEncoder:
    """text_encoder / text_encoder_2 of an SDXL pipeline: hidden_embeds + pooled_outputs (torch2coreml.py:408-441)."""

    def __init__(self, dim, pooled_dim, salt):
        self.dim, self.pooled_dim, self.salt = dim, pooled_dim, salt

    def __call__(self, input_ids):
        assert input_ids.dtype == np.float32 and input_ids.shape == (1, 77)
        rs = np.random.RandomState((int(input_ids.sum()) * 7 + self.salt) % (2 ** 31))
        return {"hidden_embeds": rs.randn(1, 77, self.dim).astype(np.float32),
                "pooled_outputs": rs.randn(1, self.pooled_dim).astype(np.float32)}


def test_sdxl_base_to_refiner_swap_with_scheduler_history_carried_over():
    """BASELINE config 4 "+ refiner" in miniature: the base UNet runs the first int(N * 0.8) evaluations, the
    refiner the rest with text_encoder_2-only embeddings and (2, 5) geometry + aesthetic-score time ids
    (StableDiffusionXLPipeline.swift:205-225, :326-358); PNDM's multistep history crosses the swap."""
    bcfg, rcfg = unet_ref.CONFIGS["mini-xl"], unet_ref.CONFIGS["mini-refiner"]
    b16, bsd = _ckpt(unet_ref.unet_param_shapes(bcfg), 31)
    r16, rsd = _ckpt(unet_ref.unet_param_shapes(rcfg), 81)
    hw = bcfg["sample_size"]
    base = HipModel(bcfg, b16, batch=2, attention_implementation="ORIGINAL")
    refiner = HipModel(dict(rcfg, num_time_ids=5), r16, batch=2, attention_implementation="ORIGINAL")
    # base context 128 = 64 (encoder 1) + 64 (encoder 2); refiner context 128 comes from encoder 2 alone
    e1, e2 = StubXLEncoder(64, 32, 1), StubXLEncoder(64, 64, 2)
    e2r = StubXLEncoder(128, 64, 2)
    pipe = HipStableDiffusionPipeline(e1, base, None, schedulers.PNDMScheduler(), StubTokenizer(), xl=True,
                                      force_zeros_for_empty_prompt=False, text_encoder_2=e2, tokenizer_2=StubTokenizer(),
                                      unet_refiner=refiner, refiner_start=0.8)
    steps, gs, seed = 9, 5.0, 3                                                   # 10 evaluations -> swap at index 8

    class Switch:                                                                  # encoder 2 serves both widths
        def __call__(self, input_ids):
            return (e2r if self.refiner else e2)(input_ids)
    sw = Switch()
    sw.refiner = False
    pipe.text_encoder_2 = sw
    orig = pipe._encode_prompt

    def enc(*a, for_refiner=False, **k):
        sw.refiner = for_refiner
        return orig(*a, for_refiner=for_refiner, **k)
    pipe._encode_prompt = enc
    fused = pipe("a prompt", num_inference_steps=steps, guidance_scale=gs, seed=seed, output_type="latent")
    assert len(fused.step_ms) == steps + 1
    host = pipe("a prompt", num_inference_steps=steps, guidance_scale=gs, seed=seed, output_type="latent", device_loop=False)
    assert psnr.compute_psnr(fused.images, host.images) >= 50.0
    # oracle: the same loop around the oracle UNets, one PNDM object across the swap
    emb_b, pooled_b = enc("a prompt", None, True, None, None)
    emb_r, pooled_r = enc("a prompt", None, True, None, None, for_refiner=True)
    H = hw * 8
    ids_b = np.tile(np.array([[H, H, 0, 0, H, H]], np.float32), (2, 1))
    ids_r = np.array([[H, H, 0, 0, 2.5], [H, H, 0, 0, 6.0]], np.float32)
    f16 = lambda a: torch.from_numpy(a.astype(np.float16).astype(np.float32))
    ub = _oracle_unet(bsd, bcfg, time_ids=f16(ids_b), text_embeds=f16(pooled_b))
    ur = _oracle_unet(rsd, rcfg, time_ids=f16(ids_r), text_embeds=f16(pooled_r))
    sch = scheduler_ref.PNDM()
    ts = sch.set_timesteps(steps)
    swap = int(len(ts) * 0.8)
    np.random.seed(seed)
    lat = np.random.randn(1, 4, hw, hw).astype(np.float16).astype(np.float32)
    for i, t in enumerate(ts):
        un, e = (ub, emb_b) if i < swap else (ur, emb_r)
        eps = un(np.concatenate([lat, lat]).astype(np.float16), np.array([t, t], np.float16), e.astype(np.float16))
        u, c = np.split(eps, 2)
        lat = sch.step((u + gs * (c - u)).astype(np.float32), int(t), lat).astype(np.float32)
    p = psnr.compute_psnr(fused.images, lat)
    assert p >= 59.0, f"base -> refiner loop: PSNR {p:.1f} dB vs the oracle loop"   # measured 65.0 (r3)
    base.close(), refiner.close()
