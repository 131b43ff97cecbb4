"""CPU suite: the oracle against the golden vectors written by oracle/pin_against_reference.py
(which compared it with the real reference modules) and against the reference's own
known-answer tests (numpy RNG golden of swift/StableDiffusionTests/StableDiffusionTests.swift:52-62)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import attention_ref, psnr, rng_ref, scheduler_ref, unet_ref, weights


def test_attention_oracle_matches_reference_golden():
    g = load_golden("attention_golden.npz")
    cases = sorted({k.split("_")[0] for k in g})
    assert cases
    for c in cases:
        b, h, d, sq, sk = g[f"{c}_meta"]
        for impl, fn in attention_ref.IMPLS.items():
            out = fn(g[f"{c}_q"], g[f"{c}_k"], g[f"{c}_v"], int(h), int(d))
            assert out.shape == (b, h * d, 1, sq)
            np.testing.assert_allclose(out, g[f"{c}_out"], atol=2e-6, rtol=0, err_msg=f"{c} {impl}")


def test_split_einsum_v2_rejects_tail_instead_of_dropping_it():
    q = np.zeros((1, 64, 1, 576), np.float32)
    k = np.zeros((1, 64, 1, 77), np.float32)
    with pytest.raises(ValueError):
        attention_ref.split_einsum_v2(q, k, k, 1, 64)
    # < 512 falls back to SPLIT_EINSUM (attention.py:88-92)
    q = np.random.RandomState(0).randn(1, 64, 1, 96).astype(np.float32)
    k = np.random.RandomState(1).randn(1, 64, 1, 77).astype(np.float32)
    np.testing.assert_allclose(attention_ref.split_einsum_v2(q, k, k, 1, 64), attention_ref.split_einsum(q, k, k, 1, 64))


def test_layernorm_oracle_matches_reference_golden():
    g = load_golden("layernorm_golden.npz")
    out = unet_ref.layer_norm_ane(torch.from_numpy(g["x"]), torch.from_numpy(g["w"]), torch.from_numpy(g["b"])).numpy()
    np.testing.assert_allclose(out, g["out"], atol=2e-6)


def test_layernorm_fold_identity_used_by_the_gemm_kernels():
    """The MFMA path never runs LayerNorm as a kernel: y = LN(x) W^T + b is evaluated as
    rstd*(x (W*gamma)^T) - rstd*mean*colsum(W*gamma) + (b + W beta)  (csrc/unet.cpp fold_layernorm,
    igemm.hip LNF epilogue), with W*gamma rounded to fp16 and Sum x, Sum x^2 accumulated in fp32 from
    the fp16 activations.  Same arithmetic in numpy vs the oracle's LayerNormANE + 1x1 conv."""
    rs = np.random.RandomState(3)
    C, N, M = 320, 192, 64
    x = (rs.randn(M, C) * 3.0 + 1.5).astype(np.float16).astype(np.float32)     # rows with a non-zero mean
    gamma, beta = (1.0 + 0.2 * rs.randn(C)).astype(np.float32), (0.3 * rs.randn(C)).astype(np.float32)
    W, b = (rs.randn(N, C) / np.sqrt(C)).astype(np.float32), rs.randn(N).astype(np.float32)
    ln = unet_ref.layer_norm_ane(torch.from_numpy(x.T.reshape(1, C, 1, M).copy()), torch.from_numpy(gamma),
                                 torch.from_numpy(beta)).numpy().reshape(C, M).T
    ref = ln.astype(np.float64) @ W.T.astype(np.float64) + b
    Wg = (W * gamma).astype(np.float16).astype(np.float32)                     # what the MFMA multiplies
    colsum = Wg.astype(np.float64).sum(1).astype(np.float32)
    bias2 = (b.astype(np.float64) + W.astype(np.float64) @ beta).astype(np.float32)
    s1, s2 = x.sum(1, dtype=np.float32), (x * x).sum(1, dtype=np.float32)
    mean = s1 / C
    rstd = 1.0 / np.sqrt(np.maximum(s2 / C - mean * mean, 0.0) + 1e-5)
    acc = x @ Wg.T
    got = acc * rstd[:, None] + (-rstd * mean)[:, None] * colsum[None, :] + bias2[None, :]
    assert psnr.compute_psnr(got, ref) > 75.0
    np.testing.assert_allclose(got, ref, atol=6e-3)


def test_timestep_embedding_oracle_matches_reference_golden():
    g = load_golden("timestep_golden.npz")
    out = unet_ref.timestep_embedding(torch.from_numpy(g["t"]), 320).numpy()
    np.testing.assert_array_equal(out, g["out"])
    half = 160     # [cos | sin] order (unet.py:721-724)
    np.testing.assert_allclose(out[:, 0], np.cos(g["t"]), atol=1e-4)
    np.testing.assert_allclose(out[:, half], np.sin(g["t"]), atol=1e-4)


def _run_unet_golden(name):
    g = load_golden(f"unet_{name}_golden.npz")
    cfg = unet_ref.CONFIGS[name]
    shapes = unet_ref.unet_param_shapes(cfg)
    assert sum(int(np.prod(s)) for s in shapes.values()) == int(g["n_params"])
    sd = weights.to_torch(weights.round_to_fp16(weights.make_state_dict(shapes, seed=int(g["seed"]))))
    kw = {}
    if "time_ids" in g:
        kw = dict(time_ids=torch.from_numpy(g["time_ids"]), text_embeds=torch.from_numpy(g["text_embeds"]))
    if cfg["support_controlnet"]:
        n = len(unet_ref.residual_shapes(cfg, 2))
        kw["additional_residuals"] = [torch.from_numpy(g[f"additional_residual_{i}"].astype(np.float32)) for i in range(n)]
    out = unet_ref.unet_forward(sd, cfg, torch.from_numpy(g["sample"].astype(np.float32)), torch.from_numpy(g["timestep"]),
                                torch.from_numpy(g["encoder_hidden_states"].astype(np.float32)), **kw).numpy()
    return out, g["noise_pred"]


@pytest.mark.parametrize("name", ["tiny", "mini", "mini-xl", "mini-control"])
def test_unet_oracle_matches_reference_golden(name):
    out, ref = _run_unet_golden(name)
    assert np.abs(out - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    assert psnr.compute_psnr(out, ref) > 100


def test_controlnet_oracle_matches_reference_golden():
    g = load_golden("controlnet_mini_golden.npz")
    cfg = unet_ref.CONFIGS["mini-control"]
    sd = weights.to_torch(weights.round_to_fp16(weights.make_state_dict(unet_ref.controlnet_param_shapes(cfg), seed=int(g["seed"]))))
    res = unet_ref.controlnet_forward(sd, cfg, torch.from_numpy(g["sample"].astype(np.float32)), torch.from_numpy(g["timestep"]),
                                      torch.from_numpy(g["encoder_hidden_states"].astype(np.float32)),
                                      torch.from_numpy(g["controlnet_cond"].astype(np.float32)))
    assert len(res) == 13
    for i, r in enumerate(res):
        np.testing.assert_allclose(r.numpy(), g[f"additional_residual_{i}"], atol=3e-5)


def test_parameter_count_of_the_baseline_model():
    n = sum(int(np.prod(s)) for s in unet_ref.unet_param_shapes(unet_ref.CONFIGS["sd21-base"]).values())
    assert n == 865_910_724          # SURVEY.md / BASELINE.md: 865.91 M
    assert len(unet_ref.residual_shapes(unet_ref.CONFIGS["sd15-control"], 2)) == 13   # controlnet.py:191-197


def test_numpy_rng_restatement_matches_swift_golden_and_numpy():
    r = rng_ref.NumpyLegacyRandom(rng_ref.GOLDEN_SEED).randn(rng_ref.GOLDEN_COUNT)
    np.testing.assert_allclose(r[-5:], rng_ref.GOLDEN_LAST5, atol=1e-8)
    np.random.seed(rng_ref.GOLDEN_SEED)
    assert np.array_equal(np.array(r), np.random.randn(rng_ref.GOLDEN_COUNT))


def test_psnr_matches_reference_formula():
    b = np.array([1.0, -2.0, 0.5])
    a = b + np.array([0.01, -0.01, 0.0])
    want = 20 * np.log10((2.0 + 1e-5) / (np.sqrt(2e-4 / 3) + 1e-10))
    assert abs(psnr.compute_psnr(a, b) - want) < 1e-9
    assert psnr.ABSOLUTE_MIN_PSNR == 35


def test_ddim_schedule_and_step_algebra():
    s = scheduler_ref.DDIM()
    ts = s.set_timesteps(20)
    assert list(ts[:3]) == [951, 901, 851] and ts[-1] == 1
    rs = np.random.RandomState(0)
    x, e = rs.randn(4, 8).astype(np.float32), rs.randn(4, 8).astype(np.float32)
    for t in (951, 1):
        cx, ce = s.coefficients(int(t))
        np.testing.assert_allclose(s.step(e, int(t), x), cx * x + ce * e, rtol=2e-5, atol=2e-6)


def test_pndm_schedule_matches_swift_restatement():
    s = scheduler_ref.PNDM()
    ts = s.set_timesteps(50)
    assert len(ts) == 51 and ts[0] == 981 and ts[1] == 961 and ts[2] == 961 and ts[-1] == 1
    x = np.ones((2, 3), np.float32)
    for t in ts[:6]:
        x = s.step(0.1 * np.ones_like(x), int(t), x)
    assert np.isfinite(x).all()


@pytest.mark.parametrize("name", ["mini-l", "mini-g"])
def test_clip_oracle_matches_the_installed_transformers(name):
    """The CLIP restatement (oracle/clip_ref.py) against transformers' CLIPTextModel(WithProjection) with the
    same random weights.  Third-party pin only: the reference holds no golden for the encoder (PARITY UNPINNED)."""
    transformers = pytest.importorskip("transformers")
    from oracle import clip_ref
    cfg = clip_ref.CONFIGS[name]
    sd = weights.to_torch(weights.make_state_dict(clip_ref.param_shapes(cfg), seed=7, gain=2.0))
    hf_cfg = transformers.CLIPTextConfig(**{k: v for k, v in cfg.items() if k != "architectures"}, bos_token_id=0, pad_token_id=1)
    cls = transformers.CLIPTextModelWithProjection if cfg.get("projection_dim") else transformers.CLIPTextModel
    model = cls(hf_cfg).eval()
    own = set(model.state_dict().keys())
    if not any(k.startswith("text_model.") for k in own):                # transformers 5.x dropped the wrapper prefix
        sd_hf = {k[len("text_model."):] if k.startswith("text_model.") else k: v for k, v in sd.items()}
    else:
        sd_hf = sd
    missing, unexpected = model.load_state_dict(sd_hf, strict=False)
    assert not [k for k in missing if "position_ids" not in k] and not unexpected
    ids = torch.from_numpy(np.random.RandomState(3).randint(3, cfg["vocab_size"], (1, 77)))
    ids[0, 20] = cfg["vocab_size"] - 1                                   # the arg-max token marks the pooled position
    with torch.no_grad():
        ref = model(input_ids=ids, output_hidden_states=True)
    mine = clip_ref.text_encoder_forward(sd, cfg, ids)
    np.testing.assert_allclose(mine["last_hidden_state"].numpy(), ref.last_hidden_state.numpy(), atol=2e-4)
    np.testing.assert_allclose(mine["hidden_embeds"].numpy(), ref.hidden_states[-2].numpy(), atol=2e-4)
    if cfg.get("projection_dim"):
        np.testing.assert_allclose(mine["text_embeds"].numpy(), ref.text_embeds.numpy(), atol=2e-4)
    else:
        np.testing.assert_allclose(mine["pooler_output"].numpy(), ref.pooler_output.numpy(), atol=2e-4)


def test_philox_block_function_matches_the_published_known_answer_vectors():
    """Philox4x32-10 (the generator of NvRandomSource.swift:25-63 / torch's CUDA RNG): the three known-answer vectors of the
    Random123 distribution (kat_vectors: all-zero, all-ones, digits of pi).  The C ABI's sd_philox_randn is compared with the
    oracle built on this function in tests/test_library_abi.py."""
    from oracle import rng_ref
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for counter, key, want in kat:
        assert rng_ref.philox4x32_10(counter, key) == want


def test_tokenizer_reproduces_the_reference_test_vectors(tmp_path):
    """The tokenizer is out of scope (transformers' CLIPTokenizer is used like the reference's Python pipeline does), but the
    reference's own test holds two known-answer prompts (swift/StableDiffusionTests/StableDiffusionTests.swift:43-49) for the
    vocabulary it ships: `load_tokenizer` on that vocabulary must give the same ids.  Reads the vocabulary from the reference
    tree in place, so it only runs where /root/reference exists (never on the GPU box)."""
    import os
    import shutil
    res = "/root/reference/swift/StableDiffusionTests/Resources"
    if not os.path.exists(os.path.join(res, "vocab.json")):
        pytest.skip("reference resources not available here")
    from python_hip_stable_diffusion.text_encoder import load_tokenizer
    folder = tmp_path / "tokenizer"
    folder.mkdir()
    for f in ("vocab.json", "merges.txt"):
        shutil.copy(os.path.join(res, f), folder / f)       # temporary copy outside the repository
    tok = load_tokenizer(str(folder))
    for prompt, want in (("a photo of an astronaut riding a horse on mars",
                          [49406, 320, 1125, 539, 550, 18376, 6765, 320, 4558, 525, 7496, 49407]),
                         ("Apple CoreML developer tools on a Macbook Air are fast",
                          [49406, 3055, 19622, 5780, 10929, 5771, 525, 320, 20617, 1922, 631, 1953, 49407])):
        assert tok(prompt)["input_ids"] == want


def test_fullsize_loop_goldens_carry_the_seeds_the_gpu_tests_regenerate_inputs_from():
    """tests/golden/loop20_{sdxl_base_refiner,sd15_controlnet}_golden.npz (oracle/pin_round3.py, the reference's own
    modules) store only output latents + the seeds of oracle/loop_inputs.py: the two must agree, and the generators must be
    deterministic and fp16-exact (what crosses the model boundary)."""
    from oracle import loop_inputs as L
    g4, g5 = load_golden("loop20_sdxl_base_refiner_golden.npz"), load_golden("loop20_sd15_controlnet_golden.npz")
    for k, v in L.SEEDS_XL.items():
        assert int(g4[f"seed_{k}"]) == v
    for k, v in L.SEEDS_CN.items():
        assert int(g5[f"seed_{k}"]) == v
    assert int(g4["hw"]) == L.HW_XL and int(g4["steps"]) == L.STEPS_XL and float(g4["guidance_scale"]) == L.GS_XL
    assert int(g4["swap"]) == int((L.STEPS_XL + 1) * L.SWAP_FRAC) == 16            # PNDM: steps + 1 evaluations
    assert int(g5["hw"]) == L.HW_CN and int(g5["steps"]) == L.STEPS_CN and float(g5["guidance_scale"]) == L.GS_CN
    assert g4["final"].shape == g4["latents_at_swap"].shape == (1, 4, L.HW_XL, L.HW_XL) and np.isfinite(g4["final"]).all()
    assert g5["final"].shape == g5["latents_step10"].shape == (1, 4, L.HW_CN, L.HW_CN) and np.isfinite(g5["final"]).all()
    a, b = L.xl_inputs(), L.xl_inputs()
    for k in a:
        assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], a[k].astype(np.float16).astype(np.float32))
    assert a["ehs_base"].shape == (2, 2048, 1, 77) and a["ehs_refiner"].shape == (2, 1280, 1, 77) and a["ids_refiner"].shape == (2, 5)
    c = L.cn_inputs()
    assert c["cond"].shape == (2, 3, 512, 512) and 0.0 <= c["cond"].min() and c["cond"].max() <= 1.0
    assert np.array_equal(c["cond"][0], c["cond"][1])                                # the same image for both CFG rows
    lat = L.initial_latents(L.SEEDS_CN["latents"], L.HW_CN)
    np.random.seed(93)
    assert np.array_equal(lat, np.random.randn(1, 4, 64, 64).astype(np.float16).astype(np.float32))


@pytest.mark.parametrize("name", ["mini", "sd"])
def test_vae_decoder_oracle_equals_the_golden_of_the_reference_blocks(name):
    """oracle/vae_ref.vae_decode against tests/golden/vae_decoder_*_golden.npz, written by oracle/pin_round4.py from the
    reference's OWN ResnetBlock2D(temb_channels=None, eps=1e-6) / Upsample2D / attention.original(heads=1) wired in the
    decoder's topology: the VAE decoder's arithmetic is pinned by the reference, its topology restated (diffusers absent)."""
    import torch
    from oracle import vae_ref, weights
    g = load_golden(f"vae_decoder_{name}_golden.npz")
    cfg = vae_ref.VAE_CONFIGS[name]
    hw = int(g["hw"])
    sd = weights.to_torch(weights.round_to_fp16(weights.make_state_dict(vae_ref.vae_decoder_param_shapes(cfg), seed=int(g["seed"]))))
    z = weights.seeded_normal((1, cfg["latent_channels"], hw, hw), int(g["z_seed"])).astype(np.float16).astype(np.float32)
    got = vae_ref.vae_decode(sd, cfg, torch.from_numpy(z)).numpy()
    assert got.shape == g["image"].shape
    assert np.abs(got - g["image"]).max() <= 1e-5 * max(1.0, np.abs(g["image"]).max())


@pytest.mark.parametrize("name", ["mini", "sd"])
def test_vae_encoder_oracle_equals_the_golden_of_the_reference_blocks(name):
    """oracle/vae_ref.vae_encode against tests/golden/vae_encoder_*_golden.npz, written by oracle/pin_round5.py from the reference's
    OWN ResnetBlock2D(temb_channels=None, eps=1e-6) / attention.original(heads=1) plus the two plain torch calls of the asymmetric
    down-sampler (F.pad (0,1,0,1) + F.conv2d stride 2) wired in the encoder's topology (torch2coreml.py:739-749): the encoder's
    arithmetic is pinned, its topology restated (diffusers absent)."""
    import torch
    from oracle import vae_ref, weights
    from oracle.pin_round5 import encoder_image
    g = load_golden(f"vae_encoder_{name}_golden.npz")
    cfg = vae_ref.VAE_CONFIGS[name]
    hw = int(g["hw"])
    sd16 = weights.make_state_dict(vae_ref.vae_encoder_param_shapes(cfg), seed=int(g["seed"]), dtype=np.float16, gain=float(g["gain"]))
    sd = weights.to_torch({k: v.astype(np.float32) for k, v in sd16.items()})
    got = vae_ref.vae_encode(sd, cfg, torch.from_numpy(encoder_image(hw, int(g["x_seed"])).astype(np.float32))).numpy()
    assert got.shape == g["moments"].shape == (1, 8, hw // 8, hw // 8)
    assert np.abs(got - g["moments"]).max() <= 1e-5 * max(1.0, np.abs(g["moments"]).max())


def test_vae_decoder_oracle_equals_the_reference_block_golden_at_the_benchmarked_size():
    """the 64x64-latent decode bench.py times (oracle/pin_round5.py --vae64; golden stored as fp16)"""
    import torch
    from oracle import vae_ref, weights
    g = load_golden("vae_decoder_sd64_golden.npz")
    cfg = vae_ref.VAE_CONFIGS["sd"]
    hw = int(g["hw"])
    sd = weights.to_torch(weights.round_to_fp16(weights.make_state_dict(vae_ref.vae_decoder_param_shapes(cfg), seed=int(g["seed"]))))
    z = weights.seeded_normal((1, cfg["latent_channels"], hw, hw), int(g["z_seed"])).astype(np.float16).astype(np.float32)
    got = vae_ref.vae_decode(sd, cfg, torch.from_numpy(z)).numpy()
    ref = g["image"].astype(np.float32)
    assert got.shape == ref.shape == (1, 3, 512, 512)
    assert np.abs(got - ref).max() <= 2.0 ** -11 * np.abs(ref).max() + 1e-6   # the golden's own fp16 storage rounding
