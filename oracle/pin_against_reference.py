"""Pin the oracle against the REAL reference and write the golden vectors.

Runs only in the build container (needs ``/root/reference``; the GPU box never does).
  python -m oracle.pin_against_reference            # check + (re)write tests/golden/*.npz
  python -m oracle.pin_against_reference --full     # also the full SD2.1-base golden (~1 min)

The reference's ``attention.py`` / ``layer_norm.py`` import unmodified; ``unet.py`` /
``controlnet.py`` need stand-ins for two bookkeeping imports (diffusers' ConfigMixin /
register_to_config / ModelMixin and coremltools' _macos_version).  The stand-ins below carry
no arithmetic.  Nothing from the reference is copied: its modules are imported from where
they lie, evaluated, and only input/output tensors are stored.
"""
import argparse
import functools
import inspect
import os
import sys
import types

import numpy as np
import torch

REF_ROOT = os.environ.get("SD_REFERENCE_ROOT", "/root/reference")
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def _install_shims():
    class _Cfg(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__

    def register_to_config(init):
        sig = inspect.signature(init)

        @functools.wraps(init)
        def wrapped(self, *a, **kw):
            bound = sig.bind(self, *a, **kw)
            bound.apply_defaults()
            cfg = {k: v for k, v in bound.arguments.items() if k not in ("self", "kwargs")}
            cfg.update(bound.arguments.get("kwargs", {}))
            object.__setattr__(self, "_cfg", _Cfg(cfg))
            init(self, *a, **kw)
        return wrapped

    class ConfigMixin:
        @property
        def config(self):
            return self._cfg

    dif = types.ModuleType("diffusers")
    dif.ModelMixin = torch.nn.Module
    cu = types.ModuleType("diffusers.configuration_utils")
    cu.ConfigMixin, cu.register_to_config = ConfigMixin, register_to_config
    dif.configuration_utils = cu
    ct = types.ModuleType("coremltools")
    ctm = types.ModuleType("coremltools.models")
    ctu = types.ModuleType("coremltools.models.utils")
    ctu._macos_version = lambda: (99, 0)
    ct.models, ctm.utils = ctm, ctu
    for name, mod in (("diffusers", dif), ("diffusers.configuration_utils", cu), ("coremltools", ct),
                      ("coremltools.models", ctm), ("coremltools.models.utils", ctu)):
        sys.modules.setdefault(name, mod)
    pkg = types.ModuleType("python_coreml_stable_diffusion")
    pkg.__path__ = [os.path.join(REF_ROOT, "python_coreml_stable_diffusion")]
    sys.modules["python_coreml_stable_diffusion"] = pkg


def load_reference():
    _install_shims()
    import importlib
    att = importlib.import_module("python_coreml_stable_diffusion.attention")
    ln = importlib.import_module("python_coreml_stable_diffusion.layer_norm")
    unet = importlib.import_module("python_coreml_stable_diffusion.unet")
    cn = importlib.import_module("python_coreml_stable_diffusion.controlnet")
    import logging
    for m in (att, unet):
        m.logger.setLevel(logging.WARNING)
    return att, ln, unet, cn


def _ref_kwargs(cfg):
    keys = ("in_channels", "out_channels", "sample_size", "block_out_channels", "down_block_types",
            "up_block_types", "layers_per_block", "attention_head_dim", "cross_attention_dim",
            "transformer_layers_per_block", "norm_num_groups", "norm_eps", "flip_sin_to_cos", "freq_shift",
            "addition_embed_type", "addition_time_embed_dim", "projection_class_embeddings_input_dim",
            "support_controlnet")
    return {k: cfg[k] for k in keys}


def _maxdiff(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also write the full SD2.1-base golden")
    ap.add_argument("--xl", action="store_true", help="also write the SDXL-base 768x768 golden (2.6 B params)")
    ap.add_argument("--control", action="store_true", help="also write the SD1.5 control-UNet golden")
    args = ap.parse_args()
    os.makedirs(GOLDEN, exist_ok=True)
    torch.set_grad_enabled(False)
    att, ln, unet, cn = load_reference()
    from oracle import attention_ref, unet_ref, weights, rng_ref
    report = []

    # ---- 1. attention: three formulations, several shapes -----------------------------------
    rs = np.random.RandomState(7)
    gold = {}
    cases = [(2, 2, 64, 128, 128), (1, 3, 64, 64, 77), (2, 1, 64, 1024, 77), (1, 2, 64, 1024, 1024),
             (1, 2, 40, 96, 77), (1, 1, 160, 64, 64)]
    for ci, (b, h, d, sq, sk) in enumerate(cases):
        q = rs.randn(b, h * d, 1, sq).astype(np.float32)
        k = rs.randn(b, h * d, 1, sk).astype(np.float32)
        v = rs.randn(b, h * d, 1, sk).astype(np.float32)
        outs = {}
        for name, fn in (("ORIGINAL", att.original), ("SPLIT_EINSUM", att.split_einsum),
                         ("SPLIT_EINSUM_V2", att.split_einsum_v2)):
            if name == "SPLIT_EINSUM_V2" and sq >= 512 and sq % 512:
                continue
            ref = fn(torch.from_numpy(q).double(), torch.from_numpy(k).double(), torch.from_numpy(v).double(),
                     None, h, d).numpy()
            mine = attention_ref.IMPLS[name](q, k, v, h, d)
            diff = _maxdiff(ref, mine)
            assert diff < 1e-12, (name, ci, diff)
            report.append(f"attention {name} case{ci} {b,h,d,sq,sk}: max|diff| {diff:.2e}")
            outs[name] = ref
        for name in outs:       # the reference's three variants agree with each other
            assert _maxdiff(outs[name], outs["ORIGINAL"]) < 1e-10
        if sq <= 128:           # keep the fixture small: store the small cases only
            gold[f"c{ci}_meta"] = np.array([b, h, d, sq, sk])
            gold[f"c{ci}_q"], gold[f"c{ci}_k"], gold[f"c{ci}_v"] = q, k, v
            gold[f"c{ci}_out"] = outs["ORIGINAL"].astype(np.float32)
    np.savez_compressed(os.path.join(GOLDEN, "attention_golden.npz"), **gold)

    # ---- 2. LayerNormANE incl. the bias/scale-order load hook ------------------------------------
    c, s = 96, 50
    x = rs.randn(2, c, 1, s).astype(np.float32)
    w = (1 + 0.2 * rs.randn(c)).astype(np.float32)
    bvec = (0.3 * rs.randn(c)).astype(np.float32)
    m = unet.LayerNormANE(c)
    m.load_state_dict({"weight": torch.from_numpy(w), "bias": torch.from_numpy(bvec)})
    ref = m(torch.from_numpy(x)).numpy()
    mine = unet_ref.layer_norm_ane(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(bvec)).numpy()
    assert _maxdiff(ref, mine) < 2e-6, _maxdiff(ref, mine)
    report.append(f"LayerNormANE: max|diff| {_maxdiff(ref, mine):.2e}")
    np.savez_compressed(os.path.join(GOLDEN, "layernorm_golden.npz"), x=x, w=w, b=bvec, out=ref)

    # ---- 3. timestep embedding -----------------------------------------------------------------
    t = torch.tensor([981.0, 1.0, 500.0])
    ref = unet.get_timestep_embedding(t, 320, flip_sin_to_cos=True, downscale_freq_shift=0).numpy()
    mine = unet_ref.timestep_embedding(t, 320).numpy()
    assert _maxdiff(ref, mine) == 0.0
    report.append("timestep embedding: bit-exact")
    np.savez_compressed(os.path.join(GOLDEN, "timestep_golden.npz"), t=t.numpy(), out=ref)

    # ---- 4. UNets: oracle == reference on seeded synthetic checkpoints; golden outputs ----------
    def run_unet(name, seed, impl_enum, full=False, hw=None):
        cfg = unet_ref.CONFIGS[name]
        xl = cfg["addition_embed_type"] == "text_time"
        shapes = unet_ref.unet_param_shapes(cfg)
        sd_np = weights.round_to_fp16(weights.make_state_dict(shapes, seed=seed))
        sd = weights.to_torch(sd_np)
        cls = unet.UNet2DConditionModelXL if xl else unet.UNet2DConditionModel
        model = cls(**_ref_kwargs(cfg)).eval()
        ref_keys = set(model.state_dict().keys())   # same inventory (module order differs, names/shapes match)
        assert ref_keys == set(shapes.keys()), (name, ref_keys ^ set(shapes.keys()))
        for k_, v_ in model.state_dict().items():
            assert tuple(v_.shape) == tuple(shapes[k_]), (k_, v_.shape, shapes[k_])
        model.load_state_dict({k_: v_.clone() for k_, v_ in sd.items()})
        hw = hw or cfg["sample_size"]
        bsz = 2
        sample = weights.seeded_normal((bsz, 4, hw, hw), seed + 1)
        ehs = weights.seeded_normal((bsz, cfg["cross_attention_dim"], 1, 77), seed + 2)
        # inputs are what the fp16 boundary would deliver (pipeline.py:531-536)
        sample = sample.astype(np.float16).astype(np.float32)
        ehs = ehs.astype(np.float16).astype(np.float32)
        ts = np.array([981.0, 981.0], np.float32)      # torch2coreml.py:854-863 protocol
        extra, extra_t = {}, []
        if xl:
            nt = (cfg["projection_class_embeddings_input_dim"] -
                  (cfg["cross_attention_dim"] if name.startswith("sdxl-base") else 0))
            text_dim = cfg["projection_class_embeddings_input_dim"] - 6 * cfg["addition_time_embed_dim"]
            time_ids = np.tile(np.array([[hw * 8, hw * 8, 0, 0, hw * 8, hw * 8]], np.float32), (bsz, 1))
            text_embeds = weights.seeded_normal((bsz, text_dim), seed + 3).astype(np.float16).astype(np.float32)
            extra = dict(time_ids=time_ids, text_embeds=text_embeds)
            extra_t = [torch.from_numpy(time_ids), torch.from_numpy(text_embeds)]
            del nt
        res = None
        if cfg["support_controlnet"]:
            res = [0.1 * weights.seeded_normal(s_, seed + 10 + i).astype(np.float16).astype(np.float32)
                   for i, s_ in enumerate(unet_ref.residual_shapes(cfg, bsz, hw, hw))]
            extra_t += [torch.from_numpy(r) for r in res]
        out = {}
        for impl in ("ORIGINAL", "SPLIT_EINSUM", "SPLIT_EINSUM_V2"):
            unet.ATTENTION_IMPLEMENTATION_IN_EFFECT = getattr(unet.AttentionImplementations, impl)
            ref = model(torch.from_numpy(sample), torch.from_numpy(ts), torch.from_numpy(ehs), *extra_t)[0].numpy()
            mine = unet_ref.unet_forward(sd, cfg, torch.from_numpy(sample), torch.from_numpy(ts),
                                         torch.from_numpy(ehs),
                                         time_ids=None if not xl else torch.from_numpy(extra["time_ids"]),
                                         text_embeds=None if not xl else torch.from_numpy(extra["text_embeds"]),
                                         additional_residuals=None if res is None else [torch.from_numpy(r) for r in res],
                                         impl=impl).numpy()
            d_ = _maxdiff(ref, mine)
            scale = float(np.abs(ref).max())
            assert d_ <= 2e-5 * max(1.0, scale), (name, impl, d_, scale)
            report.append(f"unet {name} {impl}: max|oracle-ref| {d_:.2e} (max|y| {scale:.3f}, std {ref.std():.3f})")
            out[impl] = ref
            if full:
                break
        nparams = sum(int(np.prod(s_)) for s_ in shapes.values())
        g = dict(seed=np.array(seed), config_name=np.array(name), n_params=np.array(nparams),
                 sample=sample.astype(np.float16), timestep=ts, encoder_hidden_states=ehs.astype(np.float16),
                 noise_pred=out["ORIGINAL"].astype(np.float32))
        for k_, v_ in extra.items():
            g[k_] = v_
        if res is not None:
            if full:   # full-size residuals are regenerated from their seeds by the test (keeps the fixture small)
                g["residuals_from_seed"] = np.array(1)
            else:
                for i, r in enumerate(res):
                    g[f"additional_residual_{i}"] = r.astype(np.float16)
        np.savez_compressed(os.path.join(GOLDEN, f"unet_{name}_golden.npz"), **g)
        return nparams

    for name, seed in (("tiny", 11), ("mini", 21), ("mini-xl", 31), ("mini-control", 41)):
        n = run_unet(name, seed, None)
        report.append(f"  {name}: {n / 1e6:.2f} M params")
    if args.full:
        n = run_unet("sd21-base", 0, None, full=True)
        assert n == 865_910_724, n            # 865.91 M parameters
        report.append(f"  sd21-base: {n / 1e6:.2f} M params")
    if args.xl:                               # BASELINE config 4: SDXL-base at 768x768 (96x96 latents)
        n = run_unet("sdxl-base", 5, None, full=True, hw=96)
        report.append(f"  sdxl-base @96x96: {n / 1e6:.2f} M params")
    if args.control:                          # BASELINE config 5: SD1.5 control-UNet (heads 8 -> d_head 40/80/160)
        n = run_unet("sd15-control", 7, None, full=True)
        report.append(f"  sd15-control: {n / 1e6:.2f} M params")

    # ---- 5. ControlNet ---------------------------------------------------------------------------
    cfg = unet_ref.CONFIGS["mini-control"]
    shapes = unet_ref.controlnet_param_shapes(cfg)
    sd_np = weights.round_to_fp16(weights.make_state_dict(shapes, seed=51))
    sd = weights.to_torch(sd_np)
    kw = {k_: cfg[k_] for k_ in ("in_channels", "block_out_channels", "down_block_types", "layers_per_block",
                                 "attention_head_dim", "cross_attention_dim", "transformer_layers_per_block",
                                 "norm_num_groups", "norm_eps", "flip_sin_to_cos", "freq_shift")}
    model = cn.ControlNetModel(**kw).eval()
    assert set(model.state_dict().keys()) == set(shapes.keys())
    for k_, v_ in model.state_dict().items():
        assert tuple(v_.shape) == tuple(shapes[k_]), (k_, v_.shape, shapes[k_])
    model.load_state_dict({k_: v_.clone() for k_, v_ in sd.items()})
    hw = cfg["sample_size"]
    sample = weights.seeded_normal((2, 4, hw, hw), 52).astype(np.float16).astype(np.float32)
    ehs = weights.seeded_normal((2, cfg["cross_attention_dim"], 1, 77), 53).astype(np.float16).astype(np.float32)
    cond = np.random.RandomState(54).rand(2, 3, hw * 8, hw * 8).astype(np.float16).astype(np.float32)
    ts = np.array([981.0, 981.0], np.float32)
    unet.ATTENTION_IMPLEMENTATION_IN_EFFECT = unet.AttentionImplementations.ORIGINAL
    down, mid = model(torch.from_numpy(sample).clone(), torch.from_numpy(ts), torch.from_numpy(ehs),
                      torch.from_numpy(cond))
    ref = [r.numpy() for r in down] + [mid.numpy()]
    mine = unet_ref.controlnet_forward(sd, cfg, torch.from_numpy(sample), torch.from_numpy(ts),
                                       torch.from_numpy(ehs), torch.from_numpy(cond))
    assert len(ref) == len(mine) == model.get_num_residuals()
    worst = max(_maxdiff(a, b.numpy()) for a, b in zip(ref, mine))
    assert worst < 2e-5, worst
    report.append(f"controlnet mini-control: {len(ref)} residuals, max|oracle-ref| {worst:.2e}")
    g = dict(seed=np.array(51), sample=sample.astype(np.float16), timestep=ts,
             encoder_hidden_states=ehs.astype(np.float16), controlnet_cond=cond.astype(np.float16))
    for i, r in enumerate(ref):
        g[f"additional_residual_{i}"] = r.astype(np.float32)
    np.savez_compressed(os.path.join(GOLDEN, "controlnet_mini_golden.npz"), **g)

    if args.control:
        cfgc = unet_ref.CONFIGS["sd15-control"]
        shapes = unet_ref.controlnet_param_shapes(cfgc)
        sd = weights.to_torch(weights.round_to_fp16(weights.make_state_dict(shapes, seed=71)))
        kwc = {k_: cfgc[k_] for k_ in ("in_channels", "block_out_channels", "down_block_types", "layers_per_block",
                                        "attention_head_dim", "cross_attention_dim", "transformer_layers_per_block",
                                        "norm_num_groups", "norm_eps", "flip_sin_to_cos", "freq_shift")}
        modelc = cn.ControlNetModel(**kwc).eval()
        modelc.load_state_dict({k_: v_.clone() for k_, v_ in sd.items()})
        hwc = 64
        samplec = weights.seeded_normal((2, 4, hwc, hwc), 72).astype(np.float16).astype(np.float32)
        ehsc = weights.seeded_normal((2, 768, 1, 77), 73).astype(np.float16).astype(np.float32)
        condc = np.random.RandomState(74).rand(2, 3, 512, 512).astype(np.float16).astype(np.float32)
        downc, midc = modelc(torch.from_numpy(samplec).clone(), torch.from_numpy(ts), torch.from_numpy(ehsc),
                             torch.from_numpy(condc))
        refc = [r.numpy() for r in downc] + [midc.numpy()]
        minec = unet_ref.controlnet_forward(sd, cfgc, torch.from_numpy(samplec), torch.from_numpy(ts),
                                            torch.from_numpy(ehsc), torch.from_numpy(condc))
        worst = max(_maxdiff(a_, b_.numpy()) for a_, b_ in zip(refc, minec))
        assert worst < 5e-5, worst
        report.append(f"controlnet sd15 (361.3 M params): 13 residuals, max|oracle-ref| {worst:.2e}")
        # fixture: inputs are regenerated from seeds by the test; store only the (strided) outputs
        gc = dict(seed=np.array(71), stride=np.array(16))
        for i, r in enumerate(refc):
            gc[f"additional_residual_{i}"] = r[:, ::16].astype(np.float16)    # every 16th channel: keeps the fixture small
        np.savez_compressed(os.path.join(GOLDEN, "controlnet_sd15_golden.npz"), **gc)
        del modelc, sd

    # ---- 6. numpy legacy RNG golden (StableDiffusionTests.swift:52-62) ---------------------------
    r = rng_ref.NumpyLegacyRandom(rng_ref.GOLDEN_SEED).randn(rng_ref.GOLDEN_COUNT)
    np.random.seed(rng_ref.GOLDEN_SEED)
    npy = np.random.randn(rng_ref.GOLDEN_COUNT)
    assert np.array_equal(np.array(r), npy)
    assert np.allclose(r[-5:], rng_ref.GOLDEN_LAST5, atol=1e-8)
    report.append("numpy legacy RNG: bit-exact vs numpy, matches the Swift golden")

    with open(os.path.join(GOLDEN, "PIN_REPORT.txt"), "w") as f:
        f.write("Generated by oracle/pin_against_reference.py against /root/reference\n")
        f.write("\n".join(report) + "\n")
    print("\n".join(report))


if __name__ == "__main__":
    main()
