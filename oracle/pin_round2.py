"""Round-2 pins against the REAL reference (build container only; needs ``/root/reference``).

  python -m oracle.pin_round2 --loop20      # 20-step DDIM loop, full SD2.1-base (BASELINE configs 1/2)
  python -m oracle.pin_round2 --sd21-768    # SD2.1-base at 96x96 latents (768x768)
  python -m oracle.pin_round2 --refiner     # SDXL refiner: mini-refiner + the full 2.26 B model at 96x96

Same protocol as ``pin_against_reference.py``: the reference's own ``unet.py`` modules are imported
from where they lie (bookkeeping shims only), loaded with the deterministic synthetic checkpoint
of ``oracle/weights.py`` and evaluated on torch-CPU fp32; only input/output tensors are stored
under ``tests/golden/``.  The oracle restatement is checked against the reference on the way.
"""
import argparse
import os
import time

import numpy as np
import torch

from oracle import scheduler_ref, unet_ref, weights
from oracle.pin_against_reference import GOLDEN, _maxdiff, _ref_kwargs, load_reference


def _build_reference(unet, name, seed):
    cfg = unet_ref.CONFIGS[name]
    xl = cfg["addition_embed_type"] == "text_time"
    shapes = unet_ref.unet_param_shapes(cfg)
    sd = weights.to_torch(weights.round_to_fp16(weights.make_state_dict(shapes, seed=seed)))
    cls = unet.UNet2DConditionModelXL if xl else unet.UNet2DConditionModel
    model = cls(**_ref_kwargs(cfg)).eval()
    assert set(model.state_dict().keys()) == set(shapes.keys())
    model.load_state_dict({k: v.clone() for k, v in sd.items()})
    return cfg, sd, model


def pin_loop20(unet, report):
    """pipeline.py:500-573 around the reference UNet2DConditionModel: 20 DDIM steps, guidance 7.5,
    latents np.random.seed(93) (pipeline.py:331,:726,:800), fp16 casts at the UNet boundary."""
    cfg, sd, model = _build_reference(unet, "sd21-base", 0)
    unet.ATTENTION_IMPLEMENTATION_IN_EFFECT = unet.AttentionImplementations.ORIGINAL
    np.random.seed(93)
    lat0 = np.random.randn(1, 4, 64, 64).astype(np.float16)
    ehs = weights.seeded_normal((2, 1024, 1, 77), 94).astype(np.float16)
    trace, eps_trace = [], []

    def ref_unet(x, ts, e):
        y = model(torch.from_numpy(x.astype(np.float32)), torch.from_numpy(ts.astype(np.float32)),
                  torch.from_numpy(e.astype(np.float32)))[0].numpy()
        eps_trace.append(y)
        return y

    t0 = time.time()
    final = scheduler_ref.denoise_loop(ref_unet, scheduler_ref.DDIM(), lat0.astype(np.float32), ehs, 20, 7.5,
                                       callback=lambda i, t, lat: trace.append(lat.copy()))
    dt = time.time() - t0
    # the oracle restatement on the first step (full-loop equality is implied by the per-forward pins)
    mine = unet_ref.unet_forward(sd, cfg, torch.from_numpy(np.concatenate([lat0] * 2).astype(np.float32)),
                                 torch.tensor([951.0, 951.0]), torch.from_numpy(ehs.astype(np.float32))).numpy()
    d = _maxdiff(mine, eps_trace[0])
    assert d < 2e-5 * max(1.0, np.abs(eps_trace[0]).max()), d
    report.append(f"loop20 sd21-base: 20 DDIM steps in {dt:.0f} s on {torch.get_num_threads()} threads; "
                  f"final latents std {final.std():.3f} max|x| {np.abs(final).max():.2f}; step-0 max|oracle-ref| {d:.2e}")
    np.savez_compressed(os.path.join(GOLDEN, "loop20_sd21-base_golden.npz"), seed=np.array(0),
                        latents0=lat0, ehs_seed=np.array(94), guidance_scale=np.array(7.5), steps=np.array(20),
                        trace=np.stack(trace).astype(np.float32), final=final.astype(np.float32),
                        noise_pred_step0=eps_trace[0].astype(np.float32))


def pin_sd21_768(unet, report):
    cfg, sd, model = _build_reference(unet, "sd21-base", 0)
    hw = 96
    sample = weights.seeded_normal((2, 4, hw, hw), 1).astype(np.float16).astype(np.float32)
    ehs = weights.seeded_normal((2, 1024, 1, 77), 2).astype(np.float16).astype(np.float32)
    ts = np.array([981.0, 981.0], np.float32)
    out = {}
    for impl in ("ORIGINAL", "SPLIT_EINSUM"):     # SPLIT_EINSUM_V2 drops the S_q % 512 tail (attention.py:86)
        unet.ATTENTION_IMPLEMENTATION_IN_EFFECT = getattr(unet.AttentionImplementations, impl)
        out[impl] = model(torch.from_numpy(sample), torch.from_numpy(ts), torch.from_numpy(ehs))[0].numpy()
    assert _maxdiff(out["ORIGINAL"], out["SPLIT_EINSUM"]) < 1e-4
    mine = unet_ref.unet_forward(sd, cfg, torch.from_numpy(sample), torch.from_numpy(ts), torch.from_numpy(ehs)).numpy()
    d = _maxdiff(mine, out["ORIGINAL"])
    assert d < 2e-5 * max(1.0, np.abs(mine).max()), d
    report.append(f"unet sd21-base @96x96: max|oracle-ref| {d:.2e}, ORIGINAL vs SPLIT_EINSUM "
                  f"{_maxdiff(out['ORIGINAL'], out['SPLIT_EINSUM']):.2e}")
    np.savez_compressed(os.path.join(GOLDEN, "unet_sd21-base-768_golden.npz"), seed=np.array(0), hw=np.array(hw),
                        sample=sample.astype(np.float16), timestep=ts, encoder_hidden_states=ehs.astype(np.float16),
                        noise_pred=out["ORIGINAL"].astype(np.float32))


def pin_refiner(unet, report, full=True):
    """UNet2DConditionModelXL with the refiner's conditioning: 5 time ids
    [h, w, crop_top, crop_left, aesthetic_score] (StableDiffusionXLPipeline.swift:326-358) and a
    projection input of 5*256 + 1280 = 2560 (unet.py:1076-1088)."""
    for name, seed, hw in (("mini-refiner", 81, None),) + ((("sdxl-refiner", 9, 96),) if full else ()):
        cfg, sd, model = _build_reference(unet, name, seed)
        hw = hw or cfg["sample_size"]
        n_ids = 5
        text_dim = cfg["projection_class_embeddings_input_dim"] - n_ids * cfg["addition_time_embed_dim"]
        sample = weights.seeded_normal((2, 4, hw, hw), seed + 1).astype(np.float16).astype(np.float32)
        ehs = weights.seeded_normal((2, cfg["cross_attention_dim"], 1, 77), seed + 2).astype(np.float16).astype(np.float32)
        text_embeds = weights.seeded_normal((2, text_dim), seed + 3).astype(np.float16).astype(np.float32)
        # negative / positive aesthetic scores 2.5 / 6.0 for the [uncond, cond] halves
        time_ids = np.array([[hw * 8, hw * 8, 0, 0, 2.5], [hw * 8, hw * 8, 0, 0, 6.0]], np.float32)
        ts = np.array([181.0, 181.0], np.float32)      # a refiner-range timestep (the last 20 % of the schedule)
        unet.ATTENTION_IMPLEMENTATION_IN_EFFECT = unet.AttentionImplementations.ORIGINAL
        ref = model(torch.from_numpy(sample), torch.from_numpy(ts), torch.from_numpy(ehs), torch.from_numpy(time_ids),
                    torch.from_numpy(text_embeds))[0].numpy()
        mine = unet_ref.unet_forward(sd, cfg, torch.from_numpy(sample), torch.from_numpy(ts), torch.from_numpy(ehs),
                                     time_ids=torch.from_numpy(time_ids), text_embeds=torch.from_numpy(text_embeds)).numpy()
        d = _maxdiff(mine, ref)
        assert d < 2e-5 * max(1.0, np.abs(ref).max()), (name, d)
        nparams = sum(int(np.prod(s)) for s in unet_ref.unet_param_shapes(cfg).values())
        report.append(f"unet {name} @{hw}x{hw}: {nparams / 1e6:.2f} M params, max|oracle-ref| {d:.2e} "
                      f"(max|y| {np.abs(ref).max():.3f})")
        np.savez_compressed(os.path.join(GOLDEN, f"unet_{name}_golden.npz"), seed=np.array(seed), hw=np.array(hw),
                            n_params=np.array(nparams), sample=sample.astype(np.float16), timestep=ts,
                            encoder_hidden_states=ehs.astype(np.float16), time_ids=time_ids, text_embeds=text_embeds,
                            noise_pred=ref.astype(np.float32))
        del model, sd


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--loop20", action="store_true")
    ap.add_argument("--sd21-768", action="store_true")
    ap.add_argument("--refiner", action="store_true")
    ap.add_argument("--mini-only", action="store_true", help="--refiner: skip the full 2.26 B model")
    args = ap.parse_args()
    torch.set_grad_enabled(False)
    _, _, unet, _ = load_reference()
    report = []
    if args.loop20:
        pin_loop20(unet, report)
    if args.sd21_768:
        pin_sd21_768(unet, report)
    if args.refiner:
        pin_refiner(unet, report, full=not args.mini_only)
    with open(os.path.join(GOLDEN, "PIN_REPORT_r02.txt"), "a") as f:
        f.write("\n".join(report) + "\n")
    print("\n".join(report))


if __name__ == "__main__":
    main()
