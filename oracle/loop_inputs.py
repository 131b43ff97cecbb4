"""Oracle (test infrastructure): the seeded, loop-invariant inputs of the full-size BASELINE config 4 / 5 loops.
Shared by ``oracle/pin_round3.py`` (which runs the REAL reference modules on them in the build container and stores
only the output latents under ``tests/golden/``) and by the GPU tests (which regenerate the same inputs here)."""
import numpy as np

from oracle import unet_ref, weights

HW_XL, STEPS_XL, GS_XL, SWAP_FRAC = 96, 19, 5.0, 0.8       # PNDM: 19 inference steps = 20 UNet evaluations
SEEDS_XL = dict(base=5, refiner=9, latents=93, ehs_base=201, pooled_base=202, ehs_refiner=203, pooled_refiner=204)
HW_CN, STEPS_CN, GS_CN = 64, 20, 7.5
SEEDS_CN = dict(unet=7, controlnet=71, latents=93, ehs=301, cond=302)


def f16_round(a):
    return np.asarray(a).astype(np.float16).astype(np.float32)


def initial_latents(seed, hw):
    np.random.seed(seed)                                   # pipeline.py:331, :726
    return np.random.randn(1, 4, hw, hw).astype(np.float16).astype(np.float32)


def xl_inputs():
    """Config 4 (SDXL-base -> refiner at 768x768): text embeddings / pooled embeddings / time ids of both stages, as the
    fp16 boundary delivers them ([uncond, cond] rows)."""
    bcfg, rcfg = unet_ref.CONFIGS["sdxl-base"], unet_ref.CONFIGS["sdxl-refiner"]
    px = HW_XL * 8
    s = SEEDS_XL
    return dict(
        ehs_base=f16_round(weights.seeded_normal((2, bcfg["cross_attention_dim"], 1, 77), s["ehs_base"])),
        pooled_base=f16_round(weights.seeded_normal((2, 1280), s["pooled_base"])),
        ids_base=np.tile(np.array([[px, px, 0, 0, px, px]], np.float32), (2, 1)),
        ehs_refiner=f16_round(weights.seeded_normal((2, rcfg["cross_attention_dim"], 1, 77), s["ehs_refiner"])),
        pooled_refiner=f16_round(weights.seeded_normal((2, 1280), s["pooled_refiner"])),
        ids_refiner=np.array([[px, px, 0, 0, 2.5], [px, px, 0, 0, 6.0]], np.float32),   # negative / positive aesthetic score
    )


def cn_inputs():
    """Config 5 (SD1.5 + ControlNet at 512x512): text embeddings and the conditioning image (same image for both CFG rows,
    values in [0, 1] like pipeline.py:717-721 delivers them)."""
    cfg = unet_ref.CONFIGS["sd15-control"]
    s = SEEDS_CN
    return dict(ehs=f16_round(weights.seeded_normal((2, cfg["cross_attention_dim"], 1, 77), s["ehs"])),
                cond=f16_round(np.tile(np.random.RandomState(s["cond"]).rand(1, 3, HW_CN * 8, HW_CN * 8), (2, 1, 1, 1))))
