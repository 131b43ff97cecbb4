"""``HipStableDiffusionPipeline`` - the reference's generation loop over MI355X model runners.

Mirrors ``CoreMLStableDiffusionPipeline`` (python_coreml_stable_diffusion/pipeline.py:47-589):
same constructor arguments, same ``__call__`` arguments and semantics (prompt, height/width
taken from the model, num_inference_steps, guidance_scale, negative_prompt, latents,
callback/callback_steps, controlnet_cond, SDXL size conditioning, unet_batch_one), same
tensor hand-offs to the model runners; ``get_hip_pipe`` / ``main`` mirror ``get_coreml_pipe`` /
``main`` (pipeline.py:607-697, :724-858) over a diffusers checkpoint directory instead of converted
``.mlpackage`` files.  Differences, all additive:
  * model runners are ``HipModel`` objects (``libsdmi355.so``) instead of ``CoreMLModel``;
  * when no per-step callback is requested and the scheduler exports linear tables, the whole loop
    (pipeline.py:500-573) - ControlNets included - runs device-resident in one call
    (``sd_unet_denoise_loop``); otherwise it steps through the boundary exactly like the reference
    and says so in one log line;
  * several prompts / several images per prompt are generated in one batched loop like the Swift
    pipeline's ``imageCount`` (StableDiffusionPipeline.swift:233-333) - the Python reference refuses
    them (pipeline.py:434-438); the UNet handle must have been built for that batch;
  * an SDXL refiner UNet takes over at step ``int(N * refiner_start)`` with its own conditioning
    (StableDiffusionXLPipeline.swift:205-225, :326-358);
  * ``seed`` is an argument (the reference seeds numpy globally in ``main``, pipeline.py:726) and
    the initial latents come from the bit-exact numpy legacy stream implemented in the library
    (``sd_numpy_randn``), so results do not depend on global RNG state;
  * ``--attention-implementation`` is a run-time flag of ``main`` (a conversion-time flag in the
    reference, torch2coreml.py:1678-1685).
"""
import argparse
import inspect
import json
import logging
import os
from types import SimpleNamespace

import numpy as np

from . import _lib
from .schedulers import SCHEDULER_MAP, get_available_schedulers  # noqa: F401  (pipeline.py:592-604)

logger = logging.getLogger(__name__)
VAE_DECODER_UPSAMPLE_FACTOR = 8          # pipeline.py:108
VAE_SCALING_FACTOR = 0.18215             # pipeline.py:314 (SD 1.x / 2.x); SDXL 0.13025 (main.swift:124)


class HipStableDiffusionPipeline:
    def __init__(self, text_encoder, unet, vae_decoder, scheduler, tokenizer, controlnet=None, xl=False,
                 force_zeros_for_empty_prompt=True, feature_extractor=None, safety_checker=None,
                 text_encoder_2=None, tokenizer_2=None, vae_scaling_factor=None, unet_refiner=None,
                 refiner_start=0.8, aesthetic_score=6.0, negative_aesthetic_score=2.5):
        self.text_encoder, self.text_encoder_2 = text_encoder, text_encoder_2
        self.tokenizer, self.tokenizer_2 = tokenizer, tokenizer_2
        self.unet, self.vae_decoder, self.scheduler = unet, vae_decoder, scheduler
        self.controlnet = controlnet
        self.xl = xl
        self.force_zeros_for_empty_prompt = force_zeros_for_empty_prompt
        self.feature_extractor, self.safety_checker = feature_extractor, safety_checker
        self.vae_scaling_factor = vae_scaling_factor or (0.13025 if xl else VAE_SCALING_FACTOR)
        # SDXL base -> refiner hand-off (StableDiffusionXLPipeline.swift:37-41, Configuration defaults)
        self.unet_refiner, self.refiner_start = unet_refiner, refiner_start
        self.aesthetic_score, self.negative_aesthetic_score = aesthetic_score, negative_aesthetic_score
        if safety_checker is None:
            logger.warning("safety checker disabled for %s", type(self).__name__)
        # static shapes come from the model, like the reference (pipeline.py:104-117)
        self.unet.in_channels = self.unet.expected_inputs["sample"]["shape"][1]
        latent_h, latent_w = self.unet.expected_inputs["sample"]["shape"][2:]
        self.height = latent_h * VAE_DECODER_UPSAMPLE_FACTOR
        self.width = latent_w * VAE_DECODER_UPSAMPLE_FACTOR
        logger.info("Stable Diffusion configured to generate %dx%d images", self.height, self.width)

    # ---- pipeline.py:123-257 ---------------------------------------------------------------
    def _encode_prompt(self, prompt, prompt_2=None, do_classifier_free_guidance=True, negative_prompt=None,
                       negative_prompt_2=None, for_refiner=False):
        batch_size = len(prompt) if isinstance(prompt, list) else 1
        if self.xl:
            prompts = [prompt, prompt_2 if prompt_2 is not None else prompt]
            if self.tokenizer is not None and not for_refiner:
                tokenizers, encoders = [self.tokenizer, self.tokenizer_2], [self.text_encoder, self.text_encoder_2]
            else:   # refiner: only tokenizer_2 / text_encoder_2 (pipeline.py:134-141; ...XLPipeline.swift:262-270)
                prompts, tokenizers, encoders = prompts[1:], [self.tokenizer_2], [self.text_encoder_2]
            key = "hidden_embeds"
        else:
            prompts, tokenizers, encoders, key = [prompt], [self.tokenizer], [self.text_encoder], "last_hidden_state"

        def encode(texts):
            embeds, pooled = [], None
            for text, tok, enc in zip(texts, tokenizers, encoders):
                ids = tok(text, padding="max_length", max_length=tok.model_max_length, truncation=True,
                          return_tensors="np").input_ids
                ids = np.asarray(ids).reshape(-1, ids.shape[-1])
                rows = [enc(input_ids=ids[i:i + 1].astype(np.float32)) for i in range(ids.shape[0])]   # ids as float32 (:173)
                embeds.append(np.concatenate([r[key] for r in rows]))
                if self.xl:
                    pooled = np.concatenate([r["pooled_outputs"] for r in rows])
            return np.concatenate(embeds, axis=-1), pooled

        prompt_embeds, pooled = encode(prompts)
        if do_classifier_free_guidance:
            if negative_prompt is None and self.force_zeros_for_empty_prompt:        # pipeline.py:183-187
                neg, neg_pooled = np.zeros_like(prompt_embeds), (np.zeros_like(pooled) if self.xl else None)
            else:
                negative_prompt = negative_prompt or ""
                negative_prompt_2 = negative_prompt_2 or negative_prompt
                if isinstance(prompt, list) and not isinstance(negative_prompt, list):
                    negative_prompt = batch_size * [negative_prompt]
                    negative_prompt_2 = batch_size * [negative_prompt_2]
                if type(prompt) is not type(negative_prompt):
                    raise TypeError(f"`negative_prompt` should be the same type to `prompt`, but got "
                                    f"{type(negative_prompt)} != {type(prompt)}.")
                if isinstance(negative_prompt, list) and len(negative_prompt) != batch_size:
                    raise ValueError("`negative_prompt` batch size does not match `prompt`")
                negs = [negative_prompt, negative_prompt_2]
                neg, neg_pooled = encode(negs[1:] if (self.xl and len(tokenizers) == 1) else negs[:len(tokenizers)])
            prompt_embeds = np.concatenate([neg, prompt_embeds])                     # [uncond, cond] (:245)
            if self.xl:
                pooled = np.concatenate([neg_pooled, pooled])
        return prompt_embeds.transpose(0, 2, 1)[:, :, None, :], pooled             # (B, C, 1, 77) (:252)

    # ---- pipeline.py:259-284 ---------------------------------------------------------------
    def run_controlnet(self, sample, timestep, encoder_hidden_states, controlnet_cond):
        if not self.controlnet:
            raise ValueError("Conditions for controlnet are given but the pipeline has no controlnet modules")
        if len(controlnet_cond) != len(self.controlnet):
            raise ValueError(f"need {len(self.controlnet)} controlnet conditions, got {len(controlnet_cond)}")
        total = None
        for cn, cond in zip(self.controlnet, controlnet_cond):
            out = cn(sample=sample.astype(np.float16), timestep=timestep.astype(np.float16),
                     encoder_hidden_states=encoder_hidden_states.astype(np.float16), controlnet_cond=cond)
            total = out if total is None else {k: total[k] + v for k, v in out.items()}
        return {k: v.astype(np.float16) for k, v in total.items()}

    def run_safety_checker(self, image):
        if self.safety_checker is None:
            return image, None
        raise NotImplementedError("safety checker is outside the MI355X hot path (SURVEY.md section 8)")

    # ---- pipeline.py:313-320 ---------------------------------------------------------------
    def decode_latents(self, latents):
        latents = 1 / self.vae_scaling_factor * latents
        dtype = self.vae_decoder.expected_inputs["z"]["dtype"]
        vb = self.vae_decoder.expected_inputs["z"]["shape"][0]
        # the decoder handle has a static batch like every model (pipeline.py:112-114): feed it in slices
        if latents.shape[0] % vb:
            raise ValueError(f"{latents.shape[0]} latents do not fill the VAE decoder's static batch of {vb}")
        image = np.concatenate([self.vae_decoder(z=np.ascontiguousarray(latents[i:i + vb]).astype(dtype))["image"]
                                for i in range(0, latents.shape[0], vb)])
        image = np.clip(image / 2 + 0.5, 0, 1)
        return image.transpose((0, 2, 3, 1))

    # ---- pipeline.py:322-344 ---------------------------------------------------------------
    def prepare_latents(self, batch_size, num_channels_latents, height, width, latents=None, seed=None, rng="numpy"):
        """``rng``: which of the reference's seed-exact sources draws the latents (RandomSource.swift, CLI --rng):
        "numpy" (np.random.seed + randn, the Python pipeline's), "torch" (torch CPU generator), "nvidia" (torch CUDA
        generator, Philox)."""
        shape = (batch_size, num_channels_latents, self.height // 8, self.width // 8)
        if latents is None:
            n = int(np.prod(shape))
            if seed is None:
                latents = np.random.randn(*shape).astype(np.float16)           # reference behaviour (:331)
            elif rng == "numpy":
                latents = _lib.numpy_randn(int(seed), n).reshape(shape).astype(np.float16)
            elif rng == "torch":
                latents = _lib.torch_randn(int(seed), n).reshape(shape).astype(np.float16)
            elif rng == "nvidia":
                latents = _lib.philox_randn(int(seed), n).reshape(shape).astype(np.float16)
            else:
                raise ValueError(f"rng must be 'numpy', 'torch' or 'nvidia', got {rng!r}")
        elif latents.shape != shape:
            raise ValueError(f"Unexpected latents shape, got {latents.shape}, expected {shape}")
        return latents * np.float32(self.scheduler.init_noise_sigma)

    def prepare_control_cond(self, controlnet_cond, do_classifier_free_guidance, batch_size, num_images_per_prompt):
        out = []
        for cond in controlnet_cond:                                              # pipeline.py:346-357
            cond = np.stack([cond] * batch_size * num_images_per_prompt)
            if do_classifier_free_guidance:
                cond = np.concatenate([cond] * 2)
            out.append(cond.astype(np.float16))
        return out

    def check_inputs(self, prompt, height, width, callback_steps):
        if height != self.height or width != self.width:                          # pipeline.py:359-365
            logger.warning("image %dx%d is fixed by the model; requested %sx%s ignored", self.height, self.width,
                           height, width)
        if not isinstance(prompt, (str, list)):
            raise ValueError(f"`prompt` has to be of type `str` or `list` but is {type(prompt)}")
        if self.height % 8 != 0 or self.width % 8 != 0:
            raise ValueError("`height` and `width` have to be divisible by 8")
        if callback_steps is None or not isinstance(callback_steps, int) or callback_steps <= 0:
            raise ValueError(f"`callback_steps` has to be a positive integer but is {callback_steps}")

    def prepare_extra_step_kwargs(self, eta):
        """pipeline.py:384-396: eta is forwarded only to schedulers whose ``step`` accepts it (DDIM)."""
        accepts_eta = "eta" in set(inspect.signature(self.scheduler.step).parameters.keys())
        return {"eta": eta} if accepts_eta else {}

    @staticmethod
    def _get_add_time_ids(original_size, crops_coords_top_left, target_size, dtype):
        return np.array(list(original_size + crops_coords_top_left + target_size)).astype(dtype)   # :398-401

    def _xl_kwargs(self, model, pooled, original_size, crops_coords_top_left, target_size, do_cfg, n_img, refiner):
        """SDXL micro-conditioning (pipeline.py:458-472); refiner geometry + aesthetic scores for the
        [uncond, cond] halves (StableDiffusionXLPipeline.swift:326-358)."""
        if refiner:
            neg = np.array(list(original_size + crops_coords_top_left) + [self.negative_aesthetic_score], np.float32)
            pos = np.array(list(original_size + crops_coords_top_left) + [self.aesthetic_score], np.float32)
            ids = np.concatenate([np.tile(neg, (n_img, 1)), np.tile(pos, (n_img, 1))]) if do_cfg else np.tile(pos, (n_img, 1))
        else:
            ids = self._get_add_time_ids(tuple(original_size), tuple(crops_coords_top_left), tuple(target_size), np.float32)
            if len(model.expected_inputs["time_ids"]["shape"]) > 1:
                ids = np.tile(ids[None], ((2 if do_cfg else 1) * n_img, 1))
            else:                                              # legacy flat (12,) layout (:466)
                ids = np.concatenate([ids] * ((2 if do_cfg else 1) * n_img))
        return {"text_embeds": pooled.astype(np.float16), "time_ids": ids.astype(np.float16)}

    @staticmethod
    def _per_image(emb, n_prompts, num_images_per_prompt, cfg_mul):
        """(cfg*P, ...) [uncond P | cond P] -> (cfg*P*n, ...) with every prompt's rows repeated n times."""
        if emb is None or num_images_per_prompt == 1:
            return emb
        halves = np.split(emb, cfg_mul)
        return np.concatenate([np.repeat(h, num_images_per_prompt, axis=0) for h in halves])

    # ---- pipeline.py:403-589 ---------------------------------------------------------------
    def __call__(self, prompt, height=512, width=512, num_inference_steps=50, guidance_scale=7.5, negative_prompt=None,
                 num_images_per_prompt=1, eta=0.0, latents=None, output_type="np", return_dict=True, callback=None,
                 callback_steps=1, controlnet_cond=None, original_size=None, crops_coords_top_left=(0, 0),
                 target_size=None, unet_batch_one=False, seed=None, device_loop=True, rng="numpy", **kwargs):
        self.check_inputs(prompt, height, width, callback_steps)
        height, width = self.height, self.width
        original_size = original_size or (height, width)
        target_size = target_size or (height, width)
        batch_size = 1 if isinstance(prompt, str) else len(prompt)
        n_img = batch_size * num_images_per_prompt
        do_cfg = guidance_scale > 1.0                                              # pipeline.py:443
        cfg_mul = 2 if do_cfg else 1
        batch_one = bool(unet_batch_one and do_cfg)
        model_batch = self.unet.expected_inputs["sample"]["shape"][0]
        need = 1 if batch_one else cfg_mul * n_img
        if model_batch != need or (batch_one and n_img != 1):
            raise ValueError(
                f"the UNet handle has a static batch of {model_batch} (pipeline.py:112-114) but this call needs {need}: "
                f"{batch_size} prompt(s) x {num_images_per_prompt} image(s) x {cfg_mul} (classifier-free guidance)"
                + (", evaluated one half at a time (--unet-batch-one)" if batch_one else "")
                + f"; build HipModel(batch={need})")

        text_embeddings, pooled = self._encode_prompt(prompt, None, do_cfg, negative_prompt, None)
        text_embeddings = self._per_image(text_embeddings, batch_size, num_images_per_prompt, cfg_mul)
        pooled = self._per_image(pooled, batch_size, num_images_per_prompt, cfg_mul)
        extra = {}
        if self.xl:                                                                # pipeline.py:458-472
            is_refiner = self.unet.expected_inputs["time_ids"]["shape"][-1] == 5   # ...XLPipeline.swift:151-154
            extra = self._xl_kwargs(self.unet, pooled, original_size, crops_coords_top_left, target_size, do_cfg, n_img,
                                    is_refiner)

        self.scheduler.set_timesteps(num_inference_steps)
        timesteps = self.scheduler.timesteps
        latents = self.prepare_latents(n_img, self.unet.in_channels, height, width, latents, seed, rng)
        if controlnet_cond:
            controlnet_cond = self.prepare_control_cond(controlnet_cond, do_cfg, batch_size, num_images_per_prompt)
        extra_step_kwargs = self.prepare_extra_step_kwargs(eta)

        # stages: (model, encoder_hidden_states, extra kwargs, [first, last) steps); two when a refiner takes over
        stages = [(self.unet, text_embeddings, extra, 0, len(timesteps))]
        if self.unet_refiner is not None:
            if not self.xl:
                raise ValueError("unet_refiner needs an SDXL pipeline (xl=True)")
            swap = int(float(len(timesteps)) * self.refiner_start)                 # ...XLPipeline.swift:206
            if swap < len(timesteps):
                r_emb, r_pooled = self._encode_prompt(prompt, None, do_cfg, negative_prompt, None, for_refiner=True)
                r_emb = self._per_image(r_emb, batch_size, num_images_per_prompt, cfg_mul)
                r_pooled = self._per_image(r_pooled, batch_size, num_images_per_prompt, cfg_mul)
                r_extra = self._xl_kwargs(self.unet_refiner, r_pooled, original_size, crops_coords_top_left, target_size,
                                          do_cfg, n_img, True)
                stages = [(self.unet, text_embeddings, extra, 0, swap)] if swap > 0 else []
                stages.append((self.unet_refiner, r_emb, r_extra, swap, len(timesteps)))

        reasons = []
        if not device_loop:
            reasons.append("device_loop=False")
        if callback is not None:
            reasons.append("a per-step callback was given")
        if batch_one:
            reasons.append("--unet-batch-one")
        if not hasattr(self.scheduler, "device_tables"):
            reasons.append(f"{type(self.scheduler).__name__} exports no device tables")
        if not all(hasattr(m, "denoise_loop") for m, *_ in stages):
            reasons.append("the UNet model runner has no denoise_loop")
        if controlnet_cond and not (self.controlnet and hasattr(self.unet, "attach_controlnets")
                                    and all(hasattr(cn, "set_controlnet_cond") for cn in self.controlnet)):
            reasons.append("the ControlNet model runners cannot hand residuals over on the device")
        if eta and extra_step_kwargs:
            reasons.append("eta != 0")
        fused = not reasons
        step_ms = None
        if fused:
            if not controlnet_cond and getattr(self.unet, "_attached", None):
                # a previous call attached ControlNets: without conditioning images this call must not run them with the
                # stale ones (the reference would fail on the missing additional_residual_* inputs, pipeline.py:519-529)
                self.unet.attach_controlnets([])
            if controlnet_cond:
                if len(controlnet_cond) != len(self.controlnet):
                    raise ValueError(f"need {len(self.controlnet)} controlnet conditions, got {len(controlnet_cond)}")
                for cn, cond in zip(self.controlnet, controlnet_cond):
                    cn.set_controlnet_cond(cond)
                self.unet.attach_controlnets(self.controlnet)
            ts, coef, hist = self.scheduler.device_tables()
            scale = self.scheduler.sample_scale() if hasattr(self.scheduler, "sample_scale") else None
            lat = latents.astype(np.float32)
            state = np.zeros((hist,) + lat.shape, np.float32) if (hist and len(stages) > 1) else None
            noise = self.scheduler.step_noise(lat.shape) if hasattr(self.scheduler, "step_noise") else None   # ancestral samplers
            step_ms = []
            for model, emb, kw, first, last in stages:
                lat, ms = model.denoise_loop(lat, ts[first:last], coef[first:last], guidance_scale, history=hist,
                                             sample_scale=None if scale is None else scale[first:last],
                                             history_state=state, step_noise=None if noise is None else noise[first:last],
                                             encoder_hidden_states=emb.astype(np.float16), **kw)
                step_ms.append(ms)
            latents, step_ms = lat, np.concatenate(step_ms)
        else:
            logger.info("stepping the denoising loop through the host boundary (%s)", "; ".join(reasons))
            if getattr(self.unet, "_attached", None):
                self.unet.attach_controlnets([])           # host-stepped ControlNet residuals travel like the reference's
            for i, t in enumerate(timesteps):                                      # pipeline.py:500-573
                model, emb, kw, _, _ = next(s for s in stages if s[3] <= i < s[4])
                x = np.concatenate([latents] * 2) if do_cfg else latents
                x = np.asarray(self.scheduler.scale_model_input(x, t))
                timestep = np.array([t] * (cfg_mul * n_img), np.float16)
                unet_kwargs = dict(kw)
                if controlnet_cond:
                    unet_kwargs.update(self.run_controlnet(x, timestep, emb, controlnet_cond))
                if not batch_one:
                    noise_pred = model(sample=x.astype(np.float16), timestep=timestep,
                                       encoder_hidden_states=emb.astype(np.float16), **unet_kwargs)["noise_pred"]
                    if do_cfg:
                        noise_uncond, noise_text = np.split(noise_pred, 2)
                else:                                                              # pipeline.py:537-556: one half at a time
                    x16, e16, t1 = x.astype(np.float16), emb.astype(np.float16), np.array([t], np.float16)
                    halves = []
                    for h in range(2):
                        kw_h = {k: (v[h:h + 1] if v.shape[0] == 2 else v) for k, v in unet_kwargs.items()}
                        halves.append(model(sample=x16[h:h + 1], timestep=t1, encoder_hidden_states=e16[h:h + 1],
                                            **kw_h)["noise_pred"])
                    noise_uncond, noise_text = halves
                if do_cfg:
                    noise_pred = noise_uncond + guidance_scale * (noise_text - noise_uncond)
                latents = self.scheduler.step(noise_pred, t, latents.astype(np.float32), **extra_step_kwargs).prev_sample
                if callback is not None and i % callback_steps == 0:
                    callback(i, t, latents)

        if output_type == "latent" or self.vae_decoder is None:
            image = latents
        else:
            image = self.decode_latents(latents)
        image, has_nsfw = self.run_safety_checker(image)
        if output_type == "pil" and image.ndim == 4 and image.shape[-1] == 3:
            image = self.numpy_to_pil(image)
        if not return_dict:
            return image, has_nsfw
        return PipelineOutput(images=image, nsfw_content_detected=has_nsfw, step_ms=step_ms, latents=latents)

    @staticmethod
    def numpy_to_pil(images):
        from PIL import Image
        return [Image.fromarray((im * 255).round().astype("uint8")) for im in images]


class PipelineOutput(SimpleNamespace):
    """``StableDiffusionPipelineOutput`` stand-in: attribute access like diffusers' output class and the
    ``image["images"][0]`` indexing the reference's ``main`` uses (pipeline.py:782)."""

    def __getitem__(self, key):
        return getattr(self, key)


# ---------------------------------------------------------------------------------------------
# get_coreml_pipe / main of the reference over a diffusers checkpoint directory
# ---------------------------------------------------------------------------------------------
def get_available_compute_units():
    """coreml_model.py:205-206 lists Core ML compute units; accepted for CLI compatibility, ignored."""
    return ("ALL", "CPU_AND_GPU", "CPU_ONLY", "CPU_AND_NE")


def _read_json(path):
    with open(path) as f:
        return json.load(f)


def _find_weights(folder):
    for name in ("diffusion_pytorch_model.fp16.safetensors", "diffusion_pytorch_model.safetensors", "model.fp16.safetensors",
                 "model.safetensors"):
        p = os.path.join(folder, name)
        if os.path.exists(p):
            return p
    raise FileNotFoundError(f"no .safetensors checkpoint under {folder} (coreml_model.py:176-178)")


def checkpoint_scheduler_config(model_dir):
    """``pytorch_pipe.scheduler.config`` (pipeline.py:738-741) read from ``scheduler/scheduler_config.json`` of a
    diffusers checkpoint directory; a checkpoint without one gets Stable Diffusion's PNDM config."""
    from .schedulers import load_scheduler_config
    path = os.path.join(model_dir, "scheduler", "scheduler_config.json")
    if os.path.exists(path):
        return load_scheduler_config(path)
    logger.warning("%s not found: assuming Stable Diffusion's PNDM scheduler config", path)
    return load_scheduler_config(dict(_class_name="PNDMScheduler", beta_start=0.00085, beta_end=0.012,
                                      beta_schedule="scaled_linear", skip_prk_steps=True, steps_offset=1))


def get_hip_pipe(model_dir, model_version, compute_unit="ALL", scheduler_override=None, controlnet_models=None,
                 force_zeros_for_empty_prompt=True, sources=None, attention_implementation="SPLIT_EINSUM",
                 num_images=1, guidance_scale=7.5, unet_batch_one=False, latent_size=None, device=0,
                 refiner_dir=None, tokenizer_factory=None, text_encoder_factory=None, vae_dtype=None):
    """``get_coreml_pipe`` (pipeline.py:607-697) without the conversion step: ``model_dir`` is a diffusers
    checkpoint directory (``unet/``, ``vae/``, ``text_encoder/``, ``tokenizer/``, ``scheduler/`` [,
    ``text_encoder_2/``, ``tokenizer_2/``]) instead of a folder of ``.mlpackage`` files; ControlNets are
    diffusers ControlNet directories.  ``compute_unit`` / ``sources`` are accepted and ignored.  Static shapes
    (pipeline.py:112-114) are fixed here: UNet batch = (2 if guidance_scale > 1 else 1) * num_images.
    ``vae_dtype``: compute precision of the VAE decoder; default = the reference's conversion rule (torch2coreml.py:570-578):
    float32 for an SDXL checkpoint's own VAE (it overflows fp16), float16 otherwise; pass ``np.float16`` for an
    fp16-safe replacement VAE (``--custom-vae-version``)."""
    if not os.path.isdir(model_dir):
        raise FileNotFoundError(f"{model_dir} not found (coreml_model.py:176-178)")
    from . import text_encoder as te
    from .hip_model import HipModel, HipVaeDecoder
    xl = "xl" in model_version
    do_cfg = guidance_scale > 1.0
    batch = 1 if (unet_batch_one and do_cfg) else (2 if do_cfg else 1) * num_images
    if scheduler_override is not None and not isinstance(scheduler_override, str):
        logger.warning("Overriding scheduler in pipeline: Override=%s", type(scheduler_override).__name__)
        scheduler = scheduler_override
    else:
        # pipeline.py:738-741: SCHEDULER_MAP[name].from_config(pytorch_pipe.scheduler.config) - betas, steps_offset,
        # timestep_spacing and prediction_type are the CHECKPOINT's (a v-prediction model with an epsilon scheduler
        # generates garbage); unsupported config values raise instead of being ignored
        sc_cfg = checkpoint_scheduler_config(model_dir)
        name = scheduler_override or sc_cfg["_class_name"].replace("Scheduler", "")
        if name not in SCHEDULER_MAP:
            raise NotImplementedError(f"scheduler {name} is not one of {sorted(SCHEDULER_MAP)}")
        if scheduler_override is not None:
            logger.warning("Overriding scheduler in pipeline: Override=%s", name)
        scheduler = SCHEDULER_MAP[name].from_config(sc_cfg)

    def load_unet(folder, kind="unet", support_controlnet=False):
        cfg = dict(_read_json(os.path.join(folder, "config.json")))
        cfg["support_controlnet"] = support_controlnet
        size = latent_size or cfg.get("sample_size", 64)
        return HipModel(cfg, _find_weights(folder), kind=kind, batch=batch, latent_height=size, latent_width=size,
                        attention_implementation=attention_implementation, device=device)

    kwargs = dict(xl=xl, force_zeros_for_empty_prompt=force_zeros_for_empty_prompt, safety_checker=None)
    logger.info("Loading models in HBM from %s", model_dir)
    kwargs["unet"] = load_unet(os.path.join(model_dir, "unet"), support_controlnet=bool(controlnet_models))
    kwargs["controlnet"] = ([load_unet(d, kind="controlnet") for d in controlnet_models] if controlnet_models else None)
    vcfg = _read_json(os.path.join(model_dir, "vae", "config.json"))
    lat = kwargs["unet"].latent_height
    kwargs["vae_decoder"] = HipVaeDecoder(
        dict(latent_channels=vcfg.get("latent_channels", 4), out_channels=vcfg.get("out_channels", 3),
             block_out_channels=tuple(vcfg["block_out_channels"]), layers_per_block=vcfg.get("layers_per_block", 2)),
        _find_weights(os.path.join(model_dir, "vae")), batch=1, latent_height=lat,
        latent_width=kwargs["unet"].latent_width, device=device,
        dtype=np.dtype(vae_dtype) if vae_dtype is not None else (np.float32 if xl else np.float16))
    kwargs["vae_scaling_factor"] = vcfg.get("scaling_factor")
    make_tok = tokenizer_factory or te.load_tokenizer
    make_enc = text_encoder_factory or (lambda folder, **kw: te.HipTextEncoder.from_pretrained(folder, device=device, **kw))
    if xl:
        has_first = os.path.isdir(os.path.join(model_dir, "text_encoder"))       # the refiner ships text_encoder_2 only
        kwargs["tokenizer"] = make_tok(os.path.join(model_dir, "tokenizer")) if has_first else None
        kwargs["text_encoder"] = (make_enc(os.path.join(model_dir, "text_encoder"), xl=True) if has_first else None)
        kwargs["tokenizer_2"] = make_tok(os.path.join(model_dir, "tokenizer_2"))
        kwargs["text_encoder_2"] = make_enc(os.path.join(model_dir, "text_encoder_2"), xl=True)
        if refiner_dir:
            kwargs["unet_refiner"] = load_unet(os.path.join(refiner_dir, "unet"))
    else:
        kwargs["tokenizer"] = make_tok(os.path.join(model_dir, "tokenizer"))
        kwargs["text_encoder"] = make_enc(os.path.join(model_dir, "text_encoder"))
    logger.info("Initializing HIP pipe for image generation")
    return HipStableDiffusionPipeline(scheduler=scheduler, **kwargs)


def get_image_path(args, **override_kwargs):
    """mkdir the output folder and encode metadata in the file name (pipeline.py:700-714)."""
    out_folder = os.path.join(args.o, "_".join(args.prompt.replace("/", "_").rsplit(" ")))
    os.makedirs(out_folder, exist_ok=True)
    out_fname = f"randomSeed_{override_kwargs.get('seed', None) or args.seed}"
    out_fname += f"_computeUnit_{override_kwargs.get('compute_unit', None) or args.compute_unit}"
    out_fname += f"_modelVersion_{override_kwargs.get('model_version', None) or args.model_version.replace('/', '_')}"
    if args.scheduler is not None:
        out_fname += f"_customScheduler_{override_kwargs.get('scheduler', None) or args.scheduler}"
        out_fname += f"_numInferenceSteps{override_kwargs.get('num_inference_steps', None) or args.num_inference_steps}"
    return os.path.join(out_folder, out_fname + ".png")


def prepare_controlnet_cond(image_path, height, width):
    """pipeline.py:717-721: RGB, LANCZOS resize, CHW in [0, 1]."""
    from PIL import Image
    image = Image.open(image_path).convert("RGB")
    image = image.resize((height, width), resample=Image.LANCZOS)
    return np.array(image).transpose(2, 0, 1) / 255.0


def build_parser():
    """The reference's flags with their names and defaults (pipeline.py:785-855) plus the run-time
    ``--attention-implementation`` (torch2coreml.py:1678-1685)."""
    parser = argparse.ArgumentParser(prog="python -m python_hip_stable_diffusion.pipeline")
    parser.add_argument("--prompt", required=True, help="The text prompt to be used for text-to-image generation.")
    parser.add_argument("-i", required=True, help="Path to a diffusers checkpoint directory (unet/, vae/, text_encoder/, "
                                                  "tokenizer/, scheduler/); replaces the folder of converted .mlpackage files")
    parser.add_argument("-o", required=True)
    parser.add_argument("--seed", "-s", default=93, type=int, help="Random seed to be able to reproduce results")
    parser.add_argument("--model-version", default="CompVis/stable-diffusion-v1-4",
                        help="The pre-trained model checkpoint and configuration to restore.")
    parser.add_argument("--compute-unit", choices=get_available_compute_units(), default="ALL",
                        help="Accepted for compatibility with the Core ML pipeline and ignored: compute runs on the MI355X.")
    parser.add_argument("--scheduler", choices=tuple(SCHEDULER_MAP.keys()), default=None,
                        help="The scheduler to use for running the reverse diffusion process. If not specified, the "
                             "default scheduler of the checkpoint is utilized.  It is built from the checkpoint's "
                             "scheduler/scheduler_config.json; keys that file omits take the diffusers class defaults, of which "
                             "clip_sample=True (DDIM) and skip_prk_steps=False (PNDM) are not implemented: spell out "
                             "\"clip_sample\": false / \"skip_prk_steps\": true there, as the Stable Diffusion checkpoints do")
    parser.add_argument("--num-inference-steps", default=50, type=int,
                        help="The number of iterations the unet model will be executed throughout the reverse diffusion process")
    parser.add_argument("--guidance-scale", default=7.5, type=float,
                        help="Controls the influence of the text prompt on sampling process (0=random images)")
    parser.add_argument("--controlnet", nargs="*", type=str,
                        help="Enables ControlNet (diffusers ControlNet directories) and the control-UNet inputs. "
                             "For Multi-Controlnet, provide the directories separated by spaces.")
    parser.add_argument("--controlnet-inputs", nargs="*", type=str,
                        help="Image paths for ControlNet inputs, in the order of --controlnet.")
    parser.add_argument("--negative-prompt", default=None,
                        help="The negative text prompt to be used for text-to-image generation.")
    parser.add_argument("--unet-batch-one", action="store_true",
                        help="Do not batch unet predictions for the prompt and negative prompt.")
    parser.add_argument("--model-sources", default=None, choices=["packages", "compiled"],
                        help="Accepted for compatibility and ignored (there is no conversion step).")
    parser.add_argument("--attention-implementation", choices=tuple(_lib.ATTENTION_IMPLEMENTATIONS), default="SPLIT_EINSUM",
                        help="Attention schedule of the UNet kernels (a conversion-time flag in the reference).")
    parser.add_argument("--refiner", default=None, help="SDXL: diffusers directory of the refiner checkpoint")
    parser.add_argument("--rng", choices=("numpy", "torch", "nvidia"), default="numpy",
                        help="Seed-exact random source of the initial latents (swift/StableDiffusionCLI/main.swift --rng): "
                             "numpy = np.random.seed + randn (this pipeline's and the reference's default), torch = torch's CPU "
                             "generator, nvidia = torch's CUDA generator (Philox)")
    return parser


def main(args):
    """pipeline.py:724-782."""
    logger.info("Setting random seed to %d", args.seed)
    np.random.seed(args.seed)
    scheduler = args.scheduler          # a name: get_hip_pipe builds it from the checkpoint's config (pipeline.py:738-741)
    xl = "xl" in args.model_version
    force_zeros = False                                                            # pipeline.py:744-746
    idx = os.path.join(args.i, "model_index.json")
    if xl and os.path.exists(idx):
        force_zeros = bool(_read_json(idx).get("force_zeros_for_empty_prompt", False))
    pipe = get_hip_pipe(args.i, args.model_version, args.compute_unit, scheduler_override=scheduler,
                        controlnet_models=args.controlnet, force_zeros_for_empty_prompt=force_zeros,
                        sources=args.model_sources, attention_implementation=args.attention_implementation,
                        guidance_scale=args.guidance_scale, unet_batch_one=args.unet_batch_one, refiner_dir=args.refiner)
    controlnet_cond = None
    if args.controlnet:
        controlnet_cond = [prepare_controlnet_cond(args.controlnet_inputs[i], pipe.height, pipe.width)
                           for i, _ in enumerate(args.controlnet)]
    logger.info("Beginning image generation.")
    image = pipe(prompt=args.prompt, height=pipe.height, width=pipe.width, num_inference_steps=args.num_inference_steps,
                 guidance_scale=args.guidance_scale, controlnet_cond=controlnet_cond, negative_prompt=args.negative_prompt,
                 unet_batch_one=args.unet_batch_one, seed=args.seed, output_type="pil", rng=getattr(args, "rng", "numpy"))
    out_path = get_image_path(args)
    logger.info("Saving generated image to %s", out_path)
    image["images"][0].save(out_path)
    return out_path


if __name__ == "__main__":
    logging.basicConfig(level=logging.INFO)
    main(build_parser().parse_args())
