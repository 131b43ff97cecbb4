"""``HipStableDiffusionPipeline`` - the reference's generation loop over MI355X model runners.

Mirrors ``CoreMLStableDiffusionPipeline`` (python_coreml_stable_diffusion/pipeline.py:47-589):
same constructor arguments, same ``__call__`` arguments and semantics (prompt, height/width
taken from the model, num_inference_steps, guidance_scale, negative_prompt, latents,
callback/callback_steps, controlnet_cond, SDXL size conditioning, unet_batch_one), same
tensor hand-offs to the model runners.  Differences, all additive:
  * model runners are ``HipModel`` objects (``libsdmi355.so``) instead of ``CoreMLModel``;
  * when no per-step callback / ControlNet is requested and the scheduler exports linear
    tables, the whole loop (pipeline.py:500-573) runs device-resident in one call
    (``sd_unet_denoise_loop``); otherwise it steps through the boundary exactly like the
    reference;
  * ``seed`` is an argument (the reference seeds numpy globally in ``main``, pipeline.py:726) and
    the initial latents come from the bit-exact numpy legacy stream implemented in the library
    (``sd_numpy_randn``), so results do not depend on global RNG state.
"""
import logging
from types import SimpleNamespace

import numpy as np

from . import _lib

logger = logging.getLogger(__name__)
VAE_DECODER_UPSAMPLE_FACTOR = 8          # pipeline.py:108
VAE_SCALING_FACTOR = 0.18215             # pipeline.py:314 (SD 1.x / 2.x); SDXL 0.13025 (main.swift:124)


class HipStableDiffusionPipeline:
    def __init__(self, text_encoder, unet, vae_decoder, scheduler, tokenizer, controlnet=None, xl=False,
                 force_zeros_for_empty_prompt=True, feature_extractor=None, safety_checker=None,
                 text_encoder_2=None, tokenizer_2=None, vae_scaling_factor=None):
        self.text_encoder, self.text_encoder_2 = text_encoder, text_encoder_2
        self.tokenizer, self.tokenizer_2 = tokenizer, tokenizer_2
        self.unet, self.vae_decoder, self.scheduler = unet, vae_decoder, scheduler
        self.controlnet = controlnet
        self.xl = xl
        self.force_zeros_for_empty_prompt = force_zeros_for_empty_prompt
        self.feature_extractor, self.safety_checker = feature_extractor, safety_checker
        self.vae_scaling_factor = vae_scaling_factor or (0.13025 if xl else VAE_SCALING_FACTOR)
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
                       negative_prompt_2=None):
        batch_size = len(prompt) if isinstance(prompt, list) else 1
        if self.xl:
            prompts = [prompt, prompt_2 if prompt_2 is not None else prompt]
            if self.tokenizer is not None:
                tokenizers, encoders = [self.tokenizer, self.tokenizer_2], [self.text_encoder, self.text_encoder_2]
            else:   # refiner: only tokenizer_2 / text_encoder_2
                tokenizers, encoders = [self.tokenizer_2], [self.text_encoder_2]
            key = "hidden_embeds"
        else:
            prompts, tokenizers, encoders, key = [prompt], [self.tokenizer], [self.text_encoder], "last_hidden_state"

        def encode(texts):
            embeds, pooled = [], None
            for text, tok, enc in zip(texts, tokenizers, encoders):
                ids = tok(text, padding="max_length", max_length=tok.model_max_length, truncation=True,
                          return_tensors="np").input_ids
                out = enc(input_ids=ids.astype(np.float32))          # ids passed as float32 (pipeline.py:173)
                embeds.append(out[key])
                if self.xl:
                    pooled = out["pooled_outputs"]
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
                neg, neg_pooled = encode([negative_prompt, negative_prompt_2])
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
        image = self.vae_decoder(z=latents.astype(dtype))["image"]
        image = np.clip(image / 2 + 0.5, 0, 1)
        return image.transpose((0, 2, 3, 1))

    # ---- pipeline.py:322-344 ---------------------------------------------------------------
    def prepare_latents(self, batch_size, num_channels_latents, height, width, latents=None, seed=None):
        shape = (batch_size, num_channels_latents, self.height // 8, self.width // 8)
        if latents is None:
            n = int(np.prod(shape))
            if seed is None:
                latents = np.random.randn(*shape).astype(np.float16)           # reference behaviour (:331)
            else:
                latents = _lib.numpy_randn(int(seed), n).reshape(shape).astype(np.float16)
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

    @staticmethod
    def _get_add_time_ids(original_size, crops_coords_top_left, target_size, dtype):
        return np.array(list(original_size + crops_coords_top_left + target_size)).astype(dtype)   # :398-401

    # ---- pipeline.py:403-589 ---------------------------------------------------------------
    def __call__(self, prompt, height=512, width=512, num_inference_steps=50, guidance_scale=7.5, negative_prompt=None,
                 num_images_per_prompt=1, eta=0.0, latents=None, output_type="np", return_dict=True, callback=None,
                 callback_steps=1, controlnet_cond=None, original_size=None, crops_coords_top_left=(0, 0),
                 target_size=None, unet_batch_one=False, seed=None, device_loop=True, **kwargs):
        self.check_inputs(prompt, height, width, callback_steps)
        height, width = self.height, self.width
        original_size = original_size or (height, width)
        target_size = target_size or (height, width)
        batch_size = 1 if isinstance(prompt, str) else len(prompt)
        if batch_size > 1 or num_images_per_prompt > 1:                            # pipeline.py:434-438
            raise NotImplementedError("one prompt / one image per call (use one process per GPU for more prompts)")
        do_cfg = guidance_scale > 1.0                                              # pipeline.py:443
        text_embeddings, pooled = self._encode_prompt(prompt, None, do_cfg, negative_prompt, None)

        extra = {}
        if self.xl:                                                                # pipeline.py:458-472
            ids = self._get_add_time_ids(tuple(original_size), tuple(crops_coords_top_left), tuple(target_size),
                                         text_embeddings.dtype)
            if len(self.unet.expected_inputs["time_ids"]["shape"]) > 1:
                ids = ids[None]
            if do_cfg:
                ids = np.concatenate([ids, ids])
            extra = {"text_embeds": pooled.astype(np.float16), "time_ids": ids.astype(np.float16)}

        self.scheduler.set_timesteps(num_inference_steps)
        timesteps = self.scheduler.timesteps
        latents = self.prepare_latents(batch_size * num_images_per_prompt, self.unet.in_channels, height, width,
                                       latents, seed)
        if controlnet_cond:
            controlnet_cond = self.prepare_control_cond(controlnet_cond, do_cfg, batch_size, num_images_per_prompt)

        step_ms = None
        fused = (device_loop and callback is None and not controlnet_cond and not unet_batch_one
                 and hasattr(self.scheduler, "device_tables") and hasattr(self.unet, "denoise_loop"))
        if fused:
            ts, coef, hist = self.scheduler.device_tables()
            latents, step_ms = self.unet.denoise_loop(latents.astype(np.float32), ts, coef, guidance_scale,
                                                      history=hist,
                                                      encoder_hidden_states=text_embeddings.astype(np.float16), **extra)
        else:
            for i, t in enumerate(timesteps):                                      # pipeline.py:500-573
                x = np.concatenate([latents] * 2) if do_cfg else latents
                x = self.scheduler.scale_model_input(x, t)
                timestep = np.array([t, t] if do_cfg else [t], np.float16)
                unet_kwargs = dict(extra)
                if controlnet_cond:
                    unet_kwargs.update(self.run_controlnet(x, timestep, text_embeddings, controlnet_cond))
                if not (unet_batch_one and do_cfg):
                    noise_pred = self.unet(sample=x.astype(np.float16), timestep=timestep,
                                           encoder_hidden_states=text_embeddings.astype(np.float16),
                                           **unet_kwargs)["noise_pred"]
                    if do_cfg:
                        noise_uncond, noise_text = np.split(noise_pred, 2)
                else:
                    raise NotImplementedError("unet_batch_one needs a batch-1 handle; build HipModel(batch=1)")
                if do_cfg:
                    noise_pred = noise_uncond + guidance_scale * (noise_text - noise_uncond)
                latents = self.scheduler.step(noise_pred, t, latents.astype(np.float32)).prev_sample
                if callback is not None and i % callback_steps == 0:
                    callback(i, t, latents)

        if output_type == "latent" or self.vae_decoder is None:
            image = latents
        else:
            image = self.decode_latents(latents)
        image, has_nsfw = self.run_safety_checker(image)
        if output_type == "pil" and image.ndim == 4 and image.shape[-1] == 3:
            from PIL import Image
            image = [Image.fromarray((im * 255).round().astype("uint8")) for im in image]
        if not return_dict:
            return image, has_nsfw
        return SimpleNamespace(images=image, nsfw_content_detected=has_nsfw, step_ms=step_ms, latents=latents)
