"""Host-side schedulers of the denoising loop (the step either side of the UNet).

The reference takes these from diffusers (pipeline.py:11-18, SCHEDULER_MAP :594-601); diffusers
is not available offline, so the ones on the path are implemented here with the diffusers call
surface the pipeline uses (``set_timesteps``, ``timesteps``, ``init_noise_sigma``,
``scale_model_input``, ``step(...).prev_sample``) and SD's scheduler config (scaled-linear betas
0.00085..0.012, 1000 train steps, steps_offset 1, epsilon prediction).  PNDM follows the
reference's own Swift implementation (swift/StableDiffusion/pipeline/Scheduler.swift:137-344).

Every scheduler here is a *linear multistep* rule: x_prev = cx * x + sum_j ce_j * eps_{t-j}.
``device_tables()`` exports (timesteps, coef[n,8], history) for ``sd_unet_denoise_loop`` so the
update runs on the GPU fused with the classifier-free-guidance combine.
"""
from types import SimpleNamespace

import numpy as np


def _alphas_cumprod(n_train=1000, beta_start=0.00085, beta_end=0.012):
    betas = np.linspace(beta_start ** 0.5, beta_end ** 0.5, n_train, dtype=np.float32) ** 2   # Scheduler.swift:168-173
    return np.cumprod(1.0 - betas, dtype=np.float32)


class DDIMScheduler:
    """eta = 0, "leading" timestep spacing, steps_offset = 1, set_alpha_to_one = False."""

    init_noise_sigma = 1.0
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012, steps_offset=1):
        self.num_train_timesteps = num_train_timesteps
        self.steps_offset = steps_offset
        self.alphas_cumprod = _alphas_cumprod(num_train_timesteps, beta_start, beta_end)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.timesteps = None

    def set_timesteps(self, num_inference_steps):
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        self.timesteps = (np.arange(num_inference_steps) * ratio).round()[::-1].astype(np.int64) + self.steps_offset

    def scale_model_input(self, sample, timestep=None):
        return sample

    def _coef(self, t):
        t_prev = t - self.num_train_timesteps // self.num_inference_steps
        a_t = float(self.alphas_cumprod[t])
        a_p = float(self.alphas_cumprod[t_prev]) if t_prev >= 0 else float(self.final_alpha_cumprod)
        cx = (a_p / a_t) ** 0.5
        ce = (1.0 - a_p) ** 0.5 - (a_p * (1.0 - a_t) / a_t) ** 0.5
        return cx, ce

    def step(self, model_output, timestep, sample, **kwargs):
        cx, ce = self._coef(int(timestep))
        prev = np.float32(cx) * np.asarray(sample, np.float32) + np.float32(ce) * np.asarray(model_output, np.float32)
        return SimpleNamespace(prev_sample=prev)

    def device_tables(self):
        coef = np.zeros((len(self.timesteps), 8), np.float32)
        for i, t in enumerate(self.timesteps):
            coef[i, 0], coef[i, 1] = self._coef(int(t))
        return self.timesteps.astype(np.float32), coef, 0


class PNDMScheduler:
    """PLMS (skip_prk_steps) as in Scheduler.swift:137-344: 4th-order linear multistep on eps with
    the two-evaluation warm-up at the first timestep."""

    init_noise_sigma = 1.0
    order = 1

    def __init__(self, num_train_timesteps=1000, beta_start=0.00085, beta_end=0.012):
        self.num_train_timesteps = num_train_timesteps
        self.alphas_cumprod = _alphas_cumprod(num_train_timesteps, beta_start, beta_end)
        self.timesteps = None

    def set_timesteps(self, num_inference_steps):
        self.num_inference_steps = num_inference_steps
        ratio = float(self.num_train_timesteps // num_inference_steps)
        fwd = [int(round(i * ratio)) + 1 for i in range(num_inference_steps)]
        ts = fwd[:-1] + [fwd[-2]] + [fwd[-1]]
        self.timesteps = np.array(ts[::-1], dtype=np.int64)
        self.counter, self.ets, self.cur_sample = 0, [], None

    def scale_model_input(self, sample, timestep=None):
        return sample

    def step(self, model_output, timestep, sample, **kwargs):
        eps = np.asarray(model_output, np.float32)
        x = np.asarray(sample, np.float32)
        t = int(timestep)
        inc = self.num_train_timesteps // self.num_inference_steps
        prev = t - inc
        if self.counter != 1:
            self.ets = self.ets[-3:]
            self.ets.append(eps)
        else:
            prev, t = t, t + inc
        e = self.ets
        if len(e) == 1 and self.counter == 0:
            out, self.cur_sample = eps, x
        elif len(e) == 1 and self.counter == 1:
            out, x, self.cur_sample = 0.5 * eps + 0.5 * e[-1], self.cur_sample, None
        elif len(e) == 2:
            out = 1.5 * e[-1] - 0.5 * e[-2]
        elif len(e) == 3:
            out = (23 * e[-1] - 16 * e[-2] + 5 * e[-3]) / 12.0
        else:
            out = (55 * e[-1] - 59 * e[-2] + 37 * e[-3] - 9 * e[-4]) / 24.0
        self.counter += 1
        a_t, a_p = self.alphas_cumprod[t], self.alphas_cumprod[max(0, prev)]
        sample_coeff = (a_p / a_t) ** 0.5
        denom = a_t * (1 - a_p) ** 0.5 + (a_t * (1 - a_t) * a_p) ** 0.5
        return SimpleNamespace(prev_sample=(sample_coeff * x - (a_p - a_t) / denom * out).astype(np.float32))


SCHEDULER_MAP = {"DDIM": DDIMScheduler, "PNDM": PNDMScheduler}   # pipeline.py:594-601 (subset on the path)


def get_available_schedulers():
    return dict(SCHEDULER_MAP)
