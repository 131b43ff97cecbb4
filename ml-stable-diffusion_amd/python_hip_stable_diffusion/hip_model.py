"""``HipModel`` - the MI355X drop-in for the reference's model-runner seam.

Duck-types ``CoreMLModel`` (python_coreml_stable_diffusion/coreml_model.py:36-120): same
``expected_inputs`` attribute ({name: {"shape", "dtype"}}, coreml_model.py:58-64), same call
signature ``model(**np.ndarray) -> dict[str, np.ndarray]`` (:118-120), same input validation
and exception classes (``_verify_inputs`` :97-116), same tensor names/layouts/dtypes at the
boundary (torch2coreml.py:857-863, :905-908, :952, :988, :1412) with fp32 outputs like the
Core ML models declare (torch2coreml.py:135).  Behind it sits ``libsdmi355.so`` instead of
``MLModel.predict``; static shapes are fixed at construction exactly as conversion fixes them
for a .mlpackage (pipeline.py:112-114).  The attention implementation, a conversion-time flag
in the reference (torch2coreml.py:1678-1685 -> unet.py:39), is a per-handle run-time switch.
"""
import ctypes as C
import logging
import time

import numpy as np

from . import _lib

logger = logging.getLogger(__name__)

CROSS_ATTN_DOWN, DOWN = "CrossAttnDownBlock2D", "DownBlock2D"
CROSS_ATTN_UP, UP = "CrossAttnUpBlock2D", "UpBlock2D"

# Public HF config.json values (not in the reference: it reads them via **pipe.unet.config,
# torch2coreml.py:915).  Validated by parameter count in SURVEY.md Appendix B.
UNET_CONFIGS = {
    "stabilityai/stable-diffusion-2-1-base": dict(
        block_out_channels=(320, 640, 1280, 1280), attention_head_dim=(5, 10, 20, 20), cross_attention_dim=1024),
    "runwayml/stable-diffusion-v1-5": dict(
        block_out_channels=(320, 640, 1280, 1280), attention_head_dim=8, cross_attention_dim=768),
    "stabilityai/stable-diffusion-xl-base-1.0": dict(
        sample_size=128, block_out_channels=(320, 640, 1280), down_block_types=(DOWN, CROSS_ATTN_DOWN, CROSS_ATTN_DOWN),
        up_block_types=(CROSS_ATTN_UP, CROSS_ATTN_UP, UP), attention_head_dim=(5, 10, 20), cross_attention_dim=2048,
        transformer_layers_per_block=(1, 2, 10), addition_embed_type="text_time", addition_time_embed_dim=256,
        projection_class_embeddings_input_dim=2816),
    "stabilityai/stable-diffusion-xl-refiner-1.0": dict(
        sample_size=128, block_out_channels=(384, 768, 1536, 1536),
        down_block_types=(DOWN, CROSS_ATTN_DOWN, CROSS_ATTN_DOWN, DOWN),
        up_block_types=(UP, CROSS_ATTN_UP, CROSS_ATTN_UP, UP), attention_head_dim=(6, 12, 24, 24),
        cross_attention_dim=1280, transformer_layers_per_block=4, addition_embed_type="text_time",
        addition_time_embed_dim=256, projection_class_embeddings_input_dim=2560),
}


def normalize_unet_config(config):
    """Fill UNet2DConditionModel.__init__ defaults (unet.py:801-833) and reject what the
    reference rejects (unet.py:835-880)."""
    if isinstance(config, str):
        if config not in UNET_CONFIGS:
            raise ValueError(f"unknown model version {config!r}; known: {sorted(UNET_CONFIGS)}")
        config = UNET_CONFIGS[config]
    cfg = dict(
        in_channels=4, out_channels=4, sample_size=64, block_out_channels=(320, 640, 1280, 1280),
        down_block_types=(CROSS_ATTN_DOWN,) * 3 + (DOWN,), up_block_types=(UP,) + (CROSS_ATTN_UP,) * 3,
        layers_per_block=2, attention_head_dim=8, cross_attention_dim=768, transformer_layers_per_block=1,
        norm_num_groups=32, norm_eps=1e-5, flip_sin_to_cos=True, freq_shift=0, addition_embed_type=None,
        addition_time_embed_dim=None, projection_class_embeddings_input_dim=None, support_controlnet=False,
        only_cross_attention=False, mid_block_type="UNetMidBlock2DCrossAttn", center_input_sample=False)
    unknown_ok = {"use_linear_projection", "act_fn", "downsample_padding", "mid_block_scale_factor",
                  "time_cond_proj_dim", "upcast_attention", "resnet_time_scale_shift", "_class_name",
                  "_diffusers_version", "num_class_embeds", "dual_cross_attention", "class_embed_type",
                  "conditioning_embedding_out_channels", "num_time_ids"}
    for k, v in dict(config).items():
        if k not in cfg and k not in unknown_ok:
            logger.warning("ignoring unknown UNet config key %r", k)
        cfg[k] = v
    if cfg.get("dual_cross_attention") or cfg.get("num_class_embeds") or cfg["only_cross_attention"]:
        raise NotImplementedError("dual_cross_attention / class embeddings / only_cross_attention (unet.py:835-840)")
    if cfg["addition_embed_type"] not in (None, "text_time"):
        raise NotImplementedError(f"addition_embed_type {cfg['addition_embed_type']!r} (unet.py:868-880)")
    if cfg["mid_block_type"] != "UNetMidBlock2DCrossAttn" or cfg["center_input_sample"]:
        raise NotImplementedError("mid_block_type / center_input_sample outside the path (unet.py:919, :987)")
    n = len(cfg["block_out_channels"])
    if n > _lib.SD_MAX_LEVELS:
        raise NotImplementedError(f"more than {_lib.SD_MAX_LEVELS} resolution levels")
    for key in ("attention_head_dim", "transformer_layers_per_block"):
        v = cfg[key]
        cfg[key] = tuple([v] * n) if isinstance(v, int) else tuple(v)
    for t in cfg["down_block_types"]:
        if t not in (CROSS_ATTN_DOWN, DOWN):
            raise NotImplementedError(f"down block type {t}")
    for t in cfg["up_block_types"]:
        if t not in (CROSS_ATTN_UP, UP):
            raise NotImplementedError(f"up block type {t}")
    return cfg


def _fill(arr, values):
    for i, v in enumerate(values):
        arr[i] = int(v)


class Weights:
    """Owns an ``sd_weights`` store (checkpoint tensors keyed by diffusers names)."""

    def __init__(self, tensors=None, safetensors_path=None, prefix=None):
        self._h = C.c_void_p()
        _lib.check(_lib.lib().sd_weights_create(C.byref(self._h)))
        if safetensors_path is not None:
            _lib.check(_lib.lib().sd_weights_load_safetensors(
                self._h, str(safetensors_path).encode(), None if prefix is None else prefix.encode()))
        for name, t in (tensors or {}).items():
            self.add(name, t)

    def add(self, name, t):
        t = np.asarray(t)
        if t.dtype == np.float16:
            dt = 0
        else:
            t = t.astype(np.float32, copy=False)
            dt = 1
        t = np.ascontiguousarray(t)
        shape = (C.c_int64 * max(1, t.ndim))(*t.shape)
        _lib.check(_lib.lib().sd_weights_add(self._h, name.encode(), _lib.ptr(t), dt, shape, t.ndim))

    def __len__(self):
        return _lib.lib().sd_weights_count(self._h)

    def close(self):
        if self._h:
            _lib.lib().sd_weights_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipModel:
    """UNet / control-UNet / ControlNet on one MI355X behind the CoreMLModel interface."""

    def __init__(self, config, weights, kind="unet", batch=2, latent_height=None, latent_width=None,
                 attention_implementation="SPLIT_EINSUM", device=0, use_graph=True, context_len=77):
        if kind not in ("unet", "controlnet"):
            raise ValueError(f"kind must be 'unet' or 'controlnet', got {kind!r}")
        if attention_implementation not in _lib.ATTENTION_IMPLEMENTATIONS:
            raise ValueError(f"--attention-implementation must be one of {list(_lib.ATTENTION_IMPLEMENTATIONS)}")
        start = time.time()
        cfg = normalize_unet_config(config)
        self.config = cfg
        self.kind = kind
        h = latent_height or cfg["sample_size"]
        w = latent_width or cfg["sample_size"]
        n = len(cfg["block_out_channels"])
        xl = cfg["addition_embed_type"] == "text_time"
        c = _lib.UNetConfig()
        c.batch, c.in_channels, c.out_channels = batch, cfg["in_channels"], cfg["out_channels"]
        c.height, c.width, c.n_levels = h, w, n
        _fill(c.block_out_channels, cfg["block_out_channels"])
        _fill(c.down_cross_attn, [t == CROSS_ATTN_DOWN for t in cfg["down_block_types"]])
        _fill(c.up_cross_attn, [t == CROSS_ATTN_UP for t in cfg["up_block_types"]])
        c.layers_per_block = cfg["layers_per_block"]
        _fill(c.attention_head_dim, cfg["attention_head_dim"])
        _fill(c.transformer_layers_per_block, cfg["transformer_layers_per_block"])
        c.cross_attention_dim, c.context_len = cfg["cross_attention_dim"], context_len
        c.norm_num_groups, c.norm_eps = cfg["norm_num_groups"], cfg["norm_eps"]
        c.flip_sin_to_cos, c.freq_shift = int(cfg["flip_sin_to_cos"]), float(cfg["freq_shift"])
        self.num_time_ids = 0
        self.text_embed_dim = 0
        if xl:
            c.addition_time_embed_dim = cfg["addition_time_embed_dim"]
            c.projection_class_embeddings_input_dim = cfg["projection_class_embeddings_input_dim"]
            # base: 6 ids + 1280-d pooled text; refiner: 5 ids + 1280-d (StableDiffusionXLPipeline.swift:326-358)
            self.num_time_ids = int(cfg.get("num_time_ids") or
                                    (5 if cfg["projection_class_embeddings_input_dim"] == 2560 else 6))
            c.num_time_ids = self.num_time_ids
            self.text_embed_dim = (cfg["projection_class_embeddings_input_dim"] -
                                   self.num_time_ids * cfg["addition_time_embed_dim"])
        c.support_controlnet = int(bool(cfg["support_controlnet"]) and kind == "unet")
        c.is_controlnet = int(kind == "controlnet")
        c.attention_impl = _lib.ATTENTION_IMPLEMENTATIONS[attention_implementation]
        c.use_graph = int(use_graph)
        self._cfg_struct = c
        own = not isinstance(weights, Weights)
        wstore = weights if not own else (Weights(safetensors_path=weights) if isinstance(weights, (str, bytes))
                                          else Weights(tensors=weights))
        self._h = C.c_void_p()
        try:
            _lib.check(_lib.lib().sd_unet_create(C.byref(c), wstore._h, device, C.byref(self._h)))
        finally:
            if own:
                wstore.close()
        self.batch, self.latent_height, self.latent_width = batch, h, w
        self.attention_implementation = attention_implementation
        self.num_residuals = _lib.lib().sd_unet_num_residuals(self._h)
        self._attached = []

        f16 = np.dtype(np.float16)
        ei = {
            "sample": {"shape": (batch, cfg["in_channels"], h, w), "dtype": f16},
            "timestep": {"shape": (batch,), "dtype": f16},
            "encoder_hidden_states": {"shape": (batch, cfg["cross_attention_dim"], 1, context_len), "dtype": f16},
        }
        if xl:
            ei["time_ids"] = {"shape": (batch, self.num_time_ids), "dtype": f16}
            ei["text_embeds"] = {"shape": (batch, self.text_embed_dim), "dtype": f16}
        self._res_shapes = self._residual_shapes(cfg, batch, h, w)
        if kind == "controlnet":
            ei["controlnet_cond"] = {"shape": (batch, 3, h * 8, w * 8), "dtype": f16}
        elif c.support_controlnet:
            for i, s in enumerate(self._res_shapes):
                ei[f"additional_residual_{i}"] = {"shape": s, "dtype": f16}
        self.expected_inputs = ei
        logger.info("HipModel(%s) ready in %.1f s, %.2f GB of HBM", kind, time.time() - start,
                    _lib.lib().sd_unet_device_bytes(self._h) / 1e9)

    @staticmethod
    def _residual_shapes(cfg, batch, h, w):
        boc = cfg["block_out_channels"]
        shapes = [(batch, boc[0], h, w)]
        for i in range(len(boc)):
            shapes += [(batch, boc[i], h, w)] * cfg["layers_per_block"]
            if i != len(boc) - 1:
                h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
                shapes.append((batch, boc[i], h, w))
        shapes.append((batch, boc[-1], h, w))
        return shapes

    # coreml_model.py:97-116
    def _verify_inputs(self, **kwargs):
        for k, v in kwargs.items():
            if k in self.expected_inputs:
                if not isinstance(v, np.ndarray):
                    raise TypeError(f"Expected numpy.ndarray, got {v} for input: {k}")
                expected_dtype = self.expected_inputs[k]["dtype"]
                if not v.dtype == expected_dtype:
                    raise TypeError(f"Expected dtype {expected_dtype}, got {v.dtype} for input: {k}")
                expected_shape = self.expected_inputs[k]["shape"]
                if not v.shape == expected_shape:
                    raise TypeError(f"Expected shape {expected_shape}, got {v.shape} for input: {k}")
            else:
                raise ValueError(f"Received unexpected input kwarg: {k}")
        missing = [k for k in self.expected_inputs if k not in kwargs]
        if missing:
            raise ValueError(f"Missing input kwargs: {missing}")

    def _io(self, kwargs, keep):
        io = _lib.UNetIO()

        def put(field, name):
            if name in kwargs:
                a = np.ascontiguousarray(kwargs[name])
                keep.append(a)
                setattr(io, field, a.ctypes.data)

        for f in ("sample", "timestep", "encoder_hidden_states", "time_ids", "text_embeds", "controlnet_cond"):
            put(f, f)
        if self.kind == "unet" and self._cfg_struct.support_controlnet and not self._attached:
            arrs = [np.ascontiguousarray(kwargs[f"additional_residual_{i}"]) for i in range(self.num_residuals)]
            keep.extend(arrs)
            ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
            keep.append(ptrs)
            io.additional_residuals = C.cast(ptrs, C.POINTER(C.c_void_p))
            io.num_additional_residuals = len(arrs)
        return io

    def __call__(self, **kwargs):
        self._verify_inputs(**kwargs)
        keep = []
        io = self._io(kwargs, keep)
        if self.kind == "unet":
            out = np.empty((self.batch, self.config["out_channels"], self.latent_height, self.latent_width),
                           np.float32)
            io.noise_pred = out.ctypes.data
            _lib.check(_lib.lib().sd_unet_forward(self._h, C.byref(io)))
            return {"noise_pred": out}
        outs = [np.empty(s, np.float32) for s in self._res_shapes]
        ptrs = (C.c_void_p * len(outs))(*[o.ctypes.data for o in outs])
        io.residual_outputs = C.cast(ptrs, C.POINTER(C.c_void_p))
        _lib.check(_lib.lib().sd_unet_forward(self._h, C.byref(io)))
        return {f"additional_residual_{i}": o for i, o in enumerate(outs)}

    # ---- beyond the reference surface -----------------------------------------------------
    def set_attention_implementation(self, name):
        if name not in _lib.ATTENTION_IMPLEMENTATIONS:
            raise ValueError(f"--attention-implementation must be one of {list(_lib.ATTENTION_IMPLEMENTATIONS)}")
        _lib.check(_lib.lib().sd_unet_set_attention(self._h, _lib.ATTENTION_IMPLEMENTATIONS[name]))
        self.attention_implementation = name

    def time_forward(self, warmup=2, iters=10):
        """HIP-event milliseconds per forward on the handle's stream (inputs of the last call)."""
        ms = C.c_float(0)
        _lib.check(_lib.lib().sd_unet_time_forward(self._h, warmup, iters, C.byref(ms)))
        return ms.value

    def denoise_loop(self, latents, timesteps, coef, guidance_scale, history=0, sample_scale=None, history_state=None,
                     step_noise=None, **kwargs):
        """Device-resident pipeline.py:500-573.  latents (n_img, C, H, W) float32 -> final latents,
        per-step HIP-event milliseconds.  ``kwargs`` are the loop-invariant model inputs
        (encoder_hidden_states, SDXL time_ids / text_embeds), validated like ``__call__`` validates them
        (coreml_model.py:97-116).  ``history_state`` (history, n_img, C, H, W) float32 carries the
        scheduler's multistep history in and out (updated in place) when a loop continues on another
        handle (SDXL base -> refiner).  ``step_noise`` (len(timesteps), n_img, C, H, W) float32: added to the latents at
        the end of step i - the ancestral samplers' fresh noise, already scaled by sigma_up."""
        if self.kind != "unet":
            raise ValueError("denoise_loop needs a UNet handle")
        loop_inputs = {k: v for k, v in self.expected_inputs.items()
                       if k not in ("sample", "timestep") and not k.startswith("additional_residual_")}
        for k, v in kwargs.items():
            if k not in loop_inputs:
                raise ValueError(f"Received unexpected input kwarg: {k}")
            if not isinstance(v, np.ndarray):
                raise TypeError(f"Expected numpy.ndarray, got {v} for input: {k}")
            if v.dtype != loop_inputs[k]["dtype"]:
                raise TypeError(f"Expected dtype {loop_inputs[k]['dtype']}, got {v.dtype} for input: {k}")
            if v.shape != loop_inputs[k]["shape"]:
                raise TypeError(f"Expected shape {loop_inputs[k]['shape']}, got {v.shape} for input: {k}")
        missing = [k for k in loop_inputs if k not in kwargs]
        if missing:
            raise ValueError(f"Missing input kwargs: {missing}")
        if self._cfg_struct.support_controlnet and not self._attached:
            raise ValueError("this UNet consumes ControlNet residuals: attach_controlnets() first, or step through "
                             "__call__ with additional_residual_* inputs")
        lat = np.ascontiguousarray(latents, dtype=np.float32).copy()
        cfg_mul = 2 if guidance_scale > 1.0 else 1                                    # pipeline.py:443
        want = (self.batch // cfg_mul, self.config["in_channels"], self.latent_height, self.latent_width)
        if lat.shape != want or self.batch % cfg_mul:
            raise ValueError(f"Unexpected latents shape, got {lat.shape}, expected {want} (UNet batch {self.batch}, "
                             f"guidance_scale {guidance_scale})")
        ts = np.ascontiguousarray(timesteps, dtype=np.float32).reshape(-1)
        cf = np.ascontiguousarray(coef, dtype=np.float32)
        if cf.size != len(ts) * 8 or len(ts) < 1:
            raise ValueError(f"coef must hold len(timesteps) x 8 = {len(ts) * 8} entries, got {cf.size}")
        if not 0 <= int(history) <= 3:
            raise ValueError(f"history must be in 0..3, got {history}")
        sc = None
        if sample_scale is not None:
            sc = np.ascontiguousarray(sample_scale, dtype=np.float32).reshape(-1)
            if len(sc) != len(ts):
                raise ValueError("sample_scale must have one entry per timestep")
        hs = None
        if history_state is not None and history:
            if (not isinstance(history_state, np.ndarray) or history_state.dtype != np.float32
                    or history_state.shape != (int(history),) + lat.shape or not history_state.flags.c_contiguous):
                raise ValueError(f"history_state must be a C-contiguous float32 array of shape {(int(history),) + lat.shape}")
            hs = history_state
        keep = []
        io = self._io(kwargs, keep)
        if step_noise is not None:
            sn = np.ascontiguousarray(step_noise, dtype=np.float32)
            if sn.shape != (len(ts),) + lat.shape:
                raise ValueError(f"step_noise must have shape {(len(ts),) + lat.shape}, got {sn.shape}")
            keep.append(sn)
            io.step_noise = sn.ctypes.data
        ms = np.zeros(len(ts), np.float32)
        _lib.check(_lib.lib().sd_unet_denoise_loop(self._h, C.byref(io), _lib.fptr(lat), lat.shape[0], len(ts),
                                                   _lib.fptr(ts), _lib.fptr(cf), _lib.fptr(sc), int(history),
                                                   float(guidance_scale), _lib.fptr(hs), _lib.fptr(ms)))
        return lat, ms

    def profile(self, iters=5):
        """[(label, flop, ms)] for every launch-list entry of one forward, in launch order (sd_unet_profile)."""
        return _profile(self._h, iters)

    def attach_controlnets(self, controlnets):
        """Device-resident ControlNet hand-off (sd_unet_attach_controlnets): this UNet runs the given
        ControlNet ``HipModel`` handles before every forward / loop step and reads their residuals from
        HBM.  ``[]`` detaches (residuals then come through ``additional_residual_*`` again)."""
        controlnets = list(controlnets or [])
        for cn in controlnets:
            if not isinstance(cn, HipModel) or cn.kind != "controlnet":
                raise TypeError("attach_controlnets takes HipModel(kind='controlnet') handles")
        arr = (C.c_void_p * max(1, len(controlnets)))(*[cn._h for cn in controlnets])
        _lib.check(_lib.lib().sd_unet_attach_controlnets(self._h, C.cast(arr, C.POINTER(C.c_void_p)), len(controlnets)))
        self._attached = controlnets      # keeps the handles alive while attached
        ei = {k: v for k, v in self.expected_inputs.items() if not k.startswith("additional_residual_")}
        if self._cfg_struct.support_controlnet and not controlnets:
            for i, s_ in enumerate(self._res_shapes):
                ei[f"additional_residual_{i}"] = {"shape": s_, "dtype": np.dtype(np.float16)}
        self.expected_inputs = ei

    def set_controlnet_cond(self, cond):
        """ControlNet handle: upload the conditioning image (B, 3, 8H, 8W) fp16 and embed it once
        (controlnet.py:211-215) for the device-resident hand-off."""
        if self.kind != "controlnet":
            raise ValueError("set_controlnet_cond needs a ControlNet handle")
        want = self.expected_inputs["controlnet_cond"]
        if not isinstance(cond, np.ndarray):
            raise TypeError(f"Expected numpy.ndarray, got {cond} for input: controlnet_cond")
        if cond.dtype != want["dtype"] or cond.shape != want["shape"]:
            raise TypeError(f"Expected {want['dtype']} {want['shape']}, got {cond.dtype} {cond.shape} for input: controlnet_cond")
        c = np.ascontiguousarray(cond)
        _lib.check(_lib.lib().sd_controlnet_set_cond(self._h, _lib.ptr(c), 0))

    @property
    def device_bytes(self):
        return _lib.lib().sd_unet_device_bytes(self._h)

    def close(self):
        if getattr(self, "_h", None):
            if getattr(self, "_attached", None):
                try:
                    self.attach_controlnets([])
                except Exception:
                    pass
            _lib.lib().sd_unet_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _profile(handle, iters):
    n = C.c_int(0)
    _lib.check(_lib.lib().sd_unet_profile(handle, iters, 0, None, None, None, 0, C.byref(n)))
    cap, lb = n.value, 160
    ms = np.zeros(cap, np.float32)
    flop = np.zeros(cap, np.float64)
    labels = C.create_string_buffer(cap * lb)
    _lib.check(_lib.lib().sd_unet_profile(handle, iters, cap, _lib.fptr(ms), flop.ctypes.data_as(C.POINTER(C.c_double)),
                                          labels, lb, C.byref(n)))
    return [(labels.raw[i * lb:(i + 1) * lb].split(b"\0", 1)[0].decode(), float(flop[i]), float(ms[i])) for i in range(cap)]


VAE_CONFIGS = {   # public AutoencoderKL config of SD 1.x / 2.x (decoder side)
    "stabilityai/stable-diffusion-2-1-base": dict(latent_channels=4, out_channels=3,
                                                   block_out_channels=(128, 256, 512, 512), layers_per_block=2),
}
VAE_CONFIGS["runwayml/stable-diffusion-v1-5"] = VAE_CONFIGS["stabilityai/stable-diffusion-2-1-base"]


class HipVaeDecoder:
    """``vae_decoder`` model runner: ``decoder(post_quant_conv(z))`` (torch2coreml.py:584-594) behind
    the CoreMLModel interface: ``expected_inputs["z"]`` (pipeline.py:315) and
    ``model(z=...)["image"]`` in [-1, 1] (pipeline.py:316), NCHW fp32.
    ``dtype`` is the model's declared input dtype AND its compute precision, as in the reference's conversion
    (torch2coreml.py:570-578): ``np.float16`` = the MFMA kernels (fp16 storage, fp32 accumulation); ``np.float32`` = fp32
    activations and arithmetic end to end (``compute_fp32``), what the stock SDXL VAE needs."""

    def __init__(self, config, weights, batch=1, latent_height=64, latent_width=64, device=0, use_graph=True,
                 dtype=np.float16):
        if isinstance(config, str):
            if config not in VAE_CONFIGS:
                raise ValueError(f"unknown VAE config {config!r}")
            config = VAE_CONFIGS[config]
        self.config = dict(config)
        boc = tuple(config["block_out_channels"])
        c = _lib.UNetConfig()
        c.batch, c.in_channels, c.out_channels = batch, config["latent_channels"], config["out_channels"]
        c.height, c.width, c.n_levels = latent_height, latent_width, len(boc)
        _fill(c.block_out_channels, boc)
        c.layers_per_block = config["layers_per_block"]
        c.norm_num_groups, c.norm_eps = 32, 1e-6
        c.use_graph = int(use_graph)
        if np.dtype(dtype) not in (np.dtype(np.float16), np.dtype(np.float32)):
            raise ValueError(f"VAE dtype must be float16 or float32, got {dtype}")
        c.compute_fp32 = int(np.dtype(dtype) == np.float32)
        self.compute_dtype = np.dtype(dtype)
        own = not isinstance(weights, Weights)
        wstore = weights if not own else (Weights(safetensors_path=weights) if isinstance(weights, (str, bytes))
                                          else Weights(tensors=weights))
        self._h = C.c_void_p()
        try:
            _lib.check(_lib.lib().sd_vae_decoder_create(C.byref(c), wstore._h, device, C.byref(self._h)))
        finally:
            if own:
                wstore.close()
        self.batch, self.latent_height, self.latent_width = batch, latent_height, latent_width
        up = 2 ** (len(boc) - 1)
        self.image_shape = (batch, config["out_channels"], latent_height * up, latent_width * up)
        self.expected_inputs = {"z": {"shape": (batch, config["latent_channels"], latent_height, latent_width),
                                      "dtype": np.dtype(dtype)}}

    _verify_inputs = HipModel._verify_inputs

    def __call__(self, **kwargs):
        self._verify_inputs(**kwargs)
        z = np.ascontiguousarray(kwargs["z"])
        image = np.empty(self.image_shape, np.float32)
        _lib.check(_lib.lib().sd_vae_decode(self._h, _lib.ptr(z), 1 if z.dtype == np.float32 else 0,
                                            _lib.fptr(image), 0))
        return {"image": image}

    def profile(self, iters=5):
        return _profile(self._h, iters)

    def time_forward(self, warmup=1, iters=5):
        ms = C.c_float(0)
        _lib.check(_lib.lib().sd_unet_time_forward(self._h, warmup, iters, C.byref(ms)))
        return ms.value

    @property
    def device_bytes(self):
        return _lib.lib().sd_unet_device_bytes(self._h)

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().sd_unet_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HipVaeEncoder:
    """``vae_encoder`` model runner: ``quant_conv(encoder(x))`` (torch2coreml.py:739-749) behind the CoreMLModel
    interface: ``model(x=...)["latent"]`` = the posterior moments (B, 2*latent_channels, H/8, W/8) fp32,
    [mean | logvar]; sampling and the scale factor stay with the caller (Encoder.swift:48-90)."""

    def __init__(self, config, weights, batch=1, height=512, width=512, device=0, use_graph=True, dtype=np.float16):
        if isinstance(config, str):
            if config not in VAE_CONFIGS:
                raise ValueError(f"unknown VAE config {config!r}")
            config = VAE_CONFIGS[config]
        self.config = dict(config)
        boc = tuple(config["block_out_channels"])
        c = _lib.UNetConfig()
        c.batch, c.in_channels, c.out_channels = batch, 3, 2 * config["latent_channels"]
        c.height, c.width, c.n_levels = height, width, len(boc)
        _fill(c.block_out_channels, boc)
        c.layers_per_block = config["layers_per_block"]
        c.norm_num_groups, c.norm_eps = 32, 1e-6
        c.use_graph = int(use_graph)
        if np.dtype(dtype) not in (np.dtype(np.float16), np.dtype(np.float32)):
            raise ValueError(f"VAE dtype must be float16 or float32, got {dtype}")
        c.compute_fp32 = int(np.dtype(dtype) == np.float32)   # torch2coreml.py:726-733: the SDXL encoder is float32 too
        self.compute_dtype = np.dtype(dtype)
        own = not isinstance(weights, Weights)
        wstore = weights if not own else (Weights(safetensors_path=weights) if isinstance(weights, (str, bytes))
                                          else Weights(tensors=weights))
        self._h = C.c_void_p()
        try:
            _lib.check(_lib.lib().sd_vae_encoder_create(C.byref(c), wstore._h, device, C.byref(self._h)))
        finally:
            if own:
                wstore.close()
        down = 2 ** (len(boc) - 1)
        self.latent_shape = (batch, 2 * config["latent_channels"], height // down, width // down)
        self.expected_inputs = {"x": {"shape": (batch, 3, height, width), "dtype": np.dtype(dtype)}}

    _verify_inputs = HipModel._verify_inputs

    def __call__(self, **kwargs):
        self._verify_inputs(**kwargs)
        x = np.ascontiguousarray(kwargs["x"])
        out = np.empty(self.latent_shape, np.float32)
        _lib.check(_lib.lib().sd_vae_encode(self._h, _lib.ptr(x), 1 if x.dtype == np.float32 else 0, _lib.fptr(out), 0))
        return {"latent": out}

    @property
    def device_bytes(self):
        return _lib.lib().sd_unet_device_bytes(self._h)

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().sd_unet_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
