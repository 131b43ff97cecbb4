"""``HipTextEncoder`` - the CLIP text encoder(s) behind the reference's model-runner seam.

Duck-types the ``text_encoder`` / ``text_encoder_2`` ``CoreMLModel`` objects the pipeline calls
(python_coreml_stable_diffusion/pipeline.py:151-175): ``expected_inputs["input_ids"]`` is (1, 77)
float32 (pipeline.py:173 passes the token ids as float32, a coremltools quirk), the call returns
``{"last_hidden_state", "pooled_outputs"}`` or, for SDXL, ``{"hidden_embeds", "pooled_outputs"}``
(names fixed at torch2coreml.py:439-441).  The arithmetic is transformers' CLIPTextModel /
CLIPTextModelWithProjection (third-party; transformers==4.44.2 in the reference's setup.py:20) with
the causal mask of torch2coreml.py:363-377, run by ``libsdmi355.so`` (csrc/text_encoder.cpp).
Tokenisation stays with transformers' CLIPTokenizer exactly as in the reference (pipeline.py:146-150).
"""
import ctypes as C
import json
import os

import numpy as np

from . import _lib
from .hip_model import Weights

_ACTS = {"quick_gelu": 0, "gelu": 1}


class TextEncoderConfig(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("hidden_size", C.c_int32), ("intermediate_size", C.c_int32),
                ("num_hidden_layers", C.c_int32), ("num_attention_heads", C.c_int32),
                ("max_position_embeddings", C.c_int32), ("hidden_act", C.c_int32), ("projection_dim", C.c_int32),
                ("layer_norm_eps", C.c_float), ("use_graph", C.c_int32)]


def load_tokenizer(folder):
    """transformers' CLIPTokenizer from a checkpoint's ``tokenizer/`` folder (vocab.json + merges.txt)."""
    if not os.path.isdir(folder):
        raise FileNotFoundError(f"{folder} not found (coreml_model.py:176-178)")
    from transformers import CLIPTokenizer
    return CLIPTokenizer.from_pretrained(folder)


class HipTextEncoder:
    def __init__(self, config, weights, xl=False, with_projection=None, device=0, use_graph=True):
        cfg = dict(config)
        act = cfg.get("hidden_act", "quick_gelu")
        if act not in _ACTS:
            raise NotImplementedError(f"hidden_act {act!r} (CLIP text towers use quick_gelu or gelu)")
        if with_projection is None:
            with_projection = "WithProjection" in str((cfg.get("architectures") or [""])[0])
        self.config = cfg
        self.xl = xl
        self.eos_token_id = cfg.get("eos_token_id", 2)
        c = TextEncoderConfig()
        c.vocab_size, c.hidden_size = cfg["vocab_size"], cfg["hidden_size"]
        c.intermediate_size, c.num_hidden_layers = cfg["intermediate_size"], cfg["num_hidden_layers"]
        c.num_attention_heads = cfg["num_attention_heads"]
        c.max_position_embeddings = cfg.get("max_position_embeddings", 77)
        c.hidden_act = _ACTS[act]
        c.projection_dim = int(cfg.get("projection_dim", cfg["hidden_size"])) if with_projection else 0
        c.layer_norm_eps = float(cfg.get("layer_norm_eps", 1e-5))
        c.use_graph = int(use_graph)
        self._cfg_struct = c
        own = not isinstance(weights, Weights)
        wstore = weights if not own else (Weights(safetensors_path=weights) if isinstance(weights, (str, bytes))
                                          else Weights(tensors=weights))
        self._h = C.c_void_p()
        try:
            _lib.check(_lib.lib().sd_text_encoder_create(C.byref(c), wstore._h, device, C.byref(self._h)))
        finally:
            if own:
                wstore.close()
        self.seq_len, self.hidden_size = c.max_position_embeddings, c.hidden_size
        self.pooled_dim = c.projection_dim or c.hidden_size
        self.expected_inputs = {"input_ids": {"shape": (1, self.seq_len), "dtype": np.dtype(np.float32)}}

    @classmethod
    def from_pretrained(cls, folder, **kw):
        if not os.path.isdir(folder):
            raise FileNotFoundError(f"{folder} not found (coreml_model.py:176-178)")
        with open(os.path.join(folder, "config.json")) as f:
            cfg = json.load(f)
        for name in ("model.fp16.safetensors", "model.safetensors"):
            if os.path.exists(os.path.join(folder, name)):
                return cls(cfg, os.path.join(folder, name), **kw)
        raise FileNotFoundError(f"no .safetensors checkpoint under {folder}")

    def eos_index(self, ids):
        """Position whose final-LayerNorm row is pooled (CLIPTextTransformer.forward): the arg-max token id for the
        legacy eos_token_id == 2 configs, else the first occurrence of eos_token_id."""
        ids = np.asarray(ids).reshape(-1)
        if self.eos_token_id == 2:
            return int(np.argmax(ids))
        hits = np.nonzero(ids == self.eos_token_id)[0]
        return int(hits[0]) if len(hits) else 0

    def __call__(self, **kwargs):
        for k, v in kwargs.items():                                   # coreml_model.py:97-116
            if k != "input_ids":
                raise ValueError(f"Received unexpected input kwarg: {k}")
            if not isinstance(v, np.ndarray):
                raise TypeError(f"Expected numpy.ndarray, got {v} for input: {k}")
            if v.dtype != np.float32:
                raise TypeError(f"Expected dtype float32, got {v.dtype} for input: {k}")
            if v.shape != (1, self.seq_len):
                raise TypeError(f"Expected shape {(1, self.seq_len)}, got {v.shape} for input: {k}")
        if "input_ids" not in kwargs:
            raise ValueError("Missing input kwargs: ['input_ids']")
        ids = np.ascontiguousarray(np.rint(kwargs["input_ids"]).astype(np.int32))
        hidden = np.empty((1, self.seq_len, self.hidden_size), np.float32)
        pooled = np.empty((1, self.pooled_dim), np.float32)
        i32p = ids.ctypes.data_as(C.POINTER(C.c_int32))
        if self.xl:
            _lib.check(_lib.lib().sd_text_encoder_encode(self._h, i32p, self.eos_index(ids), None, _lib.fptr(hidden),
                                                         _lib.fptr(pooled)))
            return {"hidden_embeds": hidden, "pooled_outputs": pooled}
        _lib.check(_lib.lib().sd_text_encoder_encode(self._h, i32p, self.eos_index(ids), _lib.fptr(hidden), None,
                                                     _lib.fptr(pooled)))
        return {"last_hidden_state": hidden, "pooled_outputs": pooled}

    @property
    def device_bytes(self):
        return _lib.lib().sd_text_encoder_device_bytes(self._h)

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().sd_text_encoder_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
