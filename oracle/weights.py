"""Oracle (test infrastructure): deterministic synthetic checkpoints.

No model weights exist offline, so parity and perf run on seeded random
weights laid out exactly like a diffusers checkpoint (key names of SURVEY Appendix A.7).
The generator is defined here - not by any library's default init - so that the golden
vectors made in the build container (``oracle/pin_against_reference.py``) and the tests on
the GPU box regenerate bit-identical tensors from ``(shapes, seed)``:

  stream  = numpy RandomState(seed) (MT19937; its byte stream is frozen by numpy's
            compatibility policy), consumed key by key in inventory order
  value   = int16 -> uniform[-1,1) * sqrt(3) * std      (variance == std**2)
  std     = 1/sqrt(fan_in) for conv/linear weights, 0.05 for biases,
            norm weight = 1 + 0.1*u, norm bias = 0.1*u
"""
import numpy as np

_SQRT3 = 1.7320508075688772


def _uniform(rs, shape):
    n = int(np.prod(shape))
    raw = np.frombuffer(rs.bytes(2 * n), dtype="<i2")
    return (raw.astype(np.float32) * np.float32(1.0 / 32768.0)).reshape(shape)


def make_state_dict(shapes, seed=0, dtype=np.float32, gain=1.0):
    """{key: np.ndarray} for an ordered {key: shape} inventory."""
    rs = np.random.RandomState(seed)
    sd = {}
    for key, shape in shapes.items():
        u = _uniform(rs, shape)
        is_norm = ".norm" in key or key.startswith("conv_norm_out") or ".norm." in key
        if len(shape) == 1:
            if is_norm and key.endswith(".weight"):
                w = 1.0 + 0.1 * u
            elif is_norm:
                w = 0.1 * u
            else:
                w = 0.05 * _SQRT3 * u
        else:
            fan_in = int(np.prod(shape[1:]))
            w = (gain * _SQRT3 / np.sqrt(fan_in)) * u
        sd[key] = np.ascontiguousarray(w, dtype=dtype)
    return sd


def to_torch(sd):
    import torch
    return {k: torch.from_numpy(np.asarray(v, np.float32)) for k, v in sd.items()}


def round_to_fp16(sd):
    """Weights as the fp16 product stores them, widened back to fp32 for the oracle."""
    return {k: v.astype(np.float16).astype(np.float32) for k, v in sd.items()}


def seeded_normal(shape, seed, dtype=np.float32):
    """N(0,1) test inputs from the frozen legacy stream (np.random.seed + randn, as
    python_coreml_stable_diffusion/pipeline.py:331,726 draws its latents)."""
    return np.random.RandomState(seed).randn(*shape).astype(dtype)
