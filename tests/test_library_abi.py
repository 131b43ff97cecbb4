"""CPU suite: the C-ABI library loads, exports every symbol include/sd_mi355x.h declares, and its
host-only entry points behave (no compute call needs a GPU here)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from oracle import rng_ref
from python_hip_stable_diffusion import _lib, hip_model


def _header_functions():
    text = open(os.path.join(ROOT, "include", "sd_mi355x.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sd_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(sdlib):
    declared = _header_functions()
    assert len(declared) >= 20
    bound = sorted(name for name, _, _ in _lib.SYMBOLS)
    assert declared == bound, set(declared) ^ set(bound)
    for name in declared:
        assert hasattr(sdlib, name), name


def test_version_and_error_string(sdlib):
    assert b"gfx950" in sdlib.sd_version()
    assert isinstance(sdlib.sd_last_error(), bytes)


def test_numpy_randn_is_bit_exact_host_side(sdlib):
    """StableDiffusionTests.swift:52-62 golden + numpy itself (pipeline.py:331,:726 stream)."""
    got = _lib.numpy_randn(rng_ref.GOLDEN_SEED, rng_ref.GOLDEN_COUNT)
    np.testing.assert_allclose(got[-5:], rng_ref.GOLDEN_LAST5, atol=1e-8)
    np.random.seed(rng_ref.GOLDEN_SEED)
    assert np.array_equal(got, np.random.randn(rng_ref.GOLDEN_COUNT))
    np.random.seed(93)                                   # the CLI default seed (pipeline.py:800)
    assert np.array_equal(_lib.numpy_randn(93, 4 * 64 * 64 + 1), np.random.randn(4 * 64 * 64 + 1))
    assert _lib.numpy_randn(0, 0).shape == (0,)


def test_weight_store_and_safetensors_reader(sdlib, tmp_path):
    from safetensors.numpy import save_file
    rs = np.random.RandomState(0)
    tensors = {"unet.conv_in.weight": rs.randn(8, 4, 3, 3).astype(np.float16),
               "unet.conv_in.bias": rs.randn(8).astype(np.float32)}
    path = tmp_path / "w.safetensors"
    save_file(tensors, str(path), metadata={"format": "pt"})
    w = hip_model.Weights(safetensors_path=str(path), prefix="unet.")
    assert len(w) == 2
    w.add("extra", np.zeros((2, 2), np.float32))
    assert len(w) == 3
    w.close()
    with pytest.raises(FileNotFoundError):
        hip_model.Weights(safetensors_path=str(tmp_path / "missing.safetensors"))


def test_config_normalisation_mirrors_reference_rejections():
    cfg = hip_model.normalize_unet_config("stabilityai/stable-diffusion-2-1-base")
    assert cfg["attention_head_dim"] == (5, 10, 20, 20) and cfg["cross_attention_dim"] == 1024
    with pytest.raises(NotImplementedError):      # unet.py:835-840
        hip_model.normalize_unet_config(dict(only_cross_attention=True))
    with pytest.raises(NotImplementedError):      # unet.py:868-880
        hip_model.normalize_unet_config(dict(addition_embed_type="text"))
    with pytest.raises(ValueError):
        hip_model.normalize_unet_config("no/such-model")


def test_no_cpu_fallback_without_a_gpu(sdlib):
    """On a box without a GPU every compute entry point fails loudly (RuntimeError), it never
    computes on the host."""
    if sdlib.sd_device_count() > 0:
        pytest.skip("a GPU is visible: the loud-failure path is exercised on CPU-only boxes")
    q = np.zeros((1, 64, 1, 64), np.float16)
    with pytest.raises(RuntimeError):
        _lib.attention("ORIGINAL", q, q, q, 1, 64)
    with pytest.raises(RuntimeError):
        hip_model.HipModel(dict(block_out_channels=(32, 64), down_block_types=("CrossAttnDownBlock2D", "DownBlock2D"),
                                up_block_types=("UpBlock2D", "CrossAttnUpBlock2D"), layers_per_block=1,
                                attention_head_dim=(2, 4), cross_attention_dim=48, sample_size=8), weights={})


def test_argument_validation_happens_before_any_device_work():
    with pytest.raises(ValueError):
        _lib.attention("FLASH", np.zeros((1, 64, 1, 8), np.float16), np.zeros((1, 64, 1, 8), np.float16),
                       np.zeros((1, 64, 1, 8), np.float16), 1, 64)
    with pytest.raises(ValueError):
        hip_model.HipModel("stabilityai/stable-diffusion-2-1-base", weights={}, attention_implementation="FAST")


def test_torch_cpu_and_philox_streams_match_their_restatements_and_torch(sdlib):
    """TorchRandomSource.swift:116-150 / NvRandomSource.swift:25-80.  The reference has no golden for either; the CPU
    stream is additionally checked against torch itself (float32 vectorised math there: agreement ~1e-6, not bits)."""
    import torch
    for seed, n in ((93, 4 * 64 * 64), (7, 16), (7, 5), (12345, 1000 + 7)):
        got = _lib.torch_randn(seed, n)
        want = np.array(rng_ref.TorchCpuRandom(seed).randn(n))
        assert np.array_equal(got, want), (seed, n)
        torch.manual_seed(seed)
        t = torch.randn(n).numpy()
        np.testing.assert_allclose(got, t, atol=5e-6, rtol=0)
    for seed, off, n in ((93, 0, 64), ((1 << 40) + 5, 3, 33)):
        got = _lib.philox_randn(seed, n, offset=off)
        np.testing.assert_allclose(got, np.array(rng_ref.philox_randn(seed, off, n)), atol=1e-12, rtol=0)
    g = _lib.philox_randn(93, 100000)
    assert abs(g.mean()) < 0.02 and abs(g.std() - 1.0) < 0.02


def test_tune_abi_is_dead_without_the_environment_switch():
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k != "SD_TUNE"}
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "from python_hip_stable_diffusion import _lib\n"
            "rc = _lib.lib().sd_tune_set_candidate(3, 0, 1)\n"
            "print('RC', rc, _lib.lib().sd_last_error().decode())\n") % (ROOT, os.path.join(ROOT, "ml-stable-diffusion_amd"))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "RC -4" in r.stdout and "SD_TUNE" in r.stdout, r.stdout + r.stderr


def _header_struct_fields(name):
    """Field names (arrays as 'name[N]' -> name) of `typedef struct <name> { ... } <name>;` in include/sd_mi355x.h, in order."""
    text = open(os.path.join(ROOT, "include", "sd_mi355x.h")).read()
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), text, re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if not decl:
            continue
        for part in decl.split(","):
            m = re.search(r"([A-Za-z_][A-Za-z0-9_]*)\s*(\[[^\]]*\])?\s*$", part.strip())
            fields.append(m.group(1))
    return fields


def test_ctypes_struct_mirrors_follow_the_header_field_for_field():
    """sd_unet_config / sd_unet_io cross the ABI by pointer: the ctypes mirrors must list the header's fields in the header's
    order (round 3 appended compute_fp32 and step_noise), and their sizes must be what a C compiler lays out."""
    assert [f[0] for f in _lib.UNetConfig._fields_] == _header_struct_fields("sd_unet_config")
    assert [f[0] for f in _lib.UNetIO._fields_] == _header_struct_fields("sd_unet_io")
    n = _lib.SD_MAX_LEVELS
    assert C.sizeof(_lib.UNetConfig) == 4 * (6 + 3 * n + 1 + 2 * n + 15)     # all members 4 bytes wide: no padding
    assert C.sizeof(_lib.UNetIO) % 8 == 0 and _lib.UNetIO.step_noise.offset == C.sizeof(_lib.UNetIO) - 8


def test_vae_compute_precision_is_validated_before_any_device_work():
    """HipVaeDecoder / HipVaeEncoder(dtype=...) select the compute precision (fp16 MFMA kernels or the fp32 path the reference
    converts the SDXL VAE with, torch2coreml.py:570-578): anything else is refused on the host, without touching a GPU."""
    from python_hip_stable_diffusion import HipVaeDecoder, HipVaeEncoder
    cfg = dict(latent_channels=4, out_channels=3, block_out_channels=(32, 32), layers_per_block=1)
    for cls in (HipVaeDecoder, HipVaeEncoder):
        with pytest.raises(ValueError, match="float16 or float32"):
            cls(cfg, {}, dtype=np.float64)
    c = _lib.UNetConfig()
    assert c.compute_fp32 == 0                       # default-constructed configs keep the fp16-storage kernels
