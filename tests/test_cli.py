"""CPU suite: the reference's CLI surface (pipeline.py:785-855 + --attention-implementation,
torch2coreml.py:1678-1685) is kept by python_hip_stable_diffusion.pipeline, and the host-side helpers
around it behave like the reference's."""
import os

import numpy as np
import pytest

from python_hip_stable_diffusion import pipeline as P


def test_parser_keeps_the_reference_flags_and_defaults():
    a = P.build_parser().parse_args(["--prompt", "a photo of an astronaut", "-i", "/models/sd21", "-o", "/tmp/out"])
    assert a.seed == 93 and a.num_inference_steps == 50 and a.guidance_scale == 7.5            # pipeline.py:800, :822, :828
    assert a.model_version == "CompVis/stable-diffusion-v1-4" and a.compute_unit == "ALL"
    assert a.scheduler is None and a.controlnet is None and a.negative_prompt is None and a.unet_batch_one is False
    assert a.attention_implementation == "SPLIT_EINSUM" and a.model_sources is None            # torch2coreml.py:1678-1685
    a = P.build_parser().parse_args([
        "--prompt", "p", "-i", "in", "-o", "out", "-s", "7", "--model-version", "stabilityai/stable-diffusion-2-1-base",
        "--compute-unit", "CPU_AND_NE", "--scheduler", "DPMSolverMultistep", "--num-inference-steps", "20",
        "--guidance-scale", "5", "--controlnet", "cn_a", "cn_b", "--controlnet-inputs", "a.png", "b.png",
        "--negative-prompt", "blurry", "--unet-batch-one", "--model-sources", "compiled",
        "--attention-implementation", "ORIGINAL"])
    assert (a.seed, a.scheduler, a.num_inference_steps, a.guidance_scale) == (7, "DPMSolverMultistep", 20, 5.0)
    assert a.controlnet == ["cn_a", "cn_b"] and a.controlnet_inputs == ["a.png", "b.png"] and a.unet_batch_one
    assert a.attention_implementation == "ORIGINAL" and a.compute_unit == "CPU_AND_NE"
    for bad in (["--scheduler", "Heun"], ["--attention-implementation", "FLASH"], ["--compute-unit", "GPU"]):
        with pytest.raises(SystemExit):
            P.build_parser().parse_args(["--prompt", "p", "-i", "in", "-o", "out"] + bad)
    assert sorted(P.SCHEDULER_MAP) == ["DDIM", "DPMSolverMultistep", "EulerAncestralDiscrete", "EulerDiscrete",
                                       "LMSDiscrete", "PNDM"]                                       # pipeline.py:592-604


def test_image_path_scheme_matches_the_reference(tmp_path):
    a = P.build_parser().parse_args(["--prompt", "a cat/dog on mars", "-i", "in", "-o", str(tmp_path), "--scheduler", "DDIM",
                                     "--model-version", "stabilityai/stable-diffusion-2-1-base"])
    path = P.get_image_path(a)                                                                   # pipeline.py:700-714
    assert os.path.isdir(os.path.dirname(path))
    assert os.path.basename(os.path.dirname(path)) == "a_cat_dog_on_mars"
    assert os.path.basename(path) == ("randomSeed_93_computeUnit_ALL_modelVersion_stabilityai_stable-diffusion-2-1-base"
                                      "_customScheduler_DDIM_numInferenceSteps50.png")
    assert "randomSeed_5_" in P.get_image_path(a, seed=5)


def test_prepare_controlnet_cond_is_rgb_chw_unit_range(tmp_path):
    from PIL import Image
    img = (np.random.RandomState(0).rand(40, 30, 3) * 255).astype(np.uint8)
    f = tmp_path / "edge.png"
    Image.fromarray(img).save(f)
    c = P.prepare_controlnet_cond(str(f), 64, 64)                                                # pipeline.py:717-721
    assert c.shape == (3, 64, 64) and c.min() >= 0.0 and c.max() <= 1.0 and c.dtype == np.float64


def test_get_hip_pipe_reports_a_missing_checkpoint_directory():
    with pytest.raises(FileNotFoundError):
        P.get_hip_pipe("/nonexistent/model/dir", "stabilityai/stable-diffusion-2-1-base")
