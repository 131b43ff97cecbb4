"""CPU suite for the host-side schedulers (python_hip_stable_diffusion/schedulers.py): every scheduler's
``step`` against the oracle restatement (oracle/scheduler_ref.py: PNDM / DPM-Solver++ follow the
reference's Swift code, the rest are unpinned restatements of the public algorithms), and the
coefficient rows of ``device_tables()`` - what sd_unet_denoise_loop runs on the GPU - against ``step``
through a numpy statement of cfg_sched_step_kernel (csrc/misc.hip)."""
import numpy as np
import pytest

from oracle import scheduler_ref
from python_hip_stable_diffusion import schedulers

ORACLES = {
    "DDIM": scheduler_ref.DDIM, "PNDM": scheduler_ref.PNDM, "DPMSolverMultistep": scheduler_ref.DPMSolverMultistepDiffusers,
    "EulerDiscrete": scheduler_ref.EulerDiscrete, "LMSDiscrete": scheduler_ref.LMSDiscrete,
}


def fake_unet(x, t):
    """A smooth, state-dependent stand-in for eps(x, t) so that errors in any step propagate."""
    return np.tanh(0.3 * x + np.float32(t) / 1000.0).astype(np.float32) + 0.05 * x


def host_loop(s, x0, n):
    s.set_timesteps(n)
    x = x0 * np.float32(s.init_noise_sigma)
    for t in s.timesteps:
        eps = fake_unet(np.asarray(s.scale_model_input(x, t), np.float32), t)
        x = s.step(eps, t, x).prev_sample
    return x


def oracle_loop(o, x0, n):
    ts = o.set_timesteps(n)
    x = x0 * np.float32(o.init_noise_sigma)
    for t in ts:
        eps = fake_unet(np.asarray(o.scale_model_input(x, t), np.float32), t)
        x = np.asarray(o.step(eps, int(t) if float(t).is_integer() and not isinstance(o, scheduler_ref._KDiffusion) else t, x), np.float32)
    return x


def table_loop(s, x0, n):
    """numpy statement of loop_prep_kernel + cfg_sched_step_kernel (guidance folded into fake_unet)."""
    s.set_timesteps(n)
    ts, coef, hist_n = s.device_tables()
    scale = s.sample_scale()
    assert coef.shape == (len(ts), 8) and coef.dtype == np.float32 and 0 <= hist_n <= 3
    x = (x0 * np.float32(s.init_noise_sigma)).astype(np.float32)
    hist = [np.zeros_like(x) for _ in range(hist_n)]
    for k, t in enumerate(ts):
        xin = x if scale is None else x * scale[k]
        eps = fake_unet(xin.astype(np.float32), t)
        cx, cm, ch, a, b, flags = coef[k, 0], coef[k, 1], coef[k, 2:5], coef[k, 5], coef[k, 6], coef[k, 7]
        m = a * x + b * eps
        new = cx * x + cm * m
        for j in range(hist_n):
            new = new + ch[j] * hist[j]
        if flags == 0 and hist_n:
            hist = [m] + hist[:-1]
        x = new.astype(np.float32)
    return x


@pytest.mark.parametrize("name", sorted(ORACLES))
@pytest.mark.parametrize("n", [1, 2, 5, 20, 50])
def test_scheduler_step_matches_oracle_and_device_tables_match_step(name, n):
    x0 = np.random.RandomState(5).randn(1, 4, 8, 8).astype(np.float32)
    s = schedulers.SCHEDULER_MAP[name]()
    host = host_loop(s, x0, n)
    assert np.isfinite(host).all()
    want = oracle_loop(ORACLES[name](), x0, n)
    np.testing.assert_allclose(host, want, rtol=2e-4, atol=2e-4 * max(1.0, float(np.abs(want).max())))
    dev = table_loop(schedulers.SCHEDULER_MAP[name](), x0, n)
    np.testing.assert_allclose(dev, host, rtol=2e-4, atol=2e-4 * max(1.0, float(np.abs(host).max())))


@pytest.mark.parametrize("spacing", ["leading", "linspace"])
@pytest.mark.parametrize("n", [1, 2, 10, 25])
def test_dpm_solver_swift_variant_matches_the_reference_swift_scheduler(spacing, n):
    """variant "swift" = DPMSolverMultistepScheduler.swift:27-273 (lower-order final AND second-to-last steps below 15 steps,
    last step towards timestep 0); "linspace" is the Swift default spacing (:86)."""
    x0 = np.random.RandomState(6).randn(1, 4, 8, 8).astype(np.float32)
    mk = lambda: schedulers.DPMSolverMultistepScheduler(variant="swift", timestep_spacing=spacing)   # noqa: E731
    s, o = mk(), scheduler_ref.DPMSolverMultistep(spacing=spacing)
    got, want = host_loop(s, x0, n), oracle_loop(o, x0, n)
    assert list(s.timesteps) == list(o.timesteps)
    if spacing == "linspace":
        assert s.timesteps[0] == 999
    np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(table_loop(mk(), x0, n), got, rtol=2e-4, atol=2e-4)


def test_dpm_solver_variants_differ_where_the_two_references_differ():
    """diffusers (the Python pipeline's scheduler): second order at the second-to-last step, the last step lands on the x0
    prediction (final sigma 0); Swift: first order at both for n < 15, last step towards timestep 0."""
    d = schedulers.DPMSolverMultistepScheduler()
    w = schedulers.DPMSolverMultistepScheduler(variant="swift")
    for s in (d, w):
        s.set_timesteps(10)
    assert list(d.timesteps) == list(w.timesteps)
    cd, cw = d.device_tables()[1], w.device_tables()[1]
    np.testing.assert_allclose(cd[:8], cw[:8], rtol=1e-5, atol=1e-6)       # steps 0..7 agree
    assert cd[8, 2] != 0 and cw[8, 2] == 0                                  # history term at step n-2: second vs first order
    assert cd[9, 0] == 0 and cw[9, 0] > 0                                   # cx of the last step: 0 (x_prev = x0) vs sigma_0/sigma_t
    a, b, cm = cd[9, 5], cd[9, 6], cd[9, 1]
    assert abs(cm - 1.0) < 1e-6 and a > 1 and b < 0


PRED = ["epsilon", "v_prediction"]


@pytest.mark.parametrize("pred", PRED + ["sample"])
@pytest.mark.parametrize("spacing,offset,alpha_one", [("leading", 1, False), ("leading", 0, True), ("linspace", 0, False),
                                                      ("trailing", 0, False)])
@pytest.mark.parametrize("n", [1, 7, 20])
def test_ddim_prediction_types_and_spacings(pred, spacing, offset, alpha_one, n):
    x0 = np.random.RandomState(7).randn(1, 4, 8, 8).astype(np.float32)
    mk = lambda: schedulers.DDIMScheduler(prediction_type=pred, timestep_spacing=spacing, steps_offset=offset,   # noqa: E731
                                          set_alpha_to_one=alpha_one)
    o = scheduler_ref.DDIM(prediction_type=pred, spacing=spacing, steps_offset=offset, set_alpha_to_one=alpha_one)
    host, want = host_loop(mk(), x0, n), oracle_loop(o, x0, n)
    s = mk()
    s.set_timesteps(n)
    assert list(s.timesteps) == list(o.timesteps)
    tol = dict(rtol=5e-4, atol=5e-4 * max(1.0, float(np.abs(want).max())))
    np.testing.assert_allclose(host, want, **tol)
    np.testing.assert_allclose(table_loop(mk(), x0, n), host, **tol)


@pytest.mark.parametrize("pred", PRED)
@pytest.mark.parametrize("name,oracle,kw,okw", [
    ("PNDM", scheduler_ref.PNDM, {}, {}),
    ("DPMSolverMultistep", scheduler_ref.DPMSolverMultistepDiffusers, {}, {}),
    ("DPMSolverMultistep", scheduler_ref.DPMSolverMultistepDiffusers, dict(timestep_spacing="trailing"), dict(spacing="trailing")),
    ("DPMSolverMultistep", scheduler_ref.DPMSolverMultistepDiffusers, dict(final_sigmas_type="sigma_min"),
     dict(final_sigmas_type="sigma_min")),
    ("EulerDiscrete", scheduler_ref.EulerDiscrete, {}, {}),
    ("EulerDiscrete", scheduler_ref.EulerDiscrete, dict(timestep_spacing="trailing"), dict(spacing="trailing")),
    ("EulerDiscrete", scheduler_ref.EulerDiscrete, dict(timestep_spacing="linspace", steps_offset=0), dict(spacing="linspace")),
    ("LMSDiscrete", scheduler_ref.LMSDiscrete, {}, {}),
])
@pytest.mark.parametrize("n", [2, 9, 30])
def test_v_prediction_and_spacings_of_the_other_schedulers(pred, name, oracle, kw, okw, n):
    """prediction_type from the checkpoint (pipeline.py:738-741; SD2.1-768 is a v-prediction model): step() against the
    diffusers-style oracle (convert the output, then the epsilon rule), device rows against step()."""
    x0 = np.random.RandomState(8).randn(1, 4, 8, 8).astype(np.float32)
    mk = lambda: schedulers.SCHEDULER_MAP[name](prediction_type=pred, **kw)     # noqa: E731
    host = host_loop(mk(), x0, n)
    want = oracle_loop(oracle(prediction_type=pred, **okw), x0, n)
    tol = dict(rtol=5e-4, atol=5e-4 * max(1.0, float(np.abs(want).max())))
    np.testing.assert_allclose(host, want, **tol)
    np.testing.assert_allclose(table_loop(mk(), x0, n), host, **tol)


def test_from_config_follows_diffusers_semantics():
    """SCHEDULER_MAP[name].from_config(pytorch_pipe.scheduler.config) (pipeline.py:738-741): the checkpoint's values where the
    target class takes the key, the TARGET class's diffusers default otherwise; what is not implemented raises."""
    sd15 = {"_class_name": "PNDMScheduler", "_diffusers_version": "0.6.0", "beta_end": 0.012, "beta_schedule": "scaled_linear",
            "beta_start": 0.00085, "num_train_timesteps": 1000, "set_alpha_to_one": False, "skip_prk_steps": True,
            "steps_offset": 1, "trained_betas": None, "clip_sample": False}
    cfg = schedulers.load_scheduler_config(sd15)
    assert cfg["timestep_spacing"] == "leading" and cfg["prediction_type"] == "epsilon"      # PNDM's own defaults filled in
    p = schedulers.PNDMScheduler.from_config(cfg)
    ref = schedulers.PNDMScheduler()
    p.set_timesteps(20), ref.set_timesteps(20)
    assert np.array_equal(p.device_tables()[1], ref.device_tables()[1])
    e = schedulers.EulerDiscreteScheduler.from_config(cfg)                                   # keys Euler lacks are dropped
    assert e.config.timestep_spacing == "leading" and e.config.steps_offset == 1 and e.config.interpolation_type == "linear"
    d = schedulers.DPMSolverMultistepScheduler.from_config(cfg)
    assert d.config.timestep_spacing == "leading" and d.config.final_sigmas_type == "zero" and d.variant == "diffusers"
    # SD2.1 (768): a v-prediction DDIM checkpoint
    sd21 = {"_class_name": "DDIMScheduler", "beta_end": 0.012, "beta_schedule": "scaled_linear", "beta_start": 0.00085,
            "clip_sample": False, "num_train_timesteps": 1000, "prediction_type": "v_prediction", "set_alpha_to_one": False,
            "skip_prk_steps": True, "steps_offset": 1, "trained_betas": None}
    cfg = schedulers.load_scheduler_config(sd21)
    for name, cls in schedulers.SCHEDULER_MAP.items():
        assert cls.from_config(cfg).config.prediction_type == "v_prediction", name
    v = schedulers.DDIMScheduler.from_config(cfg)
    v.set_timesteps(10)
    assert np.all(v.device_tables()[1][:, 5] > 0)                                            # a = sqrt(1 - acp): x enters the noise estimate
    # a config with no key at all gets diffusers' class defaults, not SD's
    bare = schedulers.DDIMScheduler.from_config({"clip_sample": False})
    assert bare.config.beta_schedule == "linear" and bare.config.steps_offset == 0 and bare.config.set_alpha_to_one is True
    assert abs(float(bare.alphas_cumprod[-1]) - 4.036e-05) < 2e-6
    # unsupported arithmetic is refused, never ignored
    for bad in (dict(use_karras_sigmas=True), dict(prediction_type="flow"), dict(beta_schedule="sigmoid"),
                dict(timestep_spacing="log"), dict(rescale_betas_zero_snr=True)):
        with pytest.raises(NotImplementedError):
            schedulers.EulerDiscreteScheduler.from_config({**cfg, **bad})
    with pytest.raises(NotImplementedError):
        schedulers.DDIMScheduler.from_config({**cfg, "clip_sample": True})
    with pytest.raises(NotImplementedError):
        schedulers.PNDMScheduler.from_config({**cfg, "skip_prk_steps": False})
    with pytest.raises(NotImplementedError):
        schedulers.DPMSolverMultistepScheduler.from_config({**cfg, "algorithm_type": "sde-dpmsolver++"})
    with pytest.raises(NotImplementedError):
        schedulers.load_scheduler_config({"_class_name": "UniPCMultistepScheduler"})
    with pytest.raises(TypeError):
        schedulers.DDIMScheduler(no_such_key=1)


def test_schedule_shapes_of_the_reference_map():
    assert sorted(schedulers.get_available_schedulers()) == ["DDIM", "DPMSolverMultistep", "EulerAncestralDiscrete",
                                                             "EulerDiscrete", "LMSDiscrete", "PNDM"]      # pipeline.py:594-601
    p = schedulers.PNDMScheduler()
    p.set_timesteps(50)
    assert len(p.timesteps) == 51 and list(p.timesteps[:3]) == [981, 961, 961]         # Scheduler.swift:188-202
    p.set_timesteps(1)                                                                  # single step: no IndexError
    assert list(p.timesteps) == [1]
    for variant in ("swift", "diffusers"):                                              # same "leading" timesteps in both
        d = schedulers.DPMSolverMultistepScheduler(variant=variant)
        d.set_timesteps(20)
        assert len(d.timesteps) == 20 and d.timesteps[0] == 1 + 20 * (999 // 21)        # :89-93
    e = schedulers.EulerDiscreteScheduler()
    e.set_timesteps(20)
    assert e.init_noise_sigma > 10 and e.sigmas[-1] == 0 and len(e.sample_scale()) == 20
    a = schedulers.EulerAncestralDiscreteScheduler(seed=3)
    x0 = np.random.RandomState(5).randn(1, 4, 8, 8).astype(np.float32)
    assert np.array_equal(host_loop(a, x0, 5), host_loop(schedulers.EulerAncestralDiscreteScheduler(seed=3), x0, 5))
    # stochastic, still device-resident: coefficient rows for the deterministic half + the pre-drawn, sigma_up-scaled noise of
    # every step (sd_unet_io.step_noise) reproduce step() exactly when the same stream is consumed in step order
    host, dev = schedulers.EulerAncestralDiscreteScheduler(seed=9), schedulers.EulerAncestralDiscreteScheduler(seed=9)
    for s_ in (host, dev):
        s_.set_timesteps(6)
    ts, coef, hist = dev.device_tables()
    noise = dev.step_noise(x0.shape)
    assert hist == 0 and coef.shape == (6, 8) and noise.shape == (6,) + x0.shape
    x = x0 * np.float32(host.init_noise_sigma)
    y = x.copy()
    rs = np.random.RandomState(11)
    for i, t in enumerate(ts):
        eps = rs.randn(*x0.shape).astype(np.float32)
        x = host.step(eps, t, x).prev_sample
        cx, cm, _, _, _, ma, mb, _ = coef[i]
        y = cx * y + cm * (ma * y + mb * eps) + noise[i]
    np.testing.assert_allclose(y, x, rtol=2e-5, atol=2e-5)
    with pytest.raises(NotImplementedError):
        schedulers.DDIMScheduler().step(x0, 1, x0, eta=0.5)
