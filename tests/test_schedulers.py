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
    "DDIM": scheduler_ref.DDIM, "PNDM": scheduler_ref.PNDM, "DPMSolverMultistep": scheduler_ref.DPMSolverMultistep,
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


def test_dpm_solver_linspace_spacing_matches_swift_default():
    x0 = np.random.RandomState(6).randn(1, 4, 8, 8).astype(np.float32)
    for n in (10, 25):
        s = schedulers.DPMSolverMultistepScheduler(timestep_spacing="linspace")
        o = scheduler_ref.DPMSolverMultistep(spacing="linspace")
        got, want = host_loop(s, x0, n), oracle_loop(o, x0, n)
        assert list(s.timesteps) == list(o.timesteps) and s.timesteps[0] == 999
        np.testing.assert_allclose(got, want, rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(table_loop(schedulers.DPMSolverMultistepScheduler(timestep_spacing="linspace"), x0, n),
                                   got, rtol=2e-4, atol=2e-4)


def test_schedule_shapes_of_the_reference_map():
    assert sorted(schedulers.get_available_schedulers()) == ["DDIM", "DPMSolverMultistep", "EulerAncestralDiscrete",
                                                             "EulerDiscrete", "LMSDiscrete", "PNDM"]      # pipeline.py:594-601
    p = schedulers.PNDMScheduler()
    p.set_timesteps(50)
    assert len(p.timesteps) == 51 and list(p.timesteps[:3]) == [981, 961, 961]         # Scheduler.swift:188-202
    p.set_timesteps(1)                                                                  # single step: no IndexError
    assert list(p.timesteps) == [1]
    d = schedulers.DPMSolverMultistepScheduler()
    d.set_timesteps(20)
    assert len(d.timesteps) == 20 and d.timesteps[0] == 1 + 20 * (999 // 21)            # :89-93
    e = schedulers.EulerDiscreteScheduler()
    e.set_timesteps(20)
    assert e.init_noise_sigma > 10 and e.sigmas[-1] == 0 and len(e.sample_scale()) == 20
    a = schedulers.EulerAncestralDiscreteScheduler(seed=3)
    assert not hasattr(a, "device_tables")                                              # stochastic -> host loop
    x0 = np.random.RandomState(5).randn(1, 4, 8, 8).astype(np.float32)
    assert np.array_equal(host_loop(a, x0, 5), host_loop(schedulers.EulerAncestralDiscreteScheduler(seed=3), x0, 5))
    with pytest.raises(NotImplementedError):
        schedulers.DDIMScheduler().step(x0, 1, x0, eta=0.5)
