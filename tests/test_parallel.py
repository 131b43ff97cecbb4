"""CPU suite: the N>1 path (prompt sharding + the two collectives around the loop) on gloo,
world_size 2, plus the host schedulers against the oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import scheduler_ref
from python_hip_stable_diffusion import parallel, schedulers


def test_shard_prompts_is_a_balanced_partition():
    for n, w in [(16, 8), (2, 2), (5, 2), (1, 4), (0, 3), (17, 8)]:
        shards = parallel.shard_prompts(n, w)
        assert len(shards) == w
        assert sorted(sum(shards, [])) == list(range(n))
        sizes = [len(s) for s in shards]
        assert max(sizes) - min(sizes) <= 1
    assert parallel.shard_prompts(16, 8)[3] == [6, 7]          # BASELINE config 3: 2 prompts per GPU


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        n_prompts = 4
        emb = np.arange(n_prompts * 6, dtype=np.float16).reshape(n_prompts, 6) if rank == 0 else None
        emb = parallel.broadcast_array(emb, (n_prompts, 6), np.float16, dist)
        mine = parallel.shard_prompts(n_prompts, world)[rank]
        # stand-in for the per-rank denoise loop: a deterministic function of the prompt's data
        local = np.stack([emb[g].astype(np.float32) * 2 + g for g in mine])
        allr = parallel.gather_arrays(local, dist)
        q.put((rank, emb.tolist(), allr.tolist()))
    finally:
        dist.destroy_process_group()


def test_broadcast_shard_gather_world_size_two():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    emb = np.arange(24, dtype=np.float16).reshape(4, 6)
    want = np.stack([emb[g].astype(np.float32) * 2 + g for g in range(4)])
    for rank, got_emb, got_all in results:
        assert np.array_equal(np.array(got_emb, np.float16), emb)
        assert np.array_equal(np.array(got_all, np.float32), want), rank     # rank order == prompt order


def _stub_loop(latents, ehs):
    """CPU stand-in for HipModel.denoise_loop: any function that treats prompts independently, reading the
    [uncond..., cond...] batch layout the way the UNet does."""
    n = latents.shape[0]
    assert ehs.shape[0] == 2 * n
    u, c = ehs[:n].astype(np.float32), ehs[n:].astype(np.float32)
    g = (u + 7.5 * (c - u)).mean(axis=(1, 2, 3))
    return latents * 0.5 + g[:, None, None, None]


def _config3_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(0)
        ehs = rs.randn(4, 2, 16, 1, 77).astype(np.float16) if rank == 0 else None
        lat = rs.randn(4, 4, 8, 8).astype(np.float32) if rank == 0 else None
        seen = []

        def loop(latents, e):
            seen.append((latents.shape, e.shape))
            return _stub_loop(latents, e)

        out = parallel.run_sharded(loop, ehs, lat, dist)
        q.put((rank, seen, out))
    finally:
        dist.destroy_process_group()


def test_config3_two_prompts_per_rank_world_size_two():
    """BASELINE config 3's structure (16 prompts over 8 GPUs = 2 per rank, UNet batch 4) at world size 2 on gloo:
    broadcast, contiguous sharding, [uncond..., cond...] batch per rank, all-gather in prompt order."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_config3_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rs = np.random.RandomState(0)
    ehs = rs.randn(4, 2, 16, 1, 77).astype(np.float16)
    lat = rs.randn(4, 4, 8, 8).astype(np.float32)
    want = np.concatenate([_stub_loop(lat[i:i + 1], parallel.cfg_batch(ehs[i:i + 1])) for i in range(4)])
    for rank, seen, out in results:
        assert seen == [((2, 4, 8, 8), (4, 16, 1, 77))]            # two prompts per rank -> UNet batch 4
        np.testing.assert_allclose(out, want, rtol=1e-6)
    single = parallel.run_sharded(_stub_loop, ehs, lat, None)
    np.testing.assert_allclose(single, want, rtol=1e-6)
    with pytest.raises(ValueError):
        parallel.run_sharded(_stub_loop, ehs[:3], lat[:3], type("D", (), {"get_world_size": lambda s: 2, "get_rank": lambda s: 0,
                                                                         "get_backend": lambda s: "gloo",
                                                                         "broadcast": lambda s, t, src=0: None})())


def _cfg_split_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(3)
        ehs = rs.randn(2, 16, 1, 77).astype(np.float16) if rank == 0 else None
        lat = rs.randn(1, 4, 8, 8).astype(np.float32) if rank == 0 else None
        sch = schedulers.DDIMScheduler()
        sch.set_timesteps(5)
        calls = []

        def unet(x, t, e):   # stand-in for a batch-1 HipModel: any function of (latents, timestep, this rank's embedding)
            calls.append((x.shape, e.shape))
            return 0.3 * x + float(e.astype(np.float32).mean()) + 1e-3 * float(t)

        out = parallel.run_cfg_split(unet, sch, ehs, lat, 7.5, dist)
        q.put((rank, calls, out))
    finally:
        dist.destroy_process_group()


def test_cfg_split_latency_mode_world_size_two():
    """SURVEY section 8e optional mode: one prompt on two ranks (uncond / cond), one 2-rank exchange of the noise prediction per
    step, redundant guidance combine + scheduler step - equal to the single-process CFG loop (pipeline.py:500-573)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_cfg_split_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rs = np.random.RandomState(3)
    ehs = rs.randn(2, 16, 1, 77).astype(np.float16)
    x = rs.randn(1, 4, 8, 8).astype(np.float32)
    sch = schedulers.DDIMScheduler()
    sch.set_timesteps(5)
    for t in sch.timesteps:   # the reference's loop on one process, batch [uncond, cond]
        eps = [0.3 * x + float(ehs[i].astype(np.float32).mean()) + 1e-3 * float(t) for i in range(2)]
        x = sch.step(eps[0] + 7.5 * (eps[1] - eps[0]), t, x).prev_sample
    for rank, calls, out in results:
        assert calls == [((1, 4, 8, 8), (1, 16, 1, 77))] * 5          # batch 1 per rank, one call per step
        np.testing.assert_allclose(out, x, rtol=1e-5, atol=1e-6)
    assert np.array_equal(results[0][2], results[1][2])               # both ranks hold the same latents
    with pytest.raises(ValueError):
        parallel.run_cfg_split(lambda *a: None, sch, ehs, x, 7.5, None)


def _bad_input_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        out = []
        # 3 prompts on 2 ranks; a mis-shaped embedding array; run_cfg_split with two prompts: all rejected by rank 0 BEFORE
        # its first collective - every rank must get the exception, none may be left inside the broadcast
        bad = [(np.zeros((3, 2, 16, 1, 77), np.float16), np.zeros((3, 4, 8, 8), np.float32)),
               (np.zeros((2, 16, 1, 77), np.float16), np.zeros((2, 4, 8, 8), np.float32))]
        for e, l in bad:
            try:
                parallel.run_sharded(lambda lat, ehs: lat, e if rank == 0 else None, l if rank == 0 else None, dist)
                out.append("no error")
            except ValueError as exc:
                out.append(str(exc))
        sch = schedulers.DDIMScheduler()
        sch.set_timesteps(2)
        try:
            parallel.run_cfg_split(lambda *a: None, sch, np.zeros((4, 16, 1, 77), np.float16) if rank == 0 else None,
                                   np.zeros((2, 4, 8, 8), np.float32) if rank == 0 else None, 7.5, dist)
            out.append("no error")
        except ValueError as exc:
            out.append(str(exc))
        # the group is still usable afterwards (nobody is stuck in a half-finished collective)
        ok = parallel.run_sharded(lambda lat, ehs: lat + 1, np.zeros((2, 2, 16, 1, 77), np.float16) if rank == 0 else None,
                                  np.zeros((2, 4, 8, 8), np.float32) if rank == 0 else None, dist)
        q.put((rank, out, float(ok.sum())))
    finally:
        dist.destroy_process_group()


def test_invalid_rank0_inputs_raise_on_every_rank_instead_of_deadlocking():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bad_input_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, msgs, total in results:
        assert len(msgs) == 3 and "no error" not in msgs, (rank, msgs)
        assert total == 2 * 4 * 8 * 8
    r0 = dict((r, m) for r, m, _ in results)[0]
    assert "do not split evenly" in r0[0] and "must be (P, 2, C, 1, L)" in r0[1] and "one prompt" in r0[2]


def test_single_process_degenerate_case():
    a = np.ones((2, 3), np.float32)
    assert np.array_equal(parallel.broadcast_array(a, (2, 3), np.float32, None), a)
    assert np.array_equal(parallel.gather_arrays(a, None), a)


@pytest.mark.parametrize("n", [20, 50, 7])
def test_ddim_scheduler_matches_oracle_and_exports_linear_tables(n):
    mine, ref = schedulers.DDIMScheduler(), scheduler_ref.DDIM()
    mine.set_timesteps(n)
    assert np.array_equal(mine.timesteps, ref.set_timesteps(n))
    rs = np.random.RandomState(n)
    x = rs.randn(1, 4, 8, 8).astype(np.float32)
    ts, coef, hist = mine.device_tables()
    assert hist == 0 and coef.shape == (n, 8) and np.array_equal(ts, mine.timesteps.astype(np.float32))
    xm, xr, xt = x.copy(), x.copy(), x.copy()
    for i, t in enumerate(mine.timesteps):
        e = rs.randn(*x.shape).astype(np.float32)
        xm = mine.step(e, t, xm).prev_sample
        xr = ref.step(e, int(t), xr)
        xt = coef[i, 0] * xt + coef[i, 1] * e          # what the device kernel computes
    np.testing.assert_allclose(xm, xr, rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(xt, xr, rtol=2e-4, atol=2e-5)


def test_pndm_scheduler_matches_oracle():
    mine, ref = schedulers.PNDMScheduler(), scheduler_ref.PNDM()
    mine.set_timesteps(20)
    assert np.array_equal(mine.timesteps, ref.set_timesteps(20))
    rs = np.random.RandomState(0)
    xm = xr = rs.randn(1, 4, 8, 8).astype(np.float32)
    for t in mine.timesteps:
        e = rs.randn(*xm.shape).astype(np.float32)
        xm = mine.step(e, t, xm).prev_sample
        xr = ref.step(e, int(t), xr)
    np.testing.assert_allclose(xm, xr, rtol=1e-5, atol=1e-6)
    assert set(schedulers.get_available_schedulers()) >= {"DDIM", "PNDM"}


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_bench_py_multi_gpu_launch_path_under_gloo_world_size_two():
    """The driver launches `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` on an 8-GPU node, and no
    GPU box available to the builder has more than one GPU: the very same launch line, world size 2, with bench.py's hidden
    --stub-model switch (gloo on CPU, a numpy stand-in for the UNet handle) - rendezvous, embedding broadcast, prompt sharding,
    barriers, max-over-ranks timing, result gather and the single JSON line of rank 0 are all bench.py's own code."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = {}
    for world, ppg in ((1, 2), (2, 1), (2, 2)):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", str(world), "--steps", "4", "--warmup", "1",
               "--repeats", "2", "--prompts-per-gpu", str(ppg), "--stub-model"]
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, r.stdout            # exactly one JSON line, from rank 0
        d = json.loads(lines[0])
        assert d["n_gpus"] == world and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak"
        assert d["prompts"] == world * ppg and d["gathered_latents"] == [world * ppg, 4, 64, 64] and d["value"] > 0
        outs[(world, ppg)] = d
    # two prompts on one rank and one prompt on each of two ranks are the same global job: same gathered result
    assert abs(outs[(1, 2)]["checksum"] - outs[(2, 1)]["checksum"]) <= 1e-6 * max(1.0, abs(outs[(1, 2)]["checksum"]))


def test_run_concurrent_runs_jobs_at_the_same_time_keeps_order_and_reraises():
    """python_hip_stable_diffusion.parallel.run_concurrent: one host thread per handle loop (bench.py --streams)."""
    import threading
    import time
    from python_hip_stable_diffusion.parallel import run_concurrent
    gate = threading.Barrier(3, timeout=20)          # passes only if all three jobs are in flight together

    def job(i):
        gate.wait()
        time.sleep(0.01 * (3 - i))                   # finish in reverse order
        return i * i

    assert run_concurrent([lambda i=i: job(i) for i in range(3)]) == [0, 1, 4]
    assert run_concurrent([lambda: "only"]) == ["only"]
    done = []
    with pytest.raises(ZeroDivisionError):
        run_concurrent([lambda: done.append(1), lambda: 1 / 0, lambda: done.append(2)])
    assert sorted(done) == [1, 2]                    # the others still ran to the end
