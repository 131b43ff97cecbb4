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
