"""Multi-GPU plumbing: one process per GPU, independent prompts sharded over ranks.

The reference has no parallelism at all; the path shards naturally
because every prompt (and each CFG half) is an independent UNet evaluation - GroupNorm /
LayerNorm are per-sample - so there is NO data-path collective.  RCCL (torch.distributed backend
"nccl" on ROCm; "gloo" in the CPU tests) only moves the prompt embeddings out at the start
(broadcast, 315 KB per prompt) and the results back at the end (all_gather, 64 KB of latents
per prompt).  Per-step inter-GPU traffic is zero.

Inside ONE GPU the same independence is used a second way (run_concurrent): a CFG-batch-2 step is latency-bound on an MI355X
(a batch-1 step takes 95 % of a batch-2 step, profiles/r04_shortcut_side_stream_rejected.txt), so several prompts served by
several handles - each with its own HIP stream and step graph - overlap on the 256 CUs where one stream leaves them idle.
"""
import threading

import numpy as np


def run_concurrent(jobs):
    """Run the zero-argument callables ``jobs`` at the same time, one host thread each, and return their results in order.
    Meant for device-resident loops of DIFFERENT handles (``HipModel.denoise_loop``): every handle owns a HIP stream and ctypes
    releases the GIL for the duration of the call, so the streams' graphs execute concurrently on the GPU.  An exception in
    any job is re-raised here after all of them have finished.  One job is simply called."""
    jobs = list(jobs)
    if len(jobs) == 1:
        return [jobs[0]()]
    results, errors = [None] * len(jobs), [None] * len(jobs)

    def work(i):
        try:
            results[i] = jobs[i]()
        except BaseException as e:   # noqa: BLE001 - handed to the caller below
            errors[i] = e

    threads = [threading.Thread(target=work, args=(i,), name=f"sd-stream-{i}") for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None:
            raise e
    return results


def shard_prompts(n_prompts, world_size):
    """Contiguous balanced partition: rank r owns prompts shard_prompts(n, w)[r]."""
    base, extra = divmod(n_prompts, world_size)
    out, start = [], 0
    for r in range(world_size):
        n = base + (1 if r < extra else 0)
        out.append(list(range(start, start + n)))
        start += n
    return out


def _device(dist, local_rank):
    import torch
    if dist is not None and dist.get_backend() == "nccl":
        return torch.device("cuda", local_rank)
    return torch.device("cpu")


def broadcast_array(arr, shape, dtype, dist, local_rank=0, src=0):
    """Rank `src` passes the array, the others pass None; everyone gets a numpy copy."""
    if dist is None:
        return np.asarray(arr, dtype=dtype).reshape(shape)
    import torch
    dev = _device(dist, local_rank)
    if dist.get_rank() == src:
        t = torch.from_numpy(np.ascontiguousarray(arr, dtype=dtype).reshape(shape)).to(dev)
    else:
        t = torch.empty(shape, dtype=torch.from_numpy(np.empty(0, dtype)).dtype, device=dev)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


def gather_arrays(arr, dist, local_rank=0):
    """all_gather equal-shaped per-rank arrays along axis 0 (rank order)."""
    if dist is None:
        return np.asarray(arr)
    import torch
    dev = _device(dist, local_rank)
    t = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
    outs = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(outs, t)
    return torch.cat(outs, dim=0).cpu().numpy()


def cfg_batch(ehs_pairs):
    """(P, 2, C, 1, L) per-prompt [uncond, cond] embeddings -> the UNet batch (2P, C, 1, L) in the reference's
    order [uncond_0..uncond_{P-1}, cond_0..cond_{P-1}] (pipeline.py:245, batched like the Swift imageCount loop)."""
    ehs_pairs = np.asarray(ehs_pairs)
    return np.concatenate([ehs_pairs[:, 0], ehs_pairs[:, 1]])


def _check_sharded_inputs(ehs_pairs, latents, world):
    """None, or why rank 0's inputs cannot be sharded (evaluated on rank 0 only, before any collective)."""
    try:
        e, l = np.asarray(ehs_pairs), np.asarray(latents)
    except Exception as exc:      # ragged lists etc.
        return f"run_sharded: inputs are not arrays ({exc})"
    if e.ndim != 5 or e.shape[1] != 2 or e.shape[3] != 1:
        return f"run_sharded: ehs_pairs must be (P, 2, C, 1, L), got {e.shape}"
    if l.ndim != 4 or l.shape[0] != e.shape[0]:
        return f"run_sharded: latents must be (P, 4, h, w) with P = {e.shape[0]}, got {l.shape}"
    if e.shape[0] % world:
        return f"{e.shape[0]} prompts do not split evenly over {world} ranks (static UNet batch per rank)"
    return None


def run_sharded(loop_fn, ehs_pairs, latents, dist=None, local_rank=0):
    """BASELINE config 3 in one call: rank 0 holds the text embeddings (P, 2, C, 1, L) and the initial latents
    (P, 4, h, w) of ALL prompts; they are broadcast, each rank runs ``loop_fn(latents_of_its_prompts,
    cfg_batch(embeddings_of_its_prompts)) -> final latents`` on its contiguous shard (one device-resident loop per
    rank, no data-path collective), and the final latents are all-gathered in prompt order.  Every rank must own the
    same number of prompts (static UNet batch per handle, pipeline.py:112-114)."""
    world = 1 if dist is None else dist.get_world_size()
    rank = 0 if dist is None else dist.get_rank()
    if dist is not None:
        import torch
        # rank 0 validates its inputs BEFORE the first collective and ships the verdict in the header: a user error
        # raises on every rank after the broadcast instead of leaving the other ranks waiting in it for ever
        meta = torch.zeros(8, dtype=torch.int64)
        if rank == 0:
            err = _check_sharded_inputs(ehs_pairs, latents, world)
            if err is None:
                e, l = np.asarray(ehs_pairs), np.asarray(latents)
                meta = torch.tensor(list(e.shape[:1] + e.shape[2:]) + list(l.shape[1:]) + [0], dtype=torch.int64)[:8]
            else:
                meta[7] = -1
        meta = meta.to(_device(dist, local_rank))
        dist.broadcast(meta, src=0)
        if int(meta[7]) < 0:
            raise ValueError(err if rank == 0 else "run_sharded: rank 0 rejected its inputs (see its exception)")
        p, c, one, length, lc, lh, lw = [int(v) for v in meta.tolist()[:7]]
        ehs_pairs = broadcast_array(ehs_pairs if rank == 0 else None, (p, 2, c, one, length), np.float16, dist, local_rank)
        latents = broadcast_array(latents if rank == 0 else None, (p, lc, lh, lw), np.float32, dist, local_rank)
    else:
        err = _check_sharded_inputs(ehs_pairs, latents, world)
        if err is not None:
            raise ValueError(err)
        ehs_pairs, latents = np.asarray(ehs_pairs, np.float16), np.asarray(latents, np.float32)
    n = ehs_pairs.shape[0]
    mine = shard_prompts(n, world)[rank]
    out = loop_fn(latents[mine], cfg_batch(ehs_pairs[mine]))
    return gather_arrays(np.asarray(out, np.float32), dist, local_rank)


def run_cfg_split(unet_fn, scheduler, ehs_pair, latents, guidance_scale, dist, local_rank=0):
    """DEPRECATED (kept for its gloo test only; not a supported mode, not in INTEGRATION.md's feature list): EXPERIMENTAL (kept for the SURVEY section 8e latency idea, not recommended): one MI355X runs the CFG batch of 2 in
    1.07x the time of batch 1 (profiles/r03_batch_scale_before.json), so this split saves at most 6 % of the UNet time and
    pays a host round trip per step for it.  Optional LATENCY mode for one prompt on two GPUs: rank 0 evaluates the unconditional half of
    the classifier-free-guidance batch, rank 1 the text half - each with a UNet handle of batch 1 - and the two noise
    predictions are exchanged ONCE per step (2-rank all_gather of 4 x h x w f32 = 64 KB at 512x512, one xGMI link); both ranks
    then do the guidance combine (pipeline.py:561-562) and the scheduler step redundantly, so the latents stay identical
    without a second exchange.  ``unet_fn(latents (1,4,h,w) f32, timestep, encoder_hidden_states (1,C,1,L) f16) -> noise
    (1,4,h,w) f32`` is the per-rank model call (the host-stepped `HipModel.__call__` path: a per-step collective cannot sit
    inside the device-resident loop); ``ehs_pair`` is (2, C, 1, L) = [uncond, cond] on rank 0 (broadcast to rank 1).
    Returns the final latents (the same array on both ranks).

    EXPERIMENTAL and unmeasured on hardware: every step crosses the host (D2H -> numpy -> collective -> H2D) and uses the
    host-stepped UNet call, so on one node it adds more latency than the halved batch saves; kept as the reference point for a
    device-pointer exchange, not as a recommended mode."""
    if dist is None or dist.get_world_size() != 2:
        raise ValueError("run_cfg_split needs exactly two ranks (rank 0 = uncond, rank 1 = cond)")
    rank = dist.get_rank()
    import torch
    meta = torch.zeros(7, dtype=torch.int64)      # [C, 1, L, 4, h, w, status]: status < 0 = rank 0 rejected its inputs
    msg = "run_cfg_split takes one prompt: ehs_pair (2, C, 1, L), latents (1, 4, h, w)"
    if rank == 0:
        try:
            e, l = np.asarray(ehs_pair), np.asarray(latents)
            ok = e.ndim == 4 and l.ndim == 4 and e.shape[0] == 2 and l.shape[0] == 1
        except Exception:
            ok = False
        if ok:
            meta = torch.tensor(list(e.shape[1:]) + list(l.shape[1:]) + [0], dtype=torch.int64)
        else:
            meta[6] = -1
    meta = meta.to(_device(dist, local_rank))
    dist.broadcast(meta, src=0)
    if int(meta[6]) < 0:      # raised on BOTH ranks, after the collective (a raise before it would strand rank 1 inside)
        raise ValueError(msg if rank == 0 else "run_cfg_split: rank 0 rejected its inputs: " + msg)
    c, one, length, lc, lh, lw = [int(v) for v in meta.tolist()[:6]]
    ehs_pair = broadcast_array(ehs_pair if rank == 0 else None, (2, c, one, length), np.float16, dist, local_rank)
    x = broadcast_array(latents if rank == 0 else None, (1, lc, lh, lw), np.float32, dist, local_rank)
    mine = ehs_pair[rank:rank + 1]
    for t in scheduler.timesteps:
        xin = scheduler.scale_model_input(x, t)
        eps_local = np.asarray(unet_fn(xin, t, mine), np.float32)
        both = gather_arrays(eps_local, dist, local_rank)          # [uncond, cond] in rank order
        eps = both[0:1] + guidance_scale * (both[1:2] - both[0:1])
        x = np.asarray(scheduler.step(eps, t, x).prev_sample, np.float32)
    return x
